// aev_comm.hpp -- the one collective of the path: RCCL all-gather of the per-stream detections (BASELINE config 4),
// plus the tiny all-reduce bench.py brackets its timing with.  Included by aev_hip.hip inside its extern "C" block, behind net_detect.
//
// The reference has no distributed code at all; this is what north_star asks for ("RCCL over xGMI only for the final gather of
// predictions").  RCCL is bound at run time (dlopen of /opt/rocm's librccl.so.1, the ROCm the library itself was built with): a
// process that never creates a communicator never maps it, and no PyTorch is needed in a rank -- so every rank runs the library on
// the HIP runtime it was built for (DESIGN.md "HIP runtime").  Streams are independent: nothing here is on the data path.
// (<rccl/rccl.h> -- types and prototypes only, no link-time dependency -- is included at the top of aev_hip.hip.)

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path, error;
};
static RcclApi g_rccl;
static std::once_flag g_rccl_once;
static int rccl_api(RcclApi **out)
{
    std::call_once(g_rccl_once, []() {
        RcclApi &r = g_rccl;
        // AEV_RCCL_PATH first; then the library next to the HIP runtime this process has mapped (a ROCm tree's lib/ -- or torch's
        // lib/, if torch came first: one runtime per process, and its RCCL goes with it); then /opt/rocm; then the loader's search path
        std::vector<std::string> cand;
        if (const char *e = getenv("AEV_RCCL_PATH")) cand.push_back(e);
        Dl_info di;
        if (dladdr((const void *)&hipRuntimeGetVersion, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) { dir.resize(cut); cand.push_back(dir + "/librccl.so.1"); cand.push_back(dir + "/librccl.so"); }
        }
        cand.push_back("/opt/rocm/lib/librccl.so.1"); cand.push_back("librccl.so.1"); cand.push_back("librccl.so");
        for (const std::string &p : cand) {
            r.handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.handle) { r.path = p; break; }
            r.error = dlerror();
        }
        if (!r.handle) return;
        bool ok = true;
        auto sym = [&](const char *name) { void *s = dlsym(r.handle, name); if (!s) { ok = false; r.error = std::string("missing symbol ") + name; } return s; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(r.handle); r.handle = nullptr; }
    });
    if (!g_rccl.handle) return fail(AEV_E_HIP, "RCCL is not loadable (librccl.so.1): " + g_rccl.error);
    *out = &g_rccl;
    return AEV_OK;
}
#define NCCLCHK(api, expr)                                                                                          \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return fail(AEV_E_HIP, std::string(#expr) + ": " + (api)->GetErrorString(r_));       \
    } while (0)

struct aev_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void *d_buf = nullptr; size_t cap = 0; // send | receive staging
};
static int comm_buf(aev_comm *c, size_t bytes)
{
    if (c->cap >= bytes) return AEV_OK;
    if (c->d_buf) hipFree(c->d_buf);
    c->d_buf = nullptr; c->cap = 0;
    HIPCHK(hipMalloc(&c->d_buf, bytes));
    c->cap = bytes;
    return AEV_OK;
}

static_assert(AEV_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "aev.h: AEV_COMM_ID_BYTES is ncclUniqueId's size");

int aev_comm_unique_id(uint8_t *id)
{
    if (!id) return fail(AEV_E_ARG, "null argument");
    RcclApi *api = nullptr;
    TRY(rccl_api(&api));
    ncclUniqueId u;
    NCCLCHK(api, api->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return AEV_OK;
}

void aev_comm_destroy(aev_comm *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.handle) g_rccl.CommDestroy(c->comm);
    if (c->d_buf) hipFree(c->d_buf);
    stream_release(c->device, c->stream);
    delete c;
}

int aev_comm_create(int rank, int world, const uint8_t *id, int device, aev_comm **out)
{
    if (!out || !id) return fail(AEV_E_ARG, "null argument");
    if (world <= 0 || rank < 0 || rank >= world) return fail(AEV_E_ARG, "rank must be in 0..world-1");
    RcclApi *api = nullptr;
    TRY(rccl_api(&api));
    TRY(select_device(device));
    aev_comm *c = new aev_comm();
    c->rank = rank; c->world = world; c->device = device;
    auto body = [&]() -> int {
        TRY(stream_acquire(device, &c->stream));
        ncclUniqueId u;
        memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
        NCCLCHK(api, api->CommInitRank(&c->comm, world, u, rank));
        return AEV_OK;
    };
    const int rc = body();
    if (rc < 0) { std::string keep = g_err; aev_comm_destroy(c); g_err = keep; return rc; }
    *out = c;
    return AEV_OK;
}

int aev_comm_rank(const aev_comm *c) { return c ? c->rank : -1; }
int aev_comm_world(const aev_comm *c) { return c ? c->world : 0; }

int aev_comm_allreduce(aev_comm *c, double *values, int n, int op)
{
    if (!c || !values || n <= 0 || n > 4096 || (op != AEV_REDUCE_SUM && op != AEV_REDUCE_MAX)) return fail(AEV_E_ARG, "bad argument");
    RcclApi *api = nullptr;
    TRY(rccl_api(&api));
    HIPCHK(hipSetDevice(c->device));
    TRY(comm_buf(c, (size_t)2 * n * sizeof(double)));
    double *d = (double *)c->d_buf;
    HIPCHK(hipMemcpyAsync(d, values, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(api, api->AllReduce(d, d + n, (size_t)n, ncclFloat64, op == AEV_REDUCE_SUM ? ncclSum : ncclMax, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(values, d + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AEV_OK;
}

int aev_net_gather_detections(aev_net *n, aev_comm *c, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr, int s_cap,
                              float *host_out, int32_t *counts, int *s_max_out)
{
    if (!n || !c || !host_out || !counts || !s_max_out || s_cap <= 0) return fail(AEV_E_ARG, "null argument");
    if (n->device != c->device) return fail(AEV_E_ARG, "the handle and the communicator live on different devices");
    RcclApi *api = nullptr;
    TRY(rccl_api(&api));
    HIPCHK(hipSetDevice(n->device));
    const Layer &l = n->L.back();
    if (num_bbox <= 0 || l.C != num_classes + 5 * num_bbox) return fail(AEV_E_SHAPE, "last layer channels != num_classes + 5*num_bbox");
    const int W = c->world;
    const size_t per_stream = (size_t)l.H * l.W * num_bbox * 7;
    // (1) the ranks' stream counts: shards are unequal whenever the global stream count is not a multiple of the world size, and a
    //     collective needs equal shapes -- every rank pads to the largest count (streams.gather_predictions does the same over gloo)
    TRY(comm_buf(c, (size_t)(W + 1) * sizeof(int)));
    int *d_cnt = (int *)c->d_buf;
    const int mine = n->S;
    HIPCHK(hipMemcpyAsync(d_cnt + W, &mine, sizeof(int), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(api, api->AllGather(d_cnt + W, d_cnt, 1, ncclInt32, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(counts, d_cnt, W * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int smax = 0;
    for (int r = 0; r < W; ++r) smax = std::max(smax, counts[r]);
    *s_max_out = smax;
    if (smax > s_cap) return fail(AEV_E_ARG, "host_out is too small for the largest rank's streams");
    // (2) this rank's detections [smax][nb][7] (padding rows zero), decoded on the handle's stream straight into the send buffer
    const size_t send = (size_t)smax * per_stream;
    TRY(comm_buf(c, (size_t)(W + 1) * send * sizeof(float)));
    float *d_send = (float *)c->d_buf, *d_recv = d_send + send;
    HIPCHK(hipMemsetAsync(d_send, 0, send * sizeof(float), n->stream));
    TRY(net_detect(n, num_classes, num_bbox, conf_thr, use_nms, iou_thr, d_send));
    HIPCHK(hipStreamSynchronize(n->stream));
    // (3) ONE all-gather: 2 KB per stream, latency-bound; every peer is one xGMI hop away
    NCCLCHK(api, api->AllGather(d_send, d_recv, send, ncclFloat32, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(host_out, d_recv, (size_t)W * send * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

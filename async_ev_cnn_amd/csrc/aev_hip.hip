// aev_hip.hip -- host runtime + C ABI (include/aev.h) of libaev_hip.so.  gfx950 only.
//
// One aev_net = one chain of stateful layers replicated over S independent event streams.  All state
// lives in HBM for the lifetime of the handle; a step is a fixed sequence of kernel launches on the
// handle's HIP stream whose geometry does not depend on the data (device-side counts), so the same
// sequence is captured once into a hipGraph and replayed per step.
#include "aev_kernels.hpp"
#include "../../include/aev.h"

#include <algorithm>
#include <hip/hip_ext.h>
#include <rccl/rccl.h> // (types and prototypes only: RCCL is bound at run time, see aev_comm.hpp)
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <utility>
#include <map>
#include <vector>

#include <dlfcn.h>
#include <mutex>

// ------------------------------------------------------------------------------------------------
#define AEV_MAX_DEVICES 64 // (bound of the per-device tables: stream pool, warm-up flags)
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? AEV_E_ALLOC : AEV_E_HIP,                                \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                \
    } while (0)

enum { KC_INTEG_PREP = 0, KC_INTEG_SWEEP, KC_LEAK_SWEEP, KC_COMPACT, KC_CONV_EXPAND, KC_CONV_VALU, KC_CONV_MFMA,
       KC_POOL_MARK, KC_POOL_LIST, KC_POOL_EVAL, KC_MISC, KC_CONV_PATCH };
static const char *g_kernel_names[AEV_NUM_KERNELS] = {
    "k_integ_prepare", "k_integ_sweep", "k_leak_sweep", "k_compact", "k_conv_expand", "k_conv_valu", "k_conv_mfma",
    "k_touch_mark", "k_touch_list", "k_pool_eval", "misc", "k_conv_patch"};

struct Layer {
    int kind = 0, C = 0, H = 0, W = 0, HW = 0, CHW = 0;
    long long sstride = 0; // per-stream stride of float maps / idx
    long long mstride = 0; // per-stream stride of the byte maps (HW rounded up to 64)
    // conv
    int kh = 0, kw = 0, cin = 0, K = 0, Krows = 0, CoutP = 0, pt = 0, pl = 0, stride = 1;
    float alpha = 0.f;
    int KC = 0, TS = 0, mfma_cfg = -1, gz = 1;
    int patch_cfg = -1, TH = 0, TW = 0, tilesX = 0, tilesY = 0, patch_gz = 1; // spatial-tile kernel (3x3 convs)
    int patch_resident = 256; // blocks of that kernel the device holds at once
    int *nevw = nullptr;                     // pool: per stream, the list entries that are windows of incoming events (k_post)
    uint8_t *dflag = nullptr;                // conv: "may decay" byte per site [S][mstride] (SweepSeg::flag); pool: the same for the copy's windows
    float *pfm = nullptr, *pact = nullptr;   // pool over a conv: copy of the conv's fm / act at the argmax sites [S][sstride] (a cache: see pool_copy_*)
    bool copy_dirty = true;                  // the copy has to be rebuilt before a tile kernel reads it
    int *tlist = nullptr, *tcount = nullptr; // tile kernel: active tiles of the step [AEV_TL_BUCKETS][S * tiles] and their numbers
    int tcap = 0;
    int tl_nb = 1;           // segments of the tile list by fill, walked largest first (> 1: layers whose tile-kernel blocks take few items each)
    u64 *tmask = nullptr;                    // ... and their 256-bit site masks [S * tiles][4]
    bool tlist_ready = false;                // the list was written by the k_post launched before this layer's recompute
    int patch_src = 0; // 0: previous layer is a 2x2/2 pool over a conv (read through the pool layer's copy), 1: previous layer is a conv
    int one_cfg = -1;  // 1x1 conv kernel (k_conv_1x1): channels per chunk, or -1
    unsigned short *wt3 = nullptr; // split-bf16 weights of the x6 tile kernel (aev_conv_x6.hpp; patch_cfg 10 / 11), else nullptr
    bool first3x3 = false; // k_conv_first3x3: 3x3 conv directly on the integration surface
    float *fm = nullptr, *act = nullptr, *init_fm = nullptr, *wt = nullptr, *bias = nullptr;
    // pool
    uint8_t *idx = nullptr, *init_idx = nullptr;
    uint8_t *pending = nullptr; // byte map [S][mstride]
    int *key = nullptr;
    // integration
    float *surf = nullptr;
    int *owner = nullptr;
    // event bookkeeping
    uint8_t *mapA = nullptr, *mapB = nullptr; // byte maps [S][mstride]: conv touch / changed; integration: events
    int *ev = nullptr, *nev = nullptr;  // output events [S][HW], [S]
    int *sites = nullptr, *nsites = nullptr; // conv: recompute list
    long long lstride = 0;
};

struct ProfRec { hipEvent_t a, b; int cls, layer; };

// The AEV_* switches (tests and diagnostics; never needed for correctness -- every setting computes the same bits), read ONCE when a
// handle is created: nothing on the launch path calls getenv, and a handle keeps its configuration for life.  (One more is read where
// a layer is added: AEV_CFG_<layer index> = instance of the spatial-tile kernel, see configure_conv_kernels.)
struct Switches {
    bool verbose = false;       // AEV_VERBOSE: kernel instance, tile and occupancy per layer; the plans' policy
    int plan_dump = 0;          // AEV_PLAN_DUMP=<n>: print the plans of the first n steps (stderr)
    int graph_steps = 32;       // AEV_GRAPH_STEPS: steps of the longest captured graph (round 4: 8 -> 32: one stream 0.174 -> 0.166 ms per step, 64 streams -1.5 %, 256 +-0)
    int overlap = -1;           // AEV_OVERLAP=0/1: the many-streams plan (second HIP stream for the sweeps) off / on whatever the stream count
    int lookahead = -1;         // AEV_LOOKAHEAD=0/1: leak sweeps a step ahead off / on (default: wherever the many-streams plan is)
    int tail_overlap = -1;      // AEV_TAIL_OVERLAP=0/1: the 1x1 tail of a step beside the next step's head (default: wherever the look-ahead is)
    int hoist_max = 1 << 30;    // AEV_HOIST_MAX: queued steps of at most so many streams may use the few-streams plan (0: never); which plan a handle gets: choose_plan
    int force_queue = 0;        // AEV_FORCE_QUEUE: 1, AEV_NO_QUEUE: -1 (work queue of the tile kernel whatever / never)
    int queue_static = 2;       // AEV_QUEUE_STATIC=<0..8>: eighths of a tile-kernel launch's items handed out by striding before the queue takes over
    int tl_segments = 0;        // AEV_TL_SEGMENTS=<1..32>: segments of the tile lists by fill (0: by policy)
    int tuning = 0;             // AEV_TUNING=0/1/2: default kernel-configuration policy (aev_net_set_tuning)
    bool decay_flags = true;    // AEV_NO_DECAY_FLAGS: the leak sweep reads every site's decay instead of skipping the flagged-zero ones
    int pool_in_post = 9216;    // AEV_POOL_IN_POST: largest pool layer (window x four-channel items per stream) k_post evaluates itself (round 4: 4608 -> 9216, pool3 of the e-fcn too: -0.35 % per 256-stream step; 18 000: +2.5 %)
    int half_max = 1;           // AEV_HALF_MAX: up to so many streams the few-streams instance of the tile kernel is 7 with 32-site tiles (see configure_conv_kernels)
    int stamps_layer = -1;      // AEV_STAMPS=<layer>: in-kernel phase stamps of that layer's tile kernel (eager runs; dumped by aev_net_sync)
    int chain_first = 4;        // AEV_CHAIN_FIRST=<mask> (default 4: the look-ahead forks; measured round 5, ms per step at 256 streams: none 1.940, AHEAD 1.918, TAIL 1.948, SWEEP 1.968, all 2.01 -- with every branch behind the chain the runtime puts the branches on ONE hardware queue): at a fork of a captured step whose branch runs on stream s (bit s: 2 SWEEP, 4 AHEAD, 8 TAIL) the chain's next launch is captured in front of the branch (chain_first)
    bool la_copy_on_tail = false; // AEV_LA_COPY_ON_TAIL=0/1: see plan_general (measured round 5: 1.925 against 1.902 ms per step, sustained 1.909 against 1.865 -- not kept)
    bool post_stamps = false;   // AEV_POST_STAMPS: per-phase ticks of k_post2's block 0, summed per layer, printed by aev_net_sync
    bool post_v1 = false;       // AEV_POST_V1: the list-walking k_post of rounds 1-4 instead of the bit-parallel k_post2 (aev_book.hpp)
    static Switches from_env()
    {
        Switches w;
        auto on = [](const char *k) { return getenv(k) != nullptr; };
        auto num = [](const char *k, int dflt) { const char *e = getenv(k); return e ? atoi(e) : dflt; };
        auto tri = [&](const char *k) { return on(k) ? (num(k, 0) != 0 ? 1 : 0) : -1; };
        w.verbose = on("AEV_VERBOSE"); w.plan_dump = num("AEV_PLAN_DUMP", 0); w.graph_steps = std::max(1, num("AEV_GRAPH_STEPS", 32));
        w.overlap = tri("AEV_OVERLAP"); w.lookahead = tri("AEV_LOOKAHEAD"); w.tail_overlap = tri("AEV_TAIL_OVERLAP");
        w.hoist_max = num("AEV_HOIST_MAX", 1 << 30);
        w.force_queue = on("AEV_FORCE_QUEUE") ? 1 : on("AEV_NO_QUEUE") ? -1 : 0;
        w.queue_static = std::min(std::max(num("AEV_QUEUE_STATIC", 2), 0), 8);
        w.tl_segments = std::min(std::max(num("AEV_TL_SEGMENTS", 0), 0), AEV_TL_BUCKETS);
        { const int t = num("AEV_TUNING", 0); w.tuning = t >= 0 && t <= 2 ? t : 0; }
        w.decay_flags = !on("AEV_NO_DECAY_FLAGS");
        w.pool_in_post = num("AEV_POOL_IN_POST", 9216);
        w.half_max = num("AEV_HALF_MAX", 1);
        w.stamps_layer = on("AEV_STAMPS") ? num("AEV_STAMPS", -1) : -1;
        w.post_v1 = on("AEV_POST_V1") && num("AEV_POST_V1", 1) != 0;
        w.post_stamps = on("AEV_POST_STAMPS");
        w.la_copy_on_tail = num("AEV_LA_COPY_ON_TAIL", 0) != 0;
        w.chain_first = num("AEV_CHAIN_FIRST", 4);
        return w;
    }
};

struct aev_net {
    Switches sw;
    int H = 0, W = 0, S = 0, device = 0, ncu = 256;
    double leak = 0;
    hipStream_t stream = nullptr;
    // Second stream: the HBM-bound leak sweeps run beside the MFMA-bound / latency-bound chain (forked after
    // k_integ_prepare, each conv's recompute joins on its own layer's sweep).  Disabled while profiling per kernel.
    hipStream_t stream2 = nullptr;
    // Third stream: the look-ahead sweeps.  Not the second stream: a stream that leaves the captured step (joined at the head) and
    // enters it again further down makes hipGraphLaunch of the instantiated graph crash inside the runtime (ROCm 7.0, seen as
    // an intermittent segmentation fault of the first replay); a stream of their own, forked and joined once, does not.
    hipStream_t stream3 = nullptr;
    // Fourth stream: the tail of a queued step (the trailing 1x1 conv layers: few blocks, tens of microseconds each) runs beside
    // the head of the next step, see launch_chain.  `cur`: the stream launch_kernel uses instead of `stream` when set.
    hipStream_t stream4 = nullptr, cur = nullptr;
    hipEvent_t ev_tail_a = nullptr;
    int tail_from = 0, tail_feed = 0;   // first layer of the tail / the conv layer whose state (or pool copy) the tail reads; 0: none
    int tail_from_few = 0, tail_feed_few = 0; // the same for the few-streams plan: a DEEP tail (from the last tile-kernel conv layer on), see tail_layers_few
    bool tail_overlap = false, tail_pending = false, tail_sweeps4 = false; // (state between two launch_chain calls of a run, see there)
    static constexpr int tail_at = 1; // the deferred 1x1 tail starts behind the k_post of this layer of the next step (beside the pool eval behind
                                      // the first conv layer and the tile kernel after it: measured best, docs/KERNEL_NOTES.md round 2)
    hipEvent_t ev_prep = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> ev_sweep; // per layer
    int overlap = 1;
    // Look-ahead of the leak sweeps (queued steps): the sweeps of the conv layers up to la_gate (+ the pool copies behind them)
    // for step t+1 run on the second stream beside the deep conv layers of step t, from the moment the recompute of layer
    // la_after (the last reader of those states in step t) has finished.  See launch_chain.
    int la_gate = 0, la_state = 0 /* see lookahead_layers */; bool lookahead = false;
    int *d_qrun = nullptr;          // [2] first step / end of the queue run being executed
    float *d_dlf_next = nullptr; int *d_active_next = nullptr;
    std::vector<hipEvent_t> ev_la_a; hipEvent_t ev_la_b = nullptr;
    std::vector<Layer> L;
    long long *d_prev_ts = nullptr, *d_last_ts = nullptr;
    double *d_dl64 = nullptr;
    float *d_dlf = nullptr;
    int *d_active = nullptr;
    int *d_yxt = nullptr; size_t yxt_cap = 0; // staging of aev_layer_step (rows)
    // aev_net_step: two pinned host + device staging slots, so the host can prepare batch t+1 while batch t runs
    struct Slot { int *h_yxt = nullptr, *d_yxt = nullptr; long long *h_off = nullptr, *d_off = nullptr; size_t cap = 0; hipEvent_t done = nullptr; } slot[2];
    int cur_slot = 0;
    long long *d_off = nullptr;               // [S+1]
    int *d_qyxt = nullptr; long long *d_qoff = nullptr; int qT = 0; int *d_qstep = nullptr;
    int *d_inj = nullptr; size_t inj_cap = 0; int *d_ninj = nullptr; // injected events for aev_layer_step
    u64 *d_stats = nullptr;                   // [64]: per layer; [62] = input events
    int *d_tcount = nullptr;                  // [64][AEV_TC_STRIDE] per layer: tile kernel list size / work queue (cleared by k_integ_prepare)
    float *d_scratch = nullptr; size_t scratch_cap = 0;
    // captured step graphs, longest first: AEV_GRAPH_STEPS (32) steps, then half as many ... down to one.  Steps chain inside a graph
    // as they do across replays (the device step counter), without the replay gap, and only inside a graph do they overlap (launch_chain)
    // A graph comes in up to four variants: `incoming` = its first step launches the tail the previous graph's last step left
    // behind (launch_chain: tail overlap), `defers` = its own last step leaves its tail to the graph that follows.  A run replays
    // [defers] [incoming, defers] ... [incoming]: every step of a run but the last overlaps with its successor.
    struct StepGraph { int steps; bool incoming, defers; hipGraph_t src; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    int conv_impl = 0;
    unsigned long long *d_stamps = nullptr; size_t stamps_cnt = 0; int stamp_items = 0, stamp_layer = -1; // AEV_STAMPS diagnostic buffer
    unsigned long long *d_pstamps = nullptr; // AEV_POST_STAMPS: [64 layers][2 bodies of a pair][16]
    // A chain with a pool layer directly behind a pool layer (the reference's constructors allow it, maxpool.py:14-40) is stepped layer by
    // layer with the kernels of the per-layer API (launch_chain_generic): no fused bookkeeping, no overlap plans -- correct, not fast.
    bool generic = false;
    bool chain_cleared = false; // inside launch_chain: k_integ_prepare has cleared d_tcount for this step
    bool integ_fused = false;   // the chain's steps start with k_integ_step (at most AEV_INTEG_FUSED_MAX events per stream and step) instead of k_integ_prepare + k_integ_sweep
    bool queue_fused_ok = false; // ... which holds for every cell of the uploaded queue
    int tuning = 0; // 0 auto (by stream count), 1 throughput configs, 2 few-streams (latency) configs
    int conv_math = 0; // AEV_MATH_F32 / AEV_MATH_BF16X6 (aev_net_set_conv_math)
    int dense = 0; // 1 / 2: dense-frame mode (every site of every layer recomputed each step, no leak extrapolation; 2: see make_view)
    long long host_steps = 0;
    bool profiling = false;
    int cur_layer = 0;
    std::vector<ProfRec> prof;
    std::vector<void *> allocs;
};

template <typename T> static int dalloc(aev_net *n, T **p, size_t count)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    n->allocs.push_back(q);
    *p = (T *)q;
    return AEV_OK;
}
#define TRY(expr) do { int rc_ = (expr); if (rc_ < 0) return rc_; } while (0)

// Per-kernel timing (aev_net_queue_profile): the launch carries its own start / stop events (hipExtLaunchKernel), which take
// the dispatch's begin / end timestamps -- the kernel's duration as rocprofv3 sees it, without the launch gap and the
// end-of-kernel cache write-back that a pair of hipEventRecord calls around the launch would add.
static hipError_t launch_kernel(aev_net *n, int cls, const void *fn, dim3 grid, dim3 block, void **args, size_t shmem)
{
    if (!n->profiling) return hipLaunchKernel(fn, grid, block, args, shmem, n->cur ? n->cur : n->stream);
    ProfRec r; r.cls = cls; r.layer = n->cur_layer;
    hipEventCreate(&r.a); hipEventCreate(&r.b);
    n->prof.push_back(r);
    return hipExtLaunchKernel(fn, grid, block, args, shmem, n->stream, r.a, r.b, 0);
}
template <typename... Params, typename... Args>
static void launch_typed(aev_net *n, int cls, void (*kern)(Params...), dim3 grid, dim3 block, size_t shmem, Args &&...args)
{
    std::tuple<Params...> vals{std::forward<Args>(args)...};
    void *ptrs[sizeof...(Params)];
    int i = 0;
    std::apply([&](auto &...v) { ((ptrs[i++] = (void *)&v), ...); }, vals);
    launch_kernel(n, cls, (const void *)kern, grid, block, ptrs, shmem);
}
#define LAUNCH(net, cls, kern, grid, block, shmem, ...) launch_typed(net, cls, kern, grid, block, shmem, __VA_ARGS__)

static int gx_cap(int ns) { return std::max(16, 4096 / std::max(ns, 1)); }
static int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// The instances of the spatial-tile kernel (aev_conv_patch.hpp): <channels per chunk, site groups, cout groups, cout blocks per wave,
// site blocks per wave (x 32 sites), half>.  See configure_conv_kernels for who gets which.
//   0  <16,4,1,1,1>  32 couts per block, 16-channel chunks: few streams (one tile spreads over many CUs), cin % 16 == 0
//   1  <4,4,1,1,1>   32 couts, 4-channel chunks: cin % 8 != 0
//   2  <8,2,2,1,2>   64 couts: 64-cout layers (conv3 of the e-fcn)
//   3  <4,2,2,1,2>   64 couts, 4-channel chunks
//   4  <8,1,4,1,5>   128 couts, one block per CU: layers of >= 128 couts (conv4, conv5)
//   5  <8,4,1,1,2>   32 couts, tiles of up to 196 sites: few streams, cin % 16 != 0
//   6  <8,2,2,1,2,half>  32 couts, the waves as (site group, 16-cout half): 32-cout layers (conv2)
//   7  <16,2,2,1,1,half> 32 couts, 16-channel chunks, tiles of up to 64 sites, the waves as (site group, 16-cout half): up to 16 streams --
//                        with tiles of 32 sites a wave's chain of MFMAs is one 16-site block x 16 couts, half of instance 0's
//   8  <4,1,4,1,4>   128 couts, 4-channel chunks, tiles of up to 128 sites: TWO blocks per CU (53 KB of LDS, <= 256 registers) -- the round-5
//                    experiment for conv4 / conv5 (one block's table / staging / scatter under the other's MFMAs); test hook only
//   9  <8,2,4,1,3>   128 couts, EIGHT waves (two site groups x four cout groups, 512 threads, one LDS image): two waves per SIMD at <= 256 registers -- the
//                    round-5 experiment for conv4 / conv5 (one wave's staging / scatter under the other's MFMAs); test hook only
//  10  x6 <2,2,5>    AEV_MATH_BF16X6 (aev_conv_x6.hpp): 128 couts per block (two site groups x two 64-cout groups), 16-channel chunks, tiles <= 160 sites
//  11  x6 <4,1,2,2>  ... 32 couts per block (four site groups), tiles <= 128 sites, TWO blocks per CU at 205 registers -- beside the look-ahead sweep's 80
//                    (three blocks need <= 168: 31 spilled, conv2 240 -> 278 us): conv2 (a single chunk: ONE LDS image) and conv3 (two cout blocks per tile;
//                    measured against <2,2,4,2>, 64 couts per block, one block per CU: conv3 279 -> 253 us, step -1.2 %)
#define AEV_NUM_PATCH_CFG 12
#define AEV_X6_CFG0 10
static const int g_patch_cc[AEV_NUM_PATCH_CFG] = {16, 4, 8, 4, 8, 8, 8, 16, 4, 8, 16, 16}, g_patch_cw[AEV_NUM_PATCH_CFG] = {32, 32, 64, 64, 128, 32, 32, 32, 128, 128, 128, 32},
                 g_patch_acap[AEV_NUM_PATCH_CFG] = {128, 128, 128, 128, 160, 256, 128, 64, 128, 192, 160, 128},
                 g_patch_nt[AEV_NUM_PATCH_CFG] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 512, 256, 256}; // 32 * site groups * site blocks per wave: most active sites per tile = the kernels' table size
static const void *patch_kernel(int cfg)
{
    switch (cfg) {
    case 0: return (const void *)k_conv_patch<16, 4, 1, 1, 1>;
    case 1: return (const void *)k_conv_patch<4, 4, 1, 1, 1>;
    case 2: return (const void *)k_conv_patch<8, 2, 2, 1, 2>;
    case 3: return (const void *)k_conv_patch<4, 2, 2, 1, 2>;
    case 4: return (const void *)k_conv_patch<8, 1, 4, 1, 5>;
    case 5: return (const void *)k_conv_patch<8, 4, 1, 1, 2>;
    case 6: return (const void *)k_conv_patch<8, 2, 2, 1, 2, true>;
    case 7: return (const void *)k_conv_patch<16, 2, 2, 1, 1, true>;
    case 8: return (const void *)k_conv_patch<4, 1, 4, 1, 4>;
    case 9: return (const void *)k_conv_patch<8, 2, 4, 1, 3>;
    case 10: return (const void *)k_conv_x6<2, 2, 5>;
    case 11: return (const void *)k_conv_x6<4, 1, 2, 2, 2>;
    }
    return nullptr;
}
// The 1x1 kernel's instances <channels per chunk, row split> and their dynamic LDS (two chunk buffers + tables).
static const void *one_kernel(int cc, int rs)
{
    if (rs) return (const void *)k_conv_1x1<32, 1>;
    return cc == 32 ? (const void *)k_conv_1x1<32, 0> : (const void *)k_conv_1x1<16, 0>;
}
static size_t one_lds_bytes(int cc, int rs)
{
    const size_t cols = rs ? 128 : AEV_COLS1, pp = cols + 1, cw = rs ? 32 : 128;
    return (2 * (2 * cc * pp + cc * cw) + 2 * pp + 2 + 2 * cols + 8) * sizeof(float);
}
static size_t patch_lds_bytes(int cfg, int th = 14, int nch = 2) // two chunk buffers (patch planes + weight rows) + two site tables (even / odd items), queue hand-over
{
    if (cfg >= AEV_X6_CFG0) return (nch > 1 ? 2 : 1) * (size_t)(th + 2) * AEV_PW * AEV_X6_SLOT + (2 * (size_t)g_patch_acap[cfg] + 16) * sizeof(int); // x6: two patch images of th + 2 rows (one: a single chunk), one site table
    return (2 * ((size_t)2 * g_patch_cc[cfg] * AEV_PPS + (size_t)g_patch_cc[cfg] * 9 * g_patch_cw[cfg]) + 4 * g_patch_acap[cfg] + 4) * sizeof(float);
}

// A pool layer seen through its COPY of the winners (Layer::pfm / pact, see pool_copy_fill): a plain conv-like map of the pool layer's
// own shape -- raw surface and raw decay of the conv layer at the bottom of the pool stack at the (composed) argmax sites, with that
// conv layer's slope.  The caller makes sure the copy is clean (pool_copies_clean).
static int base_conv_of(const aev_net *n, int li) { while (li > 0 && n->L[li].kind == AEV_KIND_POOL) --li; return li; }
static bool pool_behind_pool(const aev_net *n, int li) { return li >= 2 && n->L[li].kind == AEV_KIND_POOL && n->L[li - 1].kind == AEV_KIND_POOL; }
static int copy_view(const aev_net *n, int li, SrcView *v)
{
    const Layer &l = n->L[li];
    if (l.kind != AEV_KIND_POOL || !l.pfm) return fail(AEV_E_UNSUPPORTED, "this pool layer keeps no copy of its winners");
    memset(v, 0, sizeof(*v));
    v->kind = AEV_KIND_CONV; v->bkind = AEV_KIND_CONV; v->C = l.C; v->H = l.H; v->W = l.W; v->bH = l.H; v->bW = l.W;
    v->s = l.pfm; v->a = l.pact; v->bstride = l.sstride; v->alpha = n->L[base_conv_of(n, li)].alpha;
    return AEV_OK;
}

// The layer as a source for its consumer (and for the dense getters).
static int make_view(const aev_net *n, int li, SrcView *v)
{
    const Layer &l = n->L[li];
    // a pool layer directly behind a pool layer (MaxPoolLayer takes any prev_layer, maxpool.py:14-40): two levels of argmax bytes are
    // not chased anywhere -- it is read through its copy, as the tile kernels read every pool layer
    if (pool_behind_pool(n, li)) return copy_view(n, li, v);
    memset(v, 0, sizeof(*v));
    v->kind = l.kind; v->C = l.C; v->H = l.H; v->W = l.W;
    const Layer *b = &l;
    if (l.kind == AEV_KIND_POOL) {
        b = &n->L[li - 1];
        v->idx = l.idx; v->istride = l.sstride; v->pkh = l.kh; v->pkw = l.kw; v->pstride = l.stride;
    }
    v->bkind = b->kind; v->bH = b->H; v->bW = b->W; v->bstride = b->sstride;
    if (b->kind == AEV_KIND_INTEG) { v->s = b->surf; v->a = nullptr; v->alpha = 0.f; }
    else { v->s = b->fm; v->a = b->act; v->alpha = b->alpha; }
    // Dense-frame mode 2 (the reference's dense NumPy model, frame_numpy.py:77-78): the activation is applied AGAIN to
    // a pool layer's output, i.e. a non-positive pooled value carries the slope twice.  One product here instead of two in the
    // reference (alpha*alpha rounded once): identical for a dyadic alpha, within 1 ulp otherwise.
    if (n->dense == 2 && l.kind == AEV_KIND_POOL && b->kind == AEV_KIND_CONV) v->alpha = b->alpha * b->alpha;
    return AEV_OK;
}

// ------------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------------
static int launch_compact(aev_net *n, Layer &l, bool with_sites, int s0, int ns, int li)
{
    CompactArgs c{}; // (kernel arguments are value-initialised everywhere: a field nobody sets is a null pointer, not a wild one)
    c.mapA = l.mapA; c.mapB = l.mapB; c.mstride = l.mstride; c.HW = l.HW; c.W = l.W; c.s0 = s0;
    n->cur_layer = li;
    c.listA = with_sites ? l.sites : nullptr; c.nA = with_sites ? l.nsites : nullptr;
    c.listAB = l.ev; c.nAB = l.nev; c.lstride = l.lstride; c.stat = n->d_stats + li;
    LAUNCH(n, KC_COMPACT, k_compact, dim3(ns), dim3(1024), 0, c);
    return AEV_OK;
}

static int launch_integ(aev_net *n, const int *yxt, const long long *off, const int *qstep, int s0, int ns)
{
    Layer &l = n->L[0];
    IntegArgs a{};
    a.yxt = yxt; a.off = off; a.qstep = qstep; a.S = n->S; a.s0 = s0;
    a.H = l.H; a.W = l.W; a.HW = l.HW; a.leak = n->leak;
    n->cur_layer = 0;
    a.surf = l.surf; a.sstride = l.sstride; a.owner = l.owner;
    a.prev_ts = n->d_prev_ts; a.last_ts = n->d_last_ts; a.dl64 = n->d_dl64; a.dlf = n->d_dlf; a.active = n->d_active;
    a.evmap = l.mapA; a.mstride = l.mstride; a.stat_events = n->d_stats + 62;
    a.zero = nullptr; a.nzero = 0;
    a.qrun = nullptr; a.dlf_next = nullptr; a.active_next = nullptr;
    LAUNCH(n, KC_INTEG_PREP, k_integ_prepare, dim3(ns), dim3(256), 0, a);
    LAUNCH(n, KC_INTEG_SWEEP, k_integ_sweep, dim3(cdiv(l.HW, 1024), ns), dim3(256), 0, a); // (four pixels per thread)
    return launch_compact(n, l, false, s0, ns, 0);
}

// The pool layers' copies of their inputs (Layer::pfm / pact): a cache of the conv state below at the argmax sites, so that
// the tile kernels gather a dense [H/2, W/2, C] map instead of chasing argmax bytes into a map four times the size (PMC:
// 1.3 GB -> a third per conv2 launch).  The chain keeps it exact incrementally -- the leak sweep extrapolates the copy with
// the same two roundings on the same operands, k_pool_eval rewrites every window it evaluates, and every conv site that
// changes otherwise is an event, hence an evaluated window -- whatever else touches a conv state or an argmax (per-layer
// API, reset) marks it dirty, and it is rebuilt from the argmax bytes before the next reader (never inside a graph capture).
static int pool_copy_fill(aev_net *n, int li, int s0, int ns)
{
    Layer &l = n->L[li];
    const Layer &b = n->L[li - 1];
    PoolFillArgs f{};
    // (a pool layer behind a pool layer: from that layer's copy, rebuilt before this one -- pool_copies_clean goes up the chain)
    f.s = b.kind == AEV_KIND_POOL ? b.pfm : b.fm; f.a = b.kind == AEV_KIND_POOL ? b.pact : b.act;
    f.bstride = b.sstride; f.bW = b.W; f.idx = l.idx; f.pfm = l.pfm; f.pact = l.pact; f.istride = l.sstride;
    f.C = l.C; f.H = l.H; f.W = l.W; f.kh = l.kh; f.kw = l.kw; f.stride = l.stride; f.s0 = s0;
    hipLaunchKernelGGL(k_pool_fill, dim3(std::min(cdiv(l.CHW, 256), 256), ns), dim3(256), 0, n->stream, f);
    if (l.dflag) HIPCHK(hipMemsetAsync(l.dflag + (size_t)s0 * l.mstride, 1, (size_t)ns * l.mstride, n->stream)); // conservative: the sweep un-flags
    return AEV_OK;
}
static int pool_copies_clean(aev_net *n) // before a chain step / a capture
{
    for (int li = 1; li < (int)n->L.size(); ++li)
        if (n->L[li].pfm && n->L[li].copy_dirty) { TRY(pool_copy_fill(n, li, 0, n->S)); n->L[li].copy_dirty = false; }
    return AEV_OK;
}
static void pool_copy_touch(aev_net *n, int li) // layer li (conv or pool) changed outside the chain
{
    // (the copies of the whole pool stack above the layer: each is built from the one below)
    for (int lj = n->L[li].kind == AEV_KIND_POOL ? li : li + 1; lj < (int)n->L.size() && n->L[lj].kind == AEV_KIND_POOL; ++lj) n->L[lj].copy_dirty = true;
}

// the copies of the pool stack that ends at layer li, bottom up (each is built from the one below); never inside a capture
static int pool_stack_clean(aev_net *n, int li)
{
    for (int lj = base_conv_of(n, li) + 1; lj <= li; ++lj)
        if (n->L[lj].pfm && n->L[lj].copy_dirty) { TRY(pool_copy_fill(n, lj, 0, n->S)); n->L[lj].copy_dirty = false; }
    return AEV_OK;
}

static int ilog2(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }
// mode 0: the plain sweep of layers lo..hi for this step; 1: the same, but skipped on the device when the previous step of the
// run has already extrapolated these layers (look-ahead); 2: the look-ahead sweep itself, for the NEXT queue step
// parts: bit 0 = the conv layers' own state, bit 1 = the pool copies behind them (with_copies); hi_parts: the same for layer hi alone (0: as the others)
static int launch_leak_sweep(aev_net *n, int lo, int hi, int s0, int ns, hipStream_t on = nullptr, bool with_copies = false, int mode = 0, int la_bpc = 1, int parts = 3, int hi_parts = 0,
                             const IntegArgs *head = nullptr) // head: k_integ_step's arguments -- both in one launch (k_head)
{
    const bool sweep_flags = n->sw.decay_flags && !n->generic; // (the generic chain's k_compact does not flag the recomputed sites)
    hipStream_t keep = n->stream;
    if (on) n->stream = on; // LAUNCH uses n->stream
    SweepArgs a{};
    // 2: the next step's sweep beside this step's deep layers (k_leak_sweep_ahead: few persistent blocks); 3: the first step's, before its
    // run, from k_integ_peek's values -- alone on the device: the plain kernel, every segment in one launch
    const bool ahead = mode == 2;
    a.nseg = 0; a.s0 = s0; a.dlf = mode >= 2 ? n->d_dlf_next : n->d_dlf; a.active = mode >= 2 ? n->d_active_next : n->d_active;
    a.qstep = n->d_qstep; a.qrun = n->d_qrun; a.prologue = mode == 3; a.chunks = 0; a.ns = ns;
    if (ahead) { // every segment in one launch of few persistent blocks
        int segs = 0;
        for (int li = lo; li <= hi; ++li)
            if (n->L[li].kind == AEV_KIND_CONV) segs += 1 + (with_copies && li + 1 < (int)n->L.size() && n->L[li + 1].pfm ? 1 : 0);
        if (segs > 2) return fail(AEV_E_UNSUPPORTED, "a look-ahead sweep takes one conv layer and the pool copy behind it");
    }
    n->cur_layer = lo == hi ? lo : 0;
    int chunks = 0;
    const int csz = ahead ? AEV_SWEEP_CHUNK_LA : AEV_SWEEP_CHUNK;
    for (int li = lo; li <= hi; ++li) {
        Layer &l = n->L[li];
        if (l.kind != AEV_KIND_CONV) continue;
        if (a.nseg == AEV_MAX_SEGS) { // flush
            if (head) return fail(AEV_E_HIP, "internal: a head sweep of more than AEV_MAX_SEGS segments");
            LAUNCH(n, KC_LEAK_SWEEP, k_leak_sweep, dim3(chunks, ns), dim3(256), 0, a);
            a.nseg = 0; chunks = 0;
        }
        if (li == hi && hi_parts) parts = hi_parts;
        if (parts & 1) {
        SweepSeg &g = a.seg[a.nseg++];
        g.fm = l.fm; g.act = l.act; g.changed = l.mapB; g.sstride = l.sstride; g.mstride = l.mstride; g.CHW = l.CHW;
        g.C = l.C; g.chunk0 = chunks;
        g.flag = sweep_flags ? l.dflag : nullptr; g.fstride = l.mstride; g.cshift = ilog2(l.C);
        chunks += cdiv(l.CHW, csz);
        }
        // the copies of the pool stack behind the conv layer leak with their source
        for (int lp = li + 1; (parts & 2) && with_copies && lp < (int)n->L.size() && n->L[lp].kind == AEV_KIND_POOL && n->L[lp].pfm && !n->L[lp].copy_dirty; ++lp) {
            const Layer &pl = n->L[lp];
            if (a.nseg == AEV_MAX_SEGS) {
                if (head) return fail(AEV_E_HIP, "internal: a head sweep of more than AEV_MAX_SEGS segments");
                LAUNCH(n, KC_LEAK_SWEEP, k_leak_sweep, dim3(chunks, ns), dim3(256), 0, a);
                a.nseg = 0; chunks = 0;
            }
            SweepSeg &h = a.seg[a.nseg++];
            h.fm = pl.pfm; h.act = pl.pact; h.changed = nullptr; h.sstride = pl.sstride; h.mstride = 0; h.CHW = pl.CHW;
            h.C = pl.C; h.chunk0 = chunks;
            h.flag = sweep_flags ? pl.dflag : nullptr; h.fstride = pl.mstride; h.cshift = ilog2(pl.C);
            chunks += cdiv(pl.CHW, csz);
        }
    }
    if (head) { // (the caller has made sure of one launch: at most AEV_MAX_SEGS segments, not the look-ahead kernel)
        const int nib = cdiv(head->HW, 1024);
        n->cur_layer = 0;
        LAUNCH(n, KC_INTEG_SWEEP, k_head, dim3(nib + chunks, ns), dim3(256), 0, *head, a, nib);
    } else if (a.nseg && ahead) {
        const int ncu = n->ncu;
        a.chunks = chunks;
        // beside the tile kernels: la_bpc blocks per CU; alone on the device (no second stream: per-kernel timing): enough to stream
        const int bpc = on ? la_bpc : 8;
        LAUNCH(n, KC_LEAK_SWEEP, k_leak_sweep_ahead, dim3(std::min(ncu * bpc, chunks * ns)), dim3(256), 0, a);
    } else if (a.nseg)
        LAUNCH(n, KC_LEAK_SWEEP, k_leak_sweep, dim3(chunks, ns), dim3(256), 0, a);
    n->stream = keep;
    return AEV_OK;
}

// AEV_STAMPS=<layer> diagnostic: the stamp buffer is allocated here, by the entry points, before any launch or capture
// (hipMalloc is not allowed on a capturing thread); stamped runs are eager.
static bool stamps_wanted(const aev_net *n) { return n->sw.stamps_layer >= 0; }
static void drop_graph(aev_net *n);
static int ensure_stamps(aev_net *n)
{
    if (n->sw.post_stamps && !n->d_pstamps) {
        HIPCHK(hipMalloc((void **)&n->d_pstamps, 64 * 16 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(n->d_pstamps, 0, 64 * 16 * sizeof(unsigned long long), n->stream));
        drop_graph(n); // (captured launches carry the null pointer)
    }
    if (!stamps_wanted(n) || n->d_stamps) return AEV_OK;
    const int li = n->sw.stamps_layer;
    if (li <= 0 || li >= (int)n->L.size() || n->L[li].patch_cfg < 0) return AEV_OK; // not a tile-kernel layer: nothing to stamp
    const Layer &l = n->L[li];
    n->stamp_items = l.tilesX * l.tilesY * n->S * l.patch_gz;
    n->stamps_cnt = (size_t)n->stamp_items * 8 + (size_t)l.patch_resident * 4;
    HIPCHK(hipMalloc((void **)&n->d_stamps, n->stamps_cnt * sizeof(unsigned long long)));
    n->stamp_layer = li;
    return AEV_OK;
}

static int launch_conv_recompute(aev_net *n, int li, int s0, int ns)
{
    Layer &l = n->L[li];
    n->cur_layer = li;
    ConvArgs c{};
    if (pool_behind_pool(n, li - 1)) TRY(pool_stack_clean(n, li - 1)); // (read through its copy, see make_view)
    TRY(make_view(n, li - 1, &c.src));
    c.fm = l.fm; c.act = l.act; c.sstride = l.sstride; c.wt = l.wt; c.bias = l.bias;
    c.C = l.C; c.CoutP = l.CoutP; c.H = l.H; c.W = l.W; c.HW = l.HW; c.K = l.K; c.kh = l.kh; c.kw = l.kw;
    c.pt = l.pt; c.pl = l.pl; c.KC = l.KC; c.TS = l.TS;
    c.sites = l.sites; c.nsites = l.nsites; c.lstride = l.lstride; c.s0 = s0;
    const bool mfma = l.mfma_cfg >= 0 && n->conv_impl != 1;
    // the tile kernels read a pool layer through its copy of the conv below (a plain conv-like source), see pool_copy_fill
    SrcView tv = c.src;
    if ((l.one_cfg >= 0 || l.patch_cfg >= 0) && n->conv_impl == 0 && l.patch_src == 0) {
        Layer &pl = n->L[li - 1];
        TRY(pool_stack_clean(n, li - 1)); // (per-layer API / state init; the chain cleans up before it starts, outside any capture)
        tv.kind = AEV_KIND_CONV; tv.bkind = AEV_KIND_CONV; tv.bH = pl.H; tv.bW = pl.W;
        tv.s = pl.pfm; tv.a = pl.pact; tv.bstride = pl.sstride; tv.idx = nullptr; tv.istride = 0;
    }
    if (l.one_cfg >= 0 && n->conv_impl == 0) {
        OneArgs o{};
        o.src = tv; o.fm = l.fm; o.act = l.act; o.sstride = l.sstride; o.wt = l.wt; o.bias = l.bias;
        o.C = l.C; o.CoutP = l.CoutP; o.HW = l.HW; o.W = l.W; o.cin = l.cin; o.S = n->S;
        o.touch = l.mapA; o.mstride = l.mstride; o.s0 = s0; o.ns = ns;
        // Row split (128 columns x 32 couts per block, a wave per site block; see OneLds): the latency shape, for launches
        // that cannot give every CU a block of the other shape.  AEV_ROW_SPLIT=0/1 overrides.
        const long long blocks128 = (long long)cdiv((long long)ns * l.HW, AEV_COLS1) * (l.CoutP / 128);
        // (a tail that runs beside the next step's head, n->cur set: the row-split blocks -- 76 KB of LDS, two per CU, 96 registers
        //  -- find room beside the persistent blocks of the tile kernels, the others -- 115 KB -- wait for a whole CU: -0.5 % per step)
        const bool rs = l.one_cfg == 32 && (n->cur != nullptr || n->tuning == 2 || (n->tuning == 0 && blocks128 < 64));
        int G;
        if (rs) { // columns packed over up to 32 streams (offsets stay in 32 bits)
            const long long span = std::max(std::max(l.sstride, tv.bstride), std::max(tv.istride, (long long)1));
            G = (int)std::max<long long>(1, std::min<long long>(std::min(32, ns), (long long)INT_MAX / span));
        } else {
            // streams per group: fill the 160 MFMA columns of a tile, but keep >= ~256 blocks in the launch
            const int gmax = std::max(1, AEV_COLS1 / l.HW);
            G = std::max(1, std::min(gmax, (int)((long long)ns * (l.CoutP / 128) / 256)));
            // few streams: tiles of <= 64 columns, 32 channels per chunk (half the chunk hand-overs on the critical path)
            if (n->tuning == 1) G = gmax;            // throughput configs regardless of the stream count
            else if (n->tuning == 2) G = std::max(1, std::min(gmax, 64 / std::max(l.HW, 1)));
        }
        o.G = G;
        const int cc1 = l.one_cfg;
        const dim3 grid(cdiv((long long)G * l.HW, rs ? 128 : AEV_COLS1), cdiv(ns, G), l.CoutP / (rs ? 32 : 128));
        void *kargs[] = {&o};
        launch_kernel(n, KC_CONV_PATCH, one_kernel(cc1, rs), grid, dim3(256), kargs, one_lds_bytes(cc1, rs));
    } else if (l.patch_cfg >= 0 && n->conv_impl == 0) {
        PatchArgs q{};
        q.src = tv; q.fm = l.fm; q.act = l.act; q.sstride = l.sstride; q.wt = l.wt; q.bias = l.bias;
        q.C = l.C; q.CoutP = l.CoutP; q.H = l.H; q.W = l.W; q.HW = l.HW; q.cin = l.cin; q.pt = l.pt; q.pl = l.pl;
        q.TH = l.TH; q.TW = l.TW; q.tilesX = l.tilesX; q.s0 = s0;
        q.stamps = nullptr; q.stamp_items = 0;
        if (n->d_stamps && n->stamp_layer == li) { // diagnostic (AEV_STAMPS=<layer>): in-kernel phase stamps, dumped by aev_net_sync
            HIPCHK(hipMemsetAsync(n->d_stamps, 0, n->stamps_cnt * sizeof(unsigned long long), n->stream));
            q.stamp_items = n->stamp_items;
            q.stamps = n->d_stamps;
        }
        q.tiles = l.tilesX * l.tilesY; q.gz = l.patch_gz; q.tlist = l.tlist; q.tcount = l.tcount; q.tmask = l.tmask; q.cap = l.tcap; q.nb = l.tl_nb;
        q.force_queue = n->sw.force_queue;
        q.static_eighths = n->sw.queue_static;
        const bool x6 = l.patch_cfg >= AEV_X6_CFG0;
        { // the layer's active tiles (the touch map is complete at this point, whoever wrote it)
            TileListArgs ta{};
            ta.touch = l.mapA; ta.mstride = l.mstride; ta.H = l.H; ta.W = l.W; ta.TH = l.TH; ta.TW = l.TW; ta.tilesX = l.tilesX;
            ta.tiles = q.tiles; ta.s0 = s0; ta.list = l.tlist; ta.mask = l.tmask; ta.count = l.tcount; ta.cap = l.tcap; ta.nb = l.tl_nb;
            if (!l.tlist_ready) { // (k_post lists the tiles while it marks the touch map; everyone else leaves it to this scan)
                HIPCHK(hipMemsetAsync(l.tcount, 0, AEV_TC_STRIDE * sizeof(int), n->stream));
                LAUNCH(n, KC_MISC, k_tile_list, dim3(ns), dim3(256), (size_t)q.tiles * 4 * sizeof(u64), ta);
            }
            l.tlist_ready = false;
        }
        const dim3 grid(std::min(q.tiles * q.gz * ns, l.patch_resident)); // persistent blocks: what the chip holds at once
        X6Args xq{};
        xq.p = q; xq.w3 = l.wt3; xq.nch = l.cin / 16; xq.slots = (l.TH + 2) * AEV_PW;
        void *kargs[] = {x6 ? (void *)&xq : (void *)&q};
        launch_kernel(n, KC_CONV_PATCH, patch_kernel(l.patch_cfg), grid, dim3(g_patch_nt[l.patch_cfg]), kargs, patch_lds_bytes(l.patch_cfg, l.TH, l.cin / 16));
    } else if (l.first3x3 && n->conv_impl == 0) {
        const int gx = std::min(cdiv((long long)l.HW * std::max(l.C / 4, 1), 256), 4 * gx_cap(ns)); // a thread per (site, four channels)
        LAUNCH(n, KC_CONV_VALU, k_conv_first3x3, dim3(gx, ns), dim3(256), (size_t)10 * l.C * sizeof(float), c);
    } else if (!mfma) {
        const size_t sh = ((size_t)2 * l.K * l.TS + l.TS) * sizeof(float);
        const int gx = std::min(cdiv(l.HW, l.TS), gx_cap(ns));
        LAUNCH(n, KC_CONV_VALU, k_conv_valu, dim3(gx, ns), dim3(256), sh, c);
    } else {
        static const int nsw[4] = {4, 2, 1, 1};
        const int tsb = 32 * nsw[l.mfma_cfg];
        const size_t sh = ((size_t)2 * l.KC * tsb + tsb + l.KC) * sizeof(float);
        const int gx = std::min(cdiv(l.HW, tsb), gx_cap(ns));
        const dim3 grid(gx, ns, l.gz);
        switch (l.mfma_cfg) {
        case 0: LAUNCH(n, KC_CONV_MFMA, (k_conv_mfma<4, 1, 1>), grid, dim3(256), sh, c); break;
        case 1: LAUNCH(n, KC_CONV_MFMA, (k_conv_mfma<2, 2, 1>), grid, dim3(256), sh, c); break;
        case 2: LAUNCH(n, KC_CONV_MFMA, (k_conv_mfma<1, 4, 1>), grid, dim3(256), sh, c); break;
        default: LAUNCH(n, KC_CONV_MFMA, (k_conv_mfma<1, 4, 2>), grid, dim3(256), sh, c); break;
        }
    }
    return AEV_OK;
}

// Conv2DLayer.compute minus the leak sweep (launched separately so that all layers share one sweep).
static int launch_conv(aev_net *n, int li, const int *in_ev, const int *in_n, long long in_stride, int max_in,
                       int s0, int ns)
{
    Layer &l = n->L[li];
    ExpandArgs e{};
    e.in_ev = in_ev; e.in_n = in_n; e.in_stride = in_stride; e.kh = l.kh; e.kw = l.kw; e.pt = l.pt; e.pl = l.pl;
    e.H = l.H; e.W = l.W; e.s0 = s0; e.touch = l.mapA; e.mstride = l.mstride;
    n->cur_layer = li;
    const int gx = std::min(cdiv((long long)max_in * l.kh * l.kw, 256), gx_cap(ns));
    LAUNCH(n, KC_CONV_EXPAND, k_conv_expand, dim3(std::max(gx, 1), ns), dim3(256), 0, e);
    if ((l.patch_cfg >= 0 || l.one_cfg >= 0) && n->conv_impl == 0) { // the tile kernels read the touch map themselves; compaction clears it afterwards
        TRY(launch_conv_recompute(n, li, s0, ns));
        return launch_compact(n, l, true, s0, ns, li);
    }
    TRY(launch_compact(n, l, true, s0, ns, li));
    return launch_conv_recompute(n, li, s0, ns);
}

static TouchGeom pool_geom(const Layer &l, const Layer &p)
{
    TouchGeom g; g.mode = 1; g.kh = l.kh; g.kw = l.kw; g.oh = l.H; g.ow = l.W; g.ih = p.H; g.iw = p.W;
    return g;
}

static int fill_pool_args(aev_net *n, int li, int plain, int s0, PoolArgs &p)
{
    Layer &l = n->L[li];
    if (n->L[li - 1].kind == AEV_KIND_POOL) { // a pool layer behind a pool layer reads that layer through its copy (a plain conv-like source)
        TRY(pool_stack_clean(n, li - 1)); // (per-layer API; the chain has cleaned up before it started)
        TRY(copy_view(n, li - 1, &p.src));
    } else
        TRY(make_view(n, li - 1, &p.src));
    p.C = l.C; p.H = l.H; p.W = l.W; p.kh = l.kh; p.kw = l.kw; p.stride = l.stride; p.s0 = s0;
    p.idx = l.idx; p.istride = l.sstride; p.pending = l.pending; p.mstride = l.mstride; p.list = l.ev; p.nlist = l.nev;
    p.lstride = l.lstride;
    p.plain = plain;
    p.pfm = l.pfm; p.pact = l.pact; p.pflag = l.dflag;
    p.nevw = (n->chain_cleared && !plain) ? l.nevw : nullptr; // k_post wrote it in this step (chain only)
    return AEV_OK;
}
// A pool layer small enough for k_post to evaluate inside its one block per stream (2x2 / stride-2 windows over a conv layer,
// C % 4 == 0, at most a few (window, four channels) items per thread)
static bool pool_eval_in_post(const aev_net *n, int li)
{
    const int cap = n->sw.pool_in_post;
    const Layer &l = n->L[li];
    return l.kind == AEV_KIND_POOL && l.kh == 2 && l.kw == 2 && l.stride == 2 && n->L[li - 1].kind == AEV_KIND_CONV && l.C % 4 == 0 &&
           (long long)l.HW * (l.C / 4) <= cap;
}
static int launch_pool_eval(aev_net *n, int li, int plain, int s0, int ns)
{
    Layer &l = n->L[li];
    PoolArgs p{};
    TRY(fill_pool_args(n, li, plain, s0, p));
    n->cur_layer = li;
    const int gx = std::min(cdiv(l.CHW, 256), gx_cap(ns));
    const bool fast = l.kh == 2 && l.kw == 2 && l.stride == 2 && p.src.kind == AEV_KIND_CONV;
    if (fast && l.C % 4 == 0) LAUNCH(n, KC_POOL_EVAL, k_pool_eval_vec4, dim3(std::min(cdiv(l.CHW / 4, 256), 4 * gx_cap(ns)), ns), dim3(256), 0, p);
    else if (fast) LAUNCH(n, KC_POOL_EVAL, (k_pool_eval<true>), dim3(gx, ns), dim3(256), 0, p);
    else LAUNCH(n, KC_POOL_EVAL, (k_pool_eval<false>), dim3(gx, ns), dim3(256), 0, p);
    return AEV_OK;
}

// MaxPoolLayer.compute
static int launch_pool(aev_net *n, int li, const int *in_ev, const int *in_n, long long in_stride, int max_in, int s0,
                       int ns, const int *fused_active)
{
    Layer &l = n->L[li];
    MarkArgs m{};
    m.g = pool_geom(l, n->L[li - 1]); m.in_ev = in_ev; m.in_n = in_n; m.in_stride = in_stride; m.key = l.key;
    m.pending = l.pending; m.mstride = l.mstride; m.s0 = s0;
    n->cur_layer = li;
    const int gx = std::min(cdiv(max_in, 256), gx_cap(ns));
    LAUNCH(n, KC_POOL_MARK, k_touch_mark, dim3(std::max(gx, 1), ns), dim3(256), 0, m);
    ListArgs q{};
    q.g = m.g; q.in_ev = in_ev; q.in_n = in_n; q.in_stride = in_stride; q.key = l.key; q.pending = l.pending;
    q.mstride = l.mstride; q.s0 = s0; q.list = l.ev; q.nlist = l.nev; q.lstride = l.lstride; q.stat = n->d_stats + li;
    q.active = fused_active;
    LAUNCH(n, KC_POOL_LIST, k_touch_list, dim3(ns), dim3(1024), 0, q);
    return launch_pool_eval(n, li, 0, s0, ns);
}

static bool conv_needs_list(const aev_net *n, const Layer &l)
{
    return !((l.patch_cfg >= 0 || l.one_cfg >= 0) && n->conv_impl == 0);
}

// k_post of layer li (integration or conv): its event list, the following pool's window list, the consuming conv's
// touch map / site list -- one launch.
// (advance: the queued step ends with this launch -- it also moves the device step counter on, see chain_advances)
static int fill_post_args(aev_net *n, int li, int s0, int *advance, bool all_streams, bool allow_inline, PostArgs &a, size_t &shmem)
{
    Layer &l = n->L[li];
    const int L = (int)n->L.size();
    memset(&a, 0, sizeof(a));
    shmem = 0;
    a.s0 = s0; a.active = all_streams ? nullptr : n->d_active; // (all_streams: a stream without events has empty maps -- same lists, see k_post)
    a.advance = advance; a.advance_stat = n->d_stats + 63;
    if (li == 0 && n->integ_fused && n->chain_cleared) { a.prev_ts = n->d_prev_ts; a.last_ts = n->d_last_ts; } // (behind k_integ_step: inside emit_step only)
    a.mapA = l.mapA; a.mapB = l.mapB; a.mstride = l.mstride; a.HW = l.HW; a.W = l.W;
    a.ev = l.ev; a.nev = l.nev; a.lstride = l.lstride; a.statA = n->d_stats + li; a.count_touch = l.kind == AEV_KIND_CONV;
    a.dflag = l.kind == AEV_KIND_CONV ? l.dflag : nullptr;
    int consumer = li + 1;
    if (li + 1 < L && n->L[li + 1].kind == AEV_KIND_POOL) {
        Layer &p = n->L[li + 1];
        a.has_pool = 1; a.pg = pool_geom(p, l); a.pkey = p.key; a.ppending = p.pending; a.pmstride = p.mstride;
        a.plist = p.ev; a.pnlist = p.nev; a.pnevw = p.nevw; a.plstride = p.lstride; a.statP = n->d_stats + li + 1;
        consumer = li + 2;
        if (pool_eval_in_post(n, li + 1) && !n->dense && allow_inline) { // stage D: the block evaluates the windows itself
            TRY(fill_pool_args(n, li + 1, 0, s0, a.pe));
            a.pe_inline = 1;
        }
    }
    if (consumer < L && n->L[consumer].kind == AEV_KIND_CONV) {
        Layer &c = n->L[consumer];
        a.has_next = 1; a.n_kh = c.kh; a.n_kw = c.kw; a.n_pt = c.pt; a.n_pl = c.pl; a.n_H = c.H; a.n_W = c.W;
        a.n_touch = c.mapA; a.n_mstride = c.mstride;
        if (conv_needs_list(n, c)) { a.n_sites = c.sites; a.n_nsites = c.nsites; a.n_lstride = c.lstride; }
        if (c.patch_cfg >= 0 && n->conv_impl == 0 && !n->dense && s0 == 0) { // s0 == 0: list entries are launch-local stream indices
            a.n_tlist = c.tlist; a.n_tmask = c.tmask; a.n_tcount = c.tcount; a.n_TH = c.TH; a.n_TW = c.TW; a.n_tilesX = c.tilesX; a.n_tiles = c.tilesX * c.tilesY;
            a.n_cap = c.tcap; a.n_nb = c.tl_nb;
            shmem = (size_t)a.n_tiles * 4 * sizeof(u64);
            if (!n->chain_cleared) HIPCHK(hipMemsetAsync(c.tcount, 0, AEV_TC_STRIDE * sizeof(int), n->stream));
            c.tlist_ready = true;
        }
    }
    { // LDS tables of the block (PostArgs::l_*): whatever fits in 128 KB beside the tile masks, the key map first
        a.l_mask = (int)(shmem / sizeof(u64));
        size_t left = (128 * 1024 - shmem) / sizeof(int);
        if (a.has_pool) {
            const size_t nw = (size_t)a.pg.oh * a.pg.ow;
            if (nw <= left) { a.l_key = (int)nw; left -= nw; }
            if (a.has_next) { a.l_pl = (int)std::min(nw, left); left -= a.l_pl; }
        }
        if (a.has_pool || a.has_next) a.l_ev = (int)std::min<size_t>(std::min<size_t>(l.HW, 8192), left);
        shmem += ((size_t)a.l_key + a.l_pl + a.l_ev) * sizeof(int);
    }
    return AEV_OK;
}
// The LDS layout of k_post2 (aev_book.hpp) for this launch -- bit rows of the layer's map, of the pool layer's maps, of the consumer's
// map, the consumer's tile masks, a mirror of the window list -- or ok = 0: the launch takes k_post (pool windows other than 2x2, a
// kernel window beyond 16 x 64, maps beyond the 48 KB budget; AEV_POST_V1).
static void post2_layout(const aev_net *n, const PostArgs &a, Post2Lay &ly, size_t &shmem)
{
    memset(&ly, 0, sizeof(ly));
    shmem = 0;
    if (n->sw.post_v1) return;
    if (a.has_pool && !(a.pg.mode == 1 && a.pg.kh == 2 && a.pg.kw == 2)) return;
    if (a.has_next && (a.n_kh > 16 || a.n_kw > 64)) return;
    auto ww = [](int w) { return (w + 63) / 64; };
    const int H = a.HW / a.W;
    long long off = 0;
    ly.WW = ww(a.W); ly.o_E = (int)off; off += (long long)H * ly.WW;
    ly.sH = H; ly.sW = a.W; ly.SWW = ly.WW;
    if (a.has_pool) {
        ly.PWW = ww(a.pg.ow);
        const long long nw = (long long)a.pg.oh * ly.PWW;
        ly.o_top = (int)off; off += nw; ly.o_bot = (int)off; off += nw; ly.o_pend = (int)off; off += nw; ly.o_src = (int)off; off += nw;
        ly.sH = a.pg.oh; ly.sW = a.pg.ow; ly.SWW = ly.PWW;
    }
    if (a.has_next) {
        ly.TWW = ww(a.n_W); ly.o_T = (int)off; off += (long long)a.n_H * ly.TWW;
        if (a.n_tlist) { ly.o_tmask = (int)off; off += (long long)a.n_tiles * 4; }
    }
    long long pre_n = (long long)H * ly.WW; // the list emissions' per-word offset table: the longest word sequence
    if (a.has_pool) pre_n = std::max(pre_n, 2ll * a.pg.oh * ly.PWW);
    if (a.has_next && a.n_sites) pre_n = std::max(pre_n, (long long)a.n_H * ly.TWW);
    if (off * 8 + pre_n * 4 > AEV_POST2_LDS_MAX) return;
    ly.words = (int)off;
    ly.o_pre = (int)off * 2;
    off += (pre_n + 1) / 2;
    size_t bytes = (size_t)off * 8;
    if (a.has_pool && a.pe_inline) { // the window list the block evaluates itself: mirrored as far as the budget goes
        ly.o_plm = (int)off * 2;
        ly.plm_cap = (int)std::min<size_t>((size_t)a.pg.oh * a.pg.ow, (AEV_POST2_LDS_MAX - bytes) / sizeof(int));
        bytes += (size_t)ly.plm_cap * sizeof(int);
    }
    ly.ok = 1;
    shmem = bytes;
    if (n->d_pstamps) ly.dbg = n->d_pstamps + (size_t)std::min(n->cur_layer, 63) * 16;
}
static int launch_post(aev_net *n, int li, int s0, int ns, int *advance = nullptr, bool all_streams = false, bool allow_inline = true)
{
    PostArgs a;
    size_t shmem = 0, sh2 = 0;
    TRY(fill_post_args(n, li, s0, advance, all_streams, allow_inline, a, shmem));
    n->cur_layer = li;
    Post2Lay ly;
    post2_layout(n, a, ly, sh2);
    if (ly.ok && ns <= AEV_POST2_FEW_STREAMS) LAUNCH(n, KC_COMPACT, k_post2<AEV_POST2_NT_FEW>, dim3(ns), dim3(AEV_POST2_NT_FEW), sh2, a, ly);
    else if (ly.ok) LAUNCH(n, KC_COMPACT, k_post2<AEV_POST2_NT_MANY>, dim3(ns), dim3(AEV_POST2_NT_MANY), sh2, a, ly);
    else LAUNCH(n, KC_COMPACT, k_post, dim3(ns), dim3(1024), shmem, a);
    return AEV_OK;
}
// k_post of layer li and of layer li2 (the conv layer behind it) in one launch, in front of li2's recompute (k_post_pair)
static int launch_post_pair(aev_net *n, int li, int li2, int s0, int ns)
{
    PostArgs a, b;
    size_t sa = 0, sb = 0, s2a = 0, s2b = 0;
    TRY(fill_post_args(n, li, s0, nullptr, false, true, a, sa));
    TRY(fill_post_args(n, li2, s0, nullptr, false, false, b, sb)); // (no pool evaluation in place: the values do not exist yet)
    n->cur_layer = li;
    Post2Lay la, lb;
    post2_layout(n, a, la, s2a);
    n->cur_layer = li2;
    post2_layout(n, b, lb, s2b);
    n->cur_layer = li;
    if (la.ok && lb.ok && ns <= AEV_POST2_FEW_STREAMS) LAUNCH(n, KC_COMPACT, k_post2_pair<AEV_POST2_NT_FEW>, dim3(ns), dim3(AEV_POST2_NT_FEW), std::max(s2a, s2b), a, la, b, lb);
    else if (la.ok && lb.ok) LAUNCH(n, KC_COMPACT, k_post2_pair<AEV_POST2_NT_MANY>, dim3(ns), dim3(AEV_POST2_NT_MANY), std::max(s2a, s2b), a, la, b, lb);
    else LAUNCH(n, KC_COMPACT, k_post_pair, dim3(ns), dim3(1024), std::max(sa, sb), a, b);
    return AEV_OK;
}

// compute_all over the chain for streams [s0, s0+ns): 2 + 1 + 1 launches for the input layer and the leak sweep, then
// {recompute, k_post} per conv and {k_pool_eval} per pool.

static int sweep_gate_layer(const aev_net *n)
{
    long long total = 0, cum = 0;
    int last = -1;
    for (int li = 1; li < (int)n->L.size(); ++li) if (n->L[li].kind == AEV_KIND_CONV) { total += n->L[li].CHW; last = li; }
    // (a gate must be a conv layer with a conv layer behind it: anything else would never be reached by the chain's "li == gate"
    //  test, and the sweeps held back for it would never be launched)
    for (int li = 1; li < (int)n->L.size(); ++li) {
        if (n->L[li].kind != AEV_KIND_CONV) continue;
        cum += n->L[li].CHW;
        if (cum * 100 >= total * 75) return li == last ? 0 : li;
    }
    return 0;
}

// A queued step (qstep != nullptr) whose last layer is a conv moves the device step counter on in that layer's k_post (no
// separate launch); otherwise the caller launches k_advance after the chain.
static bool chain_advances(const aev_net *n) { return n->L.size() > 1 && n->L.back().kind == AEV_KIND_CONV && !n->generic; }

// The conv layers whose leak sweeps run a step ahead: conv k (in conv order) is swept beside conv k+2 of the previous step --
// conv k+1, the last reader of conv k's state within a step (through the pool layer's copy or the argmax bytes), is done by
// then, and conv k+2 is MFMA-bound work to hide the HBM-bound sweep under.  Only 3x3 tile-kernel layers count as cover (the
// 1x1 head kernels are over in tens of microseconds).  Layers 1 .. gate qualify (a prefix of the chain); returns the gate, or 0.
// *after is unused (kept for the callers' signature).
static int lookahead_layers(const aev_net *n, int *after)
{
    *after = 0;
    if (n->generic) return 0;
    std::vector<int> convs;
    for (int li = 1; li < (int)n->L.size(); ++li) if (n->L[li].kind == AEV_KIND_CONV) convs.push_back(li);
    int gate = 0;
    for (size_t k = 0; k + 2 < convs.size(); ++k) {
        if (n->L[convs[k + 2]].patch_cfg < 0) break;
        gate = convs[k];
    }
    // The first conv layer behind the gate: its OWN state can still be swept a step ahead, beside its consumer (the next conv
    // layer, a tile kernel that reads it only through the pool layer's copy); the copy is swept by the step itself.
    *after = 0;
    for (size_t k = 0; gate > 0 && k + 2 < convs.size(); ++k) {
        if (convs[k] != gate) continue;
        const int x = convs[k + 1], c = convs[k + 2];
        if (c == x + 2 && n->L[x + 1].kind == AEV_KIND_POOL && n->L[x + 1].pfm && n->L[c].patch_cfg >= 0 && n->L[c].patch_src == 0) *after = x;
    }
    return gate;
}

// Which plan a handle's queued steps get by default (AEV_OVERLAP / AEV_LOOKAHEAD / AEV_TAIL_OVERLAP force the pieces of the many-streams
// plan on or off): the many-streams plan from 232 streams on -- WHERE the items of the last tile-kernel layer (the longest of a step: one
// per tile and cout block, every tile active on saturating streams) fill the rounds of its persistent grid.  Measured on the e-fcn (round
// 4, ms per step, many / few): 192 streams 1.55 / 1.47, 224 1.736 / 1.715, 232 1.778 / 1.776, 240 1.801 / 1.812, 248 1.841 / 1.881,
// 256 1.880 / 1.937, 288 2.316 / 2.170, 320 2.466 / 2.410, 384 2.835 / 2.901.  At 288 streams conv5's 576 items are 2.25 rounds of its
// 256 blocks -- three rounds, the last one three quarters empty: the few-streams plan's other pipeline stage fills those CUs, the
// many-streams plan has nothing to put there.
static void choose_plan(aev_net *n)
{
    bool many = n->S >= 232 && !n->generic;
    int last = 0;
    for (int li = 1; li < (int)n->L.size(); ++li) if (n->L[li].kind == AEV_KIND_CONV && n->L[li].patch_cfg >= 0) last = li;
    if (many && last > 0 && n->L[last].patch_resident > 0) {
        const Layer &l = n->L[last];
        const double rounds = (double)l.tcap * l.patch_gz / l.patch_resident;
        many = rounds / std::ceil(rounds) >= 0.92;
    }
    n->overlap = n->generic ? 0 : n->sw.overlap >= 0 ? n->sw.overlap : many ? 1 : 0;
    n->lookahead = n->generic ? false : n->sw.lookahead >= 0 ? n->sw.lookahead != 0 : n->overlap != 0;
    n->tail_overlap = n->generic ? false : n->sw.tail_overlap >= 0 ? n->sw.tail_overlap != 0 : n->lookahead;
}

// The tail of a step (see launch_chain): the trailing run of 1x1-kernel conv layers, and the conv layer in front of it whose state
// (or the pool copy behind it) the tail reads.  The look-ahead sweeps must all have been launched by the time that layer is
// recomputed (they are: it is the last layer that can be their cover).
static void tail_layers(aev_net *n)
{
    n->tail_from = n->tail_feed = 0;
    if (n->generic) return;
    const int L = (int)n->L.size();
    int from = L;
    while (from - 1 >= 1 && n->L[from - 1].kind == AEV_KIND_CONV && n->L[from - 1].one_cfg >= 0) --from;
    if (from >= L || from < 2) return;
    int feed = from - 1;
    if (n->L[feed].kind == AEV_KIND_POOL) --feed;
    if (feed < 1 || n->L[feed].kind != AEV_KIND_CONV || feed <= n->la_gate) return;
    n->tail_from = from; n->tail_feed = feed;
    if (n->la_state >= feed) n->la_state = 0; // (the tail reads that layer's state or copy until the next step is under way)
}
// The few-streams plan pipelines two steps: the chip is idle enough to run a step's BACK half beside the next step's FRONT half, so
// its tail starts further up -- at the last conv layer served by the tile kernel (conv5 of the e-fcn: the tail is then conv5, pool5,
// conv6, conv7, about as long as the front half) -- as long as a tile-kernel layer is left in front of it.  The hand-over (plan_step)
// needs a pool layer with a copy between the tail and the layer in front of it: the tail must not read that layer's state directly.
static void tail_layers_few(aev_net *n)
{
    const int L = (int)n->L.size();
    auto hand_over_ok = [&](int from, int feed) { return from > 0 && feed + 2 == from && n->L[feed + 1].kind == AEV_KIND_POOL && n->L[feed + 1].pfm != nullptr; };
    n->tail_from_few = n->tail_feed_few = 0;
    if (n->generic) return;
    if (hand_over_ok(n->tail_from, n->tail_feed)) { n->tail_from_few = n->tail_from; n->tail_feed_few = n->tail_feed; } // (the 1x1 tail, unless a deeper one qualifies)
    int from = 0;
    for (int li = L - 1; li >= 1 && !from; --li) if (n->L[li].kind == AEV_KIND_CONV && n->L[li].patch_cfg >= 0) from = li;
    if (from < 2 || from >= L || n->L[from].kind != AEV_KIND_CONV) return;
    for (int li = from; li < L; ++li) { // every tail layer must be one the tail can run: tile / 1x1 kernel convs and pools over convs
        const Layer &l = n->L[li];
        if (l.kind == AEV_KIND_CONV && l.patch_cfg < 0 && l.one_cfg < 0) return;
        if (l.kind == AEV_KIND_POOL && n->L[li - 1].kind != AEV_KIND_CONV) return;
    }
    for (int li = from + 1; li < L; ++li) if (n->L[li].kind == AEV_KIND_CONV && n->L[li].patch_cfg >= 0) return; // (its only tile-kernel layer is its first)
    if (n->L.back().kind != AEV_KIND_CONV) return;
    int feed = from - 1;
    if (n->L[feed].kind == AEV_KIND_POOL) --feed;
    if (feed < 2 || n->L[feed].kind != AEV_KIND_CONV || n->L[feed].patch_cfg < 0) return; // (a tile-kernel layer in front: its k_post lists the tail's first tiles)
    if (!hand_over_ok(from, feed)) return;
    n->tail_from_few = from; n->tail_feed_few = feed;
}

static void fill_integ_args(aev_net *n, IntegArgs &a, const int *yxt, const long long *off, const int *qstep, int s0, bool ahead)
{
    Layer &l = n->L[0];
    a.yxt = yxt; a.off = off; a.qstep = qstep; a.S = n->S; a.s0 = s0;
    a.H = l.H; a.W = l.W; a.HW = l.HW; a.leak = n->leak;
    a.surf = l.surf; a.sstride = l.sstride; a.owner = l.owner;
    a.prev_ts = n->d_prev_ts; a.last_ts = n->d_last_ts; a.dl64 = n->d_dl64; a.dlf = n->d_dlf; a.active = n->d_active;
    a.evmap = l.mapA; a.mstride = l.mstride; a.stat_events = n->d_stats + 62;
    a.zero = nullptr; a.nzero = 0;
    a.qrun = ahead ? n->d_qrun : nullptr; a.dlf_next = ahead ? n->d_dlf_next : nullptr; a.active_next = ahead ? n->d_active_next : nullptr;
}

// Every event the chain records, created up front: launch_chain may run inside a stream capture, and nothing but stream
// operations should happen on a capturing thread.
static int ensure_chain_events(aev_net *n)
{
    n->ev_sweep.resize(n->L.size(), nullptr);
    n->ev_la_a.resize(n->L.size(), nullptr);
    for (size_t li = 1; li < n->L.size(); ++li) {
        if (n->L[li].kind != AEV_KIND_CONV) continue;
        if (!n->ev_sweep[li]) HIPCHK(hipEventCreateWithFlags(&n->ev_sweep[li], hipEventDisableTiming));
        if (!n->ev_la_a[li]) HIPCHK(hipEventCreateWithFlags(&n->ev_la_a[li], hipEventDisableTiming));
    }
    if (!n->ev_join) HIPCHK(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
    if (!n->ev_tail_a) HIPCHK(hipEventCreateWithFlags(&n->ev_tail_a, hipEventDisableTiming));
    n->tail_pending = n->tail_sweeps4 = false; // (a run that failed half way)
    return AEV_OK;
}

// ------------------------------------------------------------------------------------------------
// One step of the chain = a PLAN (plan_step: a pure function of the handle's configuration and of where the step sits in its
// run -> the ordered list of launches, event records and cross-stream waits) + its EMISSION (emit_step: the list, executed
// in order).  Everything that decides what overlaps with what lives in the two plan functions -- plan_few_streams (queued steps of
// up to 207 streams) and plan_general (everything else: the plain chain and the many-streams plan); check_plan asserts what every
// plan must satisfy before anything is launched.
//
// HIP streams of a step:  CHAIN  the layer chain itself (the handle's stream; a captured step starts and ends here)
//                         SWEEP  many streams: this step's own leak sweeps, forked behind k_integ_sweep; few: the bookkeeping chain
//                         AHEAD  many streams: the NEXT step's sweeps of the first conv layers, beside this step's deep layers
//                         TAIL   the PREVIOUS step's tail (many streams: the trailing 1x1 conv layers; few: its back half)
// ------------------------------------------------------------------------------------------------
enum { ST_CHAIN = 0, ST_SWEEP, ST_AHEAD, ST_TAIL, ST_COUNT };
enum { OP_INTEG = 0, OP_SWEEP, OP_POST, OP_RECOMPUTE, OP_POOL_EVAL, OP_DENSE_FILL, OP_ZERO_COUNTERS, OP_RECORD, OP_WAIT };
enum { EVT_PREP = 0, EVT_JOIN, EVT_TAIL_A, EVT_LA_B, EVT_SWEEP0 = 8, EVT_LA_A0 = 8 + 64, EVT_COUNT = 8 + 128 }; // + layer index
struct StepOp {
    int kind = 0, stream = ST_CHAIN;
    int layer = 0, layer_hi = 0;                             // POST / RECOMPUTE / POOL_EVAL: layer; SWEEP: layers [layer, layer_hi]
    int ev = 0;                                              // RECORD: event recorded on `stream`; WAIT: event `stream` waits for
    int mode = 0, bpc = 1, parts = 3, hi_parts = 0; bool with_copies = false; // SWEEP (launch_leak_sweep)
    bool advance = false, all_streams = false, no_inline = false; // POST (no_inline: a small pool layer behind it is NOT evaluated in place)
    int pair = 0;                                            // POST: > 0: the k_post of that layer too, in the same launch (k_post_pair)
    bool ahead = false;                                      // INTEG: also computes the next step's delta_leak (look-ahead)
    bool in_head = false;                                    // INTEG + the SWEEP behind it on the chain: ONE launch (k_head) where the step's events allow k_integ_step
};
struct StepPlan {
    std::vector<StepOp> ops;
    bool defers_tail = false;   // the trailing 1x1 layers are left to the next step of the run
    bool tail_sweeps_on_tail_stream = false; // ... whose sweeps of those layers this step has run on the TAIL stream (no events)
    int tails_launched = 0;     // the previous step's tail: 0 or 1
    int tail_from = 0;          // first layer of the tail this plan works with (0: none)
    const char *error = nullptr;
};

struct StepWhere { bool queued, whole, more, tail_prev, tail_prev_sw4; }; // queued run / all streams / another step follows / state handed over

// The list under construction: the operations of a step in capture order.
struct PlanOps {
    std::vector<StepOp> &ops;
    StepOp &push(const StepOp &o) { ops.push_back(o); return ops.back(); }
    void record(int ev, int st) { StepOp o; o.kind = OP_RECORD; o.stream = st; o.ev = ev; push(o); }
    void wait(int st, int ev) { StepOp o; o.kind = OP_WAIT; o.stream = st; o.ev = ev; push(o); }
    StepOp &sweep(int lo, int hi, int st, bool with_copies, int mode = 0, int bpc = 1, int parts = 3)
    {
        StepOp o; o.kind = OP_SWEEP; o.stream = st; o.layer = lo; o.layer_hi = hi; o.with_copies = with_copies; o.mode = mode; o.bpc = bpc; o.parts = parts;
        return push(o);
    }
    void recompute(int li, int st) { StepOp o; o.kind = OP_RECOMPUTE; o.stream = st; o.layer = li; push(o); }
    StepOp &post(int li, int st, bool advance, bool all_streams)
    {
        StepOp o; o.kind = OP_POST; o.stream = st; o.layer = li; o.advance = advance; o.all_streams = all_streams;
        return push(o);
    }
    void pool_eval(int li, int st) { StepOp o; o.kind = OP_POOL_EVAL; o.stream = st; o.layer = li; push(o); }
    void zero_counters(int from, int st) { StepOp o; o.kind = OP_ZERO_COUNTERS; o.stream = st; o.layer = from; push(o); }
    StepOp &integ(bool ahead, int clear_below) // (clear_below: k_integ_prepare clears the scratch counters of the layers below it)
    {
        StepOp o; o.kind = OP_INTEG; o.ahead = ahead; o.layer = clear_below;
        return push(o);
    }
};
// a pool layer is evaluated by a launch of its own unless the k_post of the conv layer below does it in place
static bool pool_needs_launch(const aev_net *n, int li) { return !(pool_eval_in_post(n, li) && !n->dense); }
// The input layer's k_post and the first conv layer's as ONE launch (k_post_pair) in front of that layer's recompute: the bookkeeping
// needs the marks, not the values -- one launch boundary less on every step's critical path.  The conv layer must be list-driven (its
// recompute reads the site list the input layer's k_post wrote, not the touch map its own k_post clears) and the pool layer behind it
// one that has a launch of its own anyway (evaluated in place it would need the values).  Not in the per-kernel timing pass (a launch
// is booked on one layer).
static bool first_posts_pair(const aev_net *n)
{
    const int L = (int)n->L.size();
    if (L < 3 || n->L[1].kind != AEV_KIND_CONV || n->dense || n->conv_impl != 0 || n->profiling) return false;
    const Layer &c = n->L[1];
    if (c.one_cfg >= 0 || c.patch_cfg >= 0 || !conv_needs_list(n, c)) return false;
    return n->L[2].kind != AEV_KIND_POOL || pool_needs_launch(n, 2);
}

// ---- FEW STREAMS (queued steps of all streams, up to Switches::hoist_max of them, tile kernels, no dense mode) --------------------
// Latency mode: the chip is mostly idle, a step is a chain of ~21 short dependent launches, 42 % of it k_post.  Two ideas:
//
// 1. The BOOKKEEPING of a step does not depend on the VALUES the step computes -- a conv layer's events are its touched sites (geometry
//    of the layer below's events) and its sign changes (the leak sweep, done at the head of the step); a pool layer's windows are those
//    events' windows and the windows left pending by the PREVIOUS step.  So the k_posts of the tile-kernel layers run as a chain of
//    their own on the second stream (ST_SWEEP), ahead of the recomputes, and the value chain (recompute -> pool eval -> recompute ...)
//    only waits for the list it is about to read.  Not hoisted: layers of the 1x1 kernel (it reads the touch map its own k_post
//    clears), the layer that ends the chain, the layers of the tail; a small pool layer behind a hoisted k_post is evaluated by a
//    launch of its own (its k_post runs before the values exist).  Every cross-stream edge of the captured step costs the waiting
//    launch ~5-10 us and a launch with successors on two streams another ~4: the first two hoisted k_posts (the value chain is right
//    behind them) get an edge each, all the later ones -- done long before the value chain arrives -- share ONE, recorded behind the
//    last of them.
//
// 2. The two halves of a step as the two stages of a pipeline (a net for which tail_layers_few finds the structure): the BACK half
//    (the tail: from the last tile-kernel conv layer on -- conv5 ... conv7 of the e-fcn) of step t runs beside the FRONT half of
//    step t + 1, on ST_TAIL.  Only three things of a front half touch what the tail touches: tail_feed's k_post (writes the first tail
//    layer's touch map, tile list and counters -- and moves the step counter on: nothing after the head of a step reads it), the pool
//    eval behind tail_feed, and the sweep of that pool's copy (what the first tail layer reads).  All three only have to wait for the
//    FIRST tail layer.  So:
//      * the tail forks at the very top of the step, in front of k_integ_prepare (its kernels read neither d_active -- its k_posts run
//        for all streams: a stream without events has empty maps and gets the same empty lists -- nor delta_leak, and
//        k_integ_prepare leaves the tail layers' scratch counters alone: the tail's tile kernel may still be reading them; they are
//        cleared behind it, before tail_feed's k_post appends the next tiles);
//      * behind its first layer the tail stream waits for k_integ_prepare (delta_leak), sweeps the pool copy in front of it and records
//        the event tail_feed's k_post and the pool eval wait for;
//      * tail_feed's own state is swept with the front half's layers (one launch), its recompute waits for nothing of the tail;
//      * the rest of the tail, then this step's sweeps of the tail layers, and ONE join at the end of the step.
//    A step lasts  max(front half, tail) + the join.
static StepPlan plan_few_streams(const aev_net *n, const StepWhere &w)
{
    StepPlan P;
    PlanOps o{P.ops};
    const int L = (int)n->L.size();
    const bool pipelined = n->tail_from_few > 0 && chain_advances(n);
    const int tfrom = pipelined ? n->tail_from_few : n->tail_from, tfeed = pipelined ? n->tail_feed_few : n->tail_feed;
    const bool incoming = w.tail_prev;            // the previous step has left its tail to this one
    const bool defers = w.more && pipelined;      // this step leaves its tail to the next
    P.tail_from = tfrom;
    if (incoming && !pipelined) { P.error = "internal: a step's tail is pending outside a queued run"; return P; }
    bool deep_tail = false; // a tile-kernel layer inside the tail: its list counters outlive the step
    for (int li = std::max(tfrom, 1); li < L && tfrom > 0; ++li) deep_tail |= n->L[li].kind == AEV_KIND_CONV && n->L[li].patch_cfg >= 0;

    // the previous step's tail, first layer: in front of everything
    if (incoming) {
        o.record(EVT_TAIL_A, ST_CHAIN);
        o.wait(ST_TAIL, EVT_TAIL_A);
        o.recompute(tfrom, ST_TAIL);
        if (deep_tail) o.zero_counters(tfrom, ST_TAIL);
        o.post(tfrom, ST_TAIL, false, true);
    }
    // head of this step
    // (k_head: k_integ_step and the head's sweep in one launch, where the sweep fits one launch -- up to four streams: one stream 0.163 ->
    //  0.160 ms per step, four 0.195 -> 0.192, eight / sixteen +-0, 64 streams +4 %: every sweep block scans the step's events for itself)
    int head_segs = 0;
    for (int li = 1; li <= (incoming ? tfeed : L - 1); ++li)
        if (n->L[li].kind == AEV_KIND_CONV) { ++head_segs; for (int lp = li + 1; lp < L && n->L[lp].kind == AEV_KIND_POOL && n->L[lp].pfm; ++lp) ++head_segs; }
    const bool head = head_segs > 0 && head_segs <= AEV_MAX_SEGS && n->S <= 4;
    o.integ(false, incoming && deep_tail ? tfrom : L).in_head = head;
    if (incoming) { StepOp &sw = o.sweep(1, tfeed, ST_CHAIN, true); sw.hi_parts = 1; sw.in_head = head; } // (tail_feed: its own state only; the tail layers: behind the tail)
    else o.sweep(1, L - 1, ST_CHAIN, true).in_head = head;
    // the rest of the previous step's tail, then this step's sweeps of its layers
    if (incoming) {
        o.record(EVT_LA_B, ST_CHAIN); // (behind k_integ_prepare: this step's delta_leak is known)
        o.wait(ST_TAIL, EVT_LA_B);
        o.sweep(tfeed, tfeed, ST_TAIL, true, 0, 1, 2); // the pool copy behind tail_feed: the first tail layer has read it
        o.record(EVT_SWEEP0 + tfeed, ST_TAIL);
        for (int li = tfrom + 1; li < L; ++li) {
            if (n->L[li].kind == AEV_KIND_POOL) { if (pool_needs_launch(n, li)) o.pool_eval(li, ST_TAIL); continue; }
            o.recompute(li, ST_TAIL);
            o.post(li, ST_TAIL, false, true);
        }
        o.sweep(tfrom, L - 1, ST_TAIL, true);
        o.record(EVT_JOIN, ST_TAIL);
        P.tail_sweeps_on_tail_stream = true;
        P.tails_launched += 1;
    }
    StepOp &post0 = o.post(0, ST_CHAIN, false, false);
    const size_t post0_at = P.ops.size() - 1;
    (void)post0;

    // the bookkeeping chain: which k_posts run ahead
    std::vector<int> hoisted; // layers, in order (a chain: every later k_post needs the marks of the one before)
    int feed_hoisted = 0;     // tail_feed, when its k_post runs ahead too: waited for behind its recompute, with the tail's first layer
    {
        int last_conv = 0;
        for (int li = 1; li < L; ++li) if (n->L[li].kind == AEV_KIND_CONV) last_conv = li;
        for (int li = 1; li < L; ++li) {
            const Layer &l = n->L[li];
            if (l.kind != AEV_KIND_CONV) continue;
            if (l.one_cfg >= 0 || li == last_conv) break;
            if ((defers || incoming) && li >= tfrom) break; // (the tail's layers run as a block of their own)
            hoisted.push_back(li);
            if ((defers || incoming) && li == tfeed) { feed_hoisted = li; break; }
        }
    }
    std::vector<int> wait_for(L, 0); // the event the consumers of a hoisted layer's lists wait for (0: covered by an earlier wait)
    std::vector<char> is_hoisted(L, 0);
    if (!hoisted.empty() && hoisted[0] == 1 && feed_hoisted != 1 && first_posts_pair(n)) {
        // the first conv layer's k_post: with the input layer's, in one launch on the chain (no edge: the value chain is behind it)
        P.ops[post0_at].pair = 1;
        is_hoisted[1] = 1;
        hoisted.erase(hoisted.begin());
    }
    if (!hoisted.empty()) {
        const int own_edges = 2;
        int last_shared = 0; // the last hoisted layer that is not tail_feed: its record serves all the layers behind the first two
        for (int li : hoisted) if (li != feed_hoisted) last_shared = li;
        o.record(EVT_PREP, ST_CHAIN);
        o.wait(ST_SWEEP, EVT_PREP);
        int seen = 0;
        bool shared_taken = false;
        for (int li : hoisted) {
            const bool feed = li == feed_hoisted;
            // tail_feed's k_post writes the tail's touch map -- it waits for the previous step's tail (whose kernel reads that map and
            // whose k_post clears it) -- and moves the step counter on when this step leaves its own tail behind
            if (feed && incoming) o.wait(ST_SWEEP, EVT_SWEEP0 + tfeed);
            StepOp &p = o.post(li, ST_SWEEP, feed && defers && w.queued, false);
            p.no_inline = true;
            is_hoisted[li] = 1;
            ++seen;
            if (seen <= own_edges || li == last_shared || feed) o.record(EVT_LA_A0 + li, ST_SWEEP);
            if (feed) continue; // (its own edge, behind its recompute)
            if (seen <= own_edges) wait_for[li] = EVT_LA_A0 + li;
            else if (!shared_taken) { shared_taken = true; wait_for[li] = EVT_LA_A0 + last_shared; }
        }
    }

    // the value chain
    for (int li = 1; li < L; ++li) {
        if (n->L[li].kind == AEV_KIND_POOL) {
            if (n->L[li - 1].kind == AEV_KIND_POOL) { P.error = "a pool layer directly after a pool layer is not supported"; return P; }
            if (defers && li >= tfrom) continue; // (inside the tail this step leaves to the next)
            if (pool_needs_launch(n, li) || is_hoisted[li - 1]) o.pool_eval(li, ST_CHAIN); // (else: by the k_post of the conv layer below)
            continue;
        }
        if (defers && li >= tfrom) { P.defers_tail = true; continue; } // left to the next step
        const bool feed_behind_tail = incoming && li == tfeed; // (its recompute waits for nothing of the tail: the waits follow it)
        if (li == feed_hoisted && !feed_behind_tail) o.wait(ST_CHAIN, EVT_LA_A0 + li); // (its k_post ran ahead)
        if (incoming && li == tfrom) o.wait(ST_CHAIN, EVT_JOIN); // (a step that runs its own tail layers: behind the previous step's tail)
        o.recompute(li, ST_CHAIN);
        if (feed_behind_tail) { o.wait(ST_CHAIN, EVT_SWEEP0 + li); if (li == feed_hoisted) o.wait(ST_CHAIN, EVT_LA_A0 + li); }
        const bool advance = defers ? li == tfeed : li + 1 == L && chain_advances(n);
        if (li == feed_hoisted) { /* waited for above */ }
        else if (is_hoisted[li]) { if (wait_for[li]) o.wait(ST_CHAIN, wait_for[li]); } // the lists the next launches read are complete
        else o.post(li, ST_CHAIN, advance && w.queued, false);
    }
    if (incoming && P.tails_launched != 1) P.error = "internal: the previous step's tail was not launched exactly once";
    if (incoming) o.wait(ST_CHAIN, EVT_JOIN); // the step's one join with the tail stream
    return P;
}

// The first step of a run with the look-ahead has nobody who swept its layers 1 .. la_gate a step ahead.  Where a step can tell that it
// is the first -- where every later step of a run has a tail handed over to it (tail overlap) -- it sweeps them itself, on the second
// stream beside its head (the first conv layer's k_post waits for that layer's sweep only); elsewhere the run's prologue sweeps them
// before the first step starts (queue_launch: alone on the device, 280 us at 256 e-fcn streams -- half of what a run costs beyond
// its steps, docs/KERNEL_NOTES.md).
static bool first_step_sweeps_itself(const aev_net *n)
{
    return n->lookahead && n->la_gate > 0 && !n->dense && n->overlap && !n->profiling && n->tail_overlap && n->tail_from > 0 && n->conv_impl == 0 &&
           chain_advances(n);
}

// ---- EVERYTHING ELSE: the plain chain (one stream), and with AEV_OVERLAP / AEV_LOOKAHEAD or from 208 streams on the many-streams
// plan -- this step's sweeps on a second stream, the first conv layers' sweeps a step AHEAD beside the deep layers, the trailing 1x1
// conv layers left to the next step.
static StepPlan plan_general(const aev_net *n, const StepWhere &w)
{
    StepPlan P;
    PlanOps o{P.ops};
    const int L = (int)n->L.size();
    // Look-ahead (queued steps of one run only: the device knows the next step's events, and nothing can come between two
    // steps of a run): this step's sweeps of layers 1..la_gate have been done by the previous step, or by the run's prologue
    // (queue_launch), and the next step's are launched below, beside the deep layers.
    const bool la = w.queued && n->lookahead && n->la_gate > 0 && !n->dense && w.whole;
    const int la_state = la && n->conv_impl == 0 ? n->la_state : 0; // (own state swept ahead, pool copy by the step: see lookahead_layers)
    const bool fork = n->overlap && !n->profiling && !n->dense;
    const bool self_sweep = la && fork && first_step_sweeps_itself(n) && !w.tail_prev; // the run's first step: see first_step_sweeps_itself
    const bool swept_ahead = la && !self_sweep;
    // Tail overlap (queued steps with the look-ahead): the trailing 1x1 conv layers of the step -- 64..288 blocks for 40-55 us
    // each on the e-fcn, and two one-block-per-stream k_posts -- are left out of the step and launched by the NEXT step, on a
    // stream of their own beside its head (conv1, pool1: bandwidth-bound kernels that leave the MFMA pipes idle), from the
    // point tail_at on (after which layer's k_post).  What the two share, and how it is ordered:
    //   * the step counter moves on in the k_post of the layer before the tail (tail_feed), after the look-ahead sweeps of this
    //     step have been joined: nothing behind that launch reads it (pool eval, 1x1 conv and k_post do not);
    //   * d_active belongs to the next step as soon as its k_integ_prepare runs: the tail's k_posts run for all streams (a
    //     stream without events has empty maps and gets the same empty lists);
    //   * the next step's sweeps of tail_feed (whose pool copy the tail reads) and of the tail layers follow the tail on its
    //     stream; the next step's tail_feed recompute waits for them, which also orders its k_post (the writer of the tail's
    //     touch map) and its pool eval behind this step's tail.
    const bool tail_prev = w.tail_prev, tail_prev_sw4 = w.tail_prev_sw4;
    const int tfrom = n->tail_from, tfeed = n->tail_feed;
    P.tail_from = tfrom;
    const bool tail_fork = w.more && la && fork && n->tail_overlap && n->tail_from > 0 && n->conv_impl == 0 && chain_advances(n);
    if (tail_prev && !(la && fork)) { P.error = "internal: a step's tail is pending outside a queued run"; return P; }
    bool deep_tail = false; // a tile-kernel layer inside the tail: its list counters outlive the step
    for (int li = std::max(tfrom, 1); li < L && tfrom > 0; ++li) deep_tail |= n->L[li].kind == AEV_KIND_CONV && n->L[li].patch_cfg >= 0;
    o.integ(la, tail_prev && deep_tail ? tfrom : L); // (the previous step's tail still reads its layers' counters: cleared behind it, see tail())
    int tail_at = std::max(-1, std::min((int)aev_net::tail_at, tfeed - 1));
    if (tail_at > 0 && n->L[tail_at].kind != AEV_KIND_CONV) --tail_at; // (a pool layer has no k_post of its own: the conv layer before it)
    // The layer behind the look-ahead gate (la_state: its own state is swept a step ahead, the pool copy behind it by the step itself): in
    // the steady state of a run that copy's sweep rides on the TAIL stream, in front of the tail layers' sweeps, instead of a fork of its
    // own at the head of the step -- one fork and one cross-stream edge into the chain fewer (la_state's recompute waits for the tail
    // stream's one event, the tail-feed layer behind it needs no wait of its own).  Round 5: conv4 / conv5 of the e-fcn each started 12-14 us
    // late behind such an edge.
    const bool copy_on_tail = n->sw.la_copy_on_tail && tail_prev && la_state > 0 && swept_ahead && fork && la_state < tfeed && tail_at >= 0;
    auto tail = [&]() { // the previous step's tail, then this step's sweeps of those layers: on the tail's stream
        o.record(EVT_TAIL_A, ST_CHAIN); // (behind k_integ_prepare: this step's delta_leak is known)
        o.wait(ST_TAIL, EVT_TAIL_A);
        for (int li = tfrom; li < L; ++li) {
            if (n->L[li].kind == AEV_KIND_POOL) { if (pool_needs_launch(n, li)) o.pool_eval(li, ST_TAIL); continue; }
            if (fork && !tail_prev_sw4) o.wait(ST_TAIL, EVT_SWEEP0 + li); // (the previous step's sweep of the layer: second stream)
            o.recompute(li, ST_TAIL);
            o.post(li, ST_TAIL, false, true);
        }
        // the tail layers' scratch counters (tile-list sizes, work queues): this step's k_integ_prepare has left them alone (the tail's
        // tile kernels may still have been reading them), they are cleared here, before tail_feed's k_post appends the next tiles
        if (deep_tail) o.zero_counters(tfrom, ST_TAIL);
        if (copy_on_tail) o.sweep(la_state, la_state, ST_TAIL, true, 0, 1, 2);
        o.sweep(tfeed, L - 1, ST_TAIL, true);
        o.record(EVT_SWEEP0 + tfeed, ST_TAIL);
        P.tail_sweeps_on_tail_stream = true;
        P.tails_launched += 1;
    };
    int sweep_gate = 0;
    if (fork) { // per-layer sweeps on the second stream, in layer order; they only need delta_leak (k_integ_prepare)
        bool forked = false; // (the second stream joins the step with its first launch: a fork without work would be left unjoined)
        // The sweeps of the first conv layers (the bulk of the state) start right away; those of the deep layers -- the
        // last ~25 % of the bytes -- start when the recompute of the layer before them does, instead of crowding the
        // bandwidth-bound head of the step (A/B on the e-fcn: gate at conv2 -1.9 % against everything up front, -0.3 %
        // against a gate at conv3).
        sweep_gate = sweep_gate_layer(n);
        // (with the look-ahead the head of a step has almost no sweeps left: the few that remain start right away -- beside conv2
        //  of the e-fcn they cost its tile kernel 67 us)
        if (la) sweep_gate = 0;
        for (int li = 1; li < L; ++li) {
            if (n->L[li].kind != AEV_KIND_CONV) continue;
            if (swept_ahead && li <= n->la_gate) continue;  // swept a step ahead
            if (sweep_gate > 0 && li > sweep_gate) continue; // launched later, beside the recompute of layer sweep_gate
            if (tail_prev && li >= tfeed) continue;   // behind the previous step's tail, on its stream (tail())
            if (copy_on_tail && li == la_state) continue; // ... and so is this copy
            if (!forked) { // (recorded behind k_integ_sweep, although the sweeps need k_integ_prepare only: forked right behind that, +0.8 %)
                o.record(EVT_PREP, ST_CHAIN);
                o.wait(ST_SWEEP, EVT_PREP);
                forked = true;
            }
            o.sweep(li, li, ST_SWEEP, true, 0, 1, li == la_state && swept_ahead ? 2 : 3);
            o.record(EVT_SWEEP0 + li, ST_SWEEP);
        }
    } else if (!n->dense) {
        if (la) {
            int lo = n->la_gate + 1;
            if (la_state) { o.sweep(lo, la_state, ST_CHAIN, true, 0, 1, 2); lo = la_state + 1; }
            if (lo < L) o.sweep(lo, L - 1, ST_CHAIN, true);
        } else
            o.sweep(1, L - 1, ST_CHAIN, true);
    }
    if (tail_prev && tail_at < 0) tail();
    // (k_post_pair: the first conv layer's k_post with the input layer's, in front of its recompute -- behind its sweep)
    const bool pair01 = first_posts_pair(n) && !(tail_fork ? tfeed == 1 : false) && !(tail_prev && (tail_at == 0 || tfeed <= 1)) && !(sweep_gate > 0 && 1 >= sweep_gate);
    if (pair01 && fork && !(swept_ahead && 1 <= n->la_gate)) o.wait(ST_CHAIN, EVT_SWEEP0 + 1);
    o.post(0, ST_CHAIN, false, false).pair = pair01 ? 1 : 0;
    if (tail_prev && tail_at == 0) tail();
    if (n->dense && L > 1) { // dense-frame baseline: the layer after the input recomputes everything
        if (n->L[1].kind != AEV_KIND_CONV) { P.error = "dense mode needs a conv layer after the input layer"; return P; }
        StepOp d; d.kind = OP_DENSE_FILL; d.layer = 1; o.push(d);
    }
    int last_gated = 0; // last layer whose sweep was held back (its event stands for all of them: the SWEEP stream runs in order)
    bool la_join = false;
    int prev_conv = 0, prev_conv2 = 0;
    for (int li = 1; li < L; ++li) {
        if (n->L[li].kind == AEV_KIND_POOL) {
            if (n->L[li - 1].kind == AEV_KIND_POOL) { P.error = "a pool layer directly after a pool layer is not supported"; return P; }
            if (tail_fork && li >= tfrom) continue; // (inside the tail this step leaves to the next)
            if (pool_needs_launch(n, li)) o.pool_eval(li, ST_CHAIN); // (else: by the k_post of the conv layer below)
            continue;
        }
        if (tail_fork && li >= tfrom) { P.defers_tail = true; continue; } // left to the next step
        if (fork && sweep_gate > 0 && li == sweep_gate) { // the deeper layers' sweeps start when this layer's recompute does
            o.wait(ST_CHAIN, EVT_SWEEP0 + li);
            o.record(EVT_JOIN, ST_CHAIN);
            o.wait(ST_SWEEP, EVT_JOIN);
            // (one join for all of them, at the first of these layers: every cross-stream edge of the captured graph
            // costs the waiting kernel ~5 us, and these sweeps are done long before the second of the layers starts)
            for (int lj = li + 1; lj < L; ++lj) {
                if (n->L[lj].kind != AEV_KIND_CONV) continue;
                o.sweep(lj, lj, ST_SWEEP, true);
                last_gated = lj;
            }
            if (last_gated) o.record(EVT_SWEEP0 + last_gated, ST_SWEEP);
        }
        if (fork && sweep_gate > 0 && li > sweep_gate) {
            if (last_gated) o.wait(ST_CHAIN, EVT_SWEEP0 + last_gated);
            last_gated = 0;
        } else if (copy_on_tail && li == la_state) o.wait(ST_CHAIN, EVT_SWEEP0 + tfeed); // (recorded behind the tail stream's sweeps, this copy's among them)
        else if (copy_on_tail && li == tfeed) { /* the chain has waited for that event at la_state */ }
        else if (fork && !(tail_prev && li > tfeed) && !(swept_ahead && li <= n->la_gate) && !(pair01 && li == 1)) // this layer's state has been extrapolated
            o.wait(ST_CHAIN, EVT_SWEEP0 + li);
        const bool last = li + 1 == L;
        if (la && prev_conv2 >= 1 && prev_conv2 <= n->la_gate) {
            // Nothing in this step touches the conv layer two before this one (or the pool copy behind it) any more -- the conv
            // layer in between was its last reader: its sweep for the NEXT step runs now, beside this layer's MFMA-bound
            // recompute.  (One launch per layer on the AHEAD stream.  Beside the one-block-per-CU instances of the tile kernel
            // there are registers for one sweep wave per SIMD, beside the others for two -- one sweep block per CU everywhere: two
            // beside the smaller instances finish sooner and cost the tile kernels what they gain.)
            if (fork) {
                o.record(EVT_LA_A0 + li, ST_CHAIN);
                o.wait(ST_AHEAD, EVT_LA_A0 + li);
                o.sweep(prev_conv2, prev_conv2, ST_AHEAD, true, 2, 1);
                o.record(EVT_LA_B, ST_AHEAD);
                la_join = true;
            } else
                o.sweep(prev_conv2, prev_conv2, ST_CHAIN, true, 2, 1);
        }
        if (la_state && prev_conv == la_state) { // the previous conv layer's own state: this layer reads it through the pool copy only
            if (fork) {
                o.record(EVT_LA_A0 + li, ST_CHAIN);
                o.wait(ST_AHEAD, EVT_LA_A0 + li);
                o.sweep(la_state, la_state, ST_AHEAD, true, 2, 1, 1);
                o.record(EVT_LA_B, ST_AHEAD);
                la_join = true;
            } else
                o.sweep(la_state, la_state, ST_CHAIN, true, 2, 1, 1);
        }
        o.recompute(li, ST_CHAIN);
        prev_conv2 = prev_conv; prev_conv = li;
        const bool advance = tail_fork ? li == tfeed : last && chain_advances(n);
        if ((last || advance) && la_join) { o.wait(ST_CHAIN, EVT_LA_B); la_join = false; } // (before the step counter moves on)
        if (!(pair01 && li == 1)) o.post(li, ST_CHAIN, advance && w.queued, false);
        if (tail_prev && tail_at == li) tail();
    }
    if (la_join) o.wait(ST_CHAIN, EVT_LA_B); // (a chain that ends with a pool layer: k_advance follows)
    if (tail_fork && !tail_prev) // the tail layers' sweeps of THIS step ran on the second stream and nobody has waited for them yet:
        for (int li = tfrom; li < L; ++li) if (n->L[li].kind == AEV_KIND_CONV) o.wait(ST_CHAIN, EVT_SWEEP0 + li); // (a captured step must end joined)
    if (tail_prev && P.tails_launched != 1) P.error = "internal: the previous step's tail was not launched exactly once";
    return P;
}

// Capture order at a fork (round 5).  A captured step becomes a graph whose nodes the runtime spreads over hardware queues: at a node with two
// successors the one captured FIRST stays on its predecessor's queue, the other moves to another queue and starts 12-18 us late (a
// cross-queue barrier packet; profiles/r05*_timeline_S256.txt: conv4 / conv5 each started 16-18 us behind the k_post in front of them,
// because the look-ahead sweep forked there had been captured first).  So at every fork  RECORD(e, CHAIN); WAIT(branch, e); <branch ops>;
// <next CHAIN launch>  the chain's next launch is moved in front of the branch: the critical path keeps its queue, the branch -- off the
// critical path by construction -- pays the hop.  Dependencies are unchanged: the branch waits for e, recorded before the moved launch; the
// moved launch waited for nothing in between (a CHAIN wait inside the run stops the move); a second RECORD of the same event on the
// chain inside the run (the look-ahead's two sweeps at one layer) is dropped, its WAIT then refers to the first.
static void chain_first(std::vector<StepOp> &ops, int which) // which: bit s = forks whose branch runs on stream s (ST_*)
{
    auto launch = [](const StepOp &o) { return o.kind != OP_RECORD && o.kind != OP_WAIT; };
    for (size_t i = 0; i + 1 < ops.size(); ++i) {
        if (!(ops[i].kind == OP_RECORD && ops[i].stream == ST_CHAIN)) continue;
        if (!(ops[i + 1].kind == OP_WAIT && ops[i + 1].stream != ST_CHAIN && ops[i + 1].ev == ops[i].ev)) continue;
        if (!((which >> ops[i + 1].stream) & 1)) continue;
        size_t j = i + 1;
        bool ok = true;
        std::vector<size_t> dup;
        for (; j < ops.size(); ++j) {
            const StepOp &o = ops[j];
            if (o.stream != ST_CHAIN) continue;                      // the branch's launches, records and waits
            if (o.kind == OP_WAIT) { ok = false; break; }            // the chain joins something here: its next launch depends on it
            if (o.kind == OP_RECORD) { if (o.ev == ops[i].ev) { dup.push_back(j); continue; } ok = false; break; }
            break;                                                    // the chain's next launch
        }
        if (!ok || j >= ops.size() || !launch(ops[j])) continue;
        const StepOp moved = ops[j];
        ops.erase(ops.begin() + (long)j);
        for (size_t d = dup.size(); d-- > 0;) ops.erase(ops.begin() + (long)dup[d]);
        ops.insert(ops.begin() + (long)i + 1, moved);
    }
}

// Which plan a step gets.
static bool few_streams_plan(const aev_net *n, const StepWhere &w)
{
    const bool la = w.queued && n->lookahead && n->la_gate > 0 && !n->dense && w.whole;
    const bool fork = n->overlap && !n->profiling && !n->dense;
    return w.queued && w.whole && !fork && !la && !n->dense && n->conv_impl == 0 && !n->profiling && n->S <= n->sw.hoist_max;
}
static StepPlan plan_step(const aev_net *n, const StepWhere &w)
{
    StepPlan P = few_streams_plan(n, w) ? plan_few_streams(n, w) : plan_general(n, w);
    if (!P.error && n->sw.chain_first && !few_streams_plan(n, w)) chain_first(P.ops, n->sw.chain_first);
    return P;
}

// What every plan satisfies, whatever the switches: (1) the step ends JOINED -- every launch on another stream is ordered before
// the end of the CHAIN stream through record / wait pairs (a captured step that leaves a stream unjoined cannot end its capture,
// and an eager one would let the next step run into it), except the sweeps the TAIL stream carries into the next step's tail_feed
// wait, which that step's plan joins; (2) a stream's first launch is behind a wait (forked, not free-running); (3) every event is
// recorded before it is waited for, unless it belongs to the previous step (the incoming tail's sweep events); (4) every conv
// layer is recomputed exactly once or deferred as a whole tail, every pool layer evaluated exactly once.
static const char *check_plan(const aev_net *n, const StepPlan &P, const StepWhere &w)
{
    const int L = (int)n->L.size();
    long long clk[ST_COUNT][ST_COUNT] = {{0}}, evclk[EVT_COUNT][ST_COUNT];
    bool recorded[EVT_COUNT] = {false}, started[ST_COUNT] = {true, false, false, false};
    memset(evclk, 0, sizeof(evclk));
    std::vector<int> recomputed(L, 0), evaluated(L, 0), posted(L, 0);
    for (const StepOp &o : P.ops) {
        if (o.stream < 0 || o.stream >= ST_COUNT) return "plan: bad stream";
        if (o.kind == OP_RECORD) {
            if (o.ev < 0 || o.ev >= EVT_COUNT) return "plan: bad event";
            memcpy(evclk[o.ev], clk[o.stream], sizeof(clk[0]));
            recorded[o.ev] = true;
        } else if (o.kind == OP_WAIT) {
            if (o.ev < 0 || o.ev >= EVT_COUNT) return "plan: bad event";
            const bool from_prev_step = w.tail_prev && o.ev >= EVT_SWEEP0 && o.ev < EVT_LA_A0; // (recorded by the step before: complete or on the same stream)
            if (!recorded[o.ev] && !from_prev_step) return "plan: an event is waited for before it is recorded";
            for (int s = 0; s < ST_COUNT; ++s) clk[o.stream][s] = std::max(clk[o.stream][s], evclk[o.ev][s]);
            started[o.stream] = true;
        } else {
            if (!started[o.stream]) return "plan: a stream launches without having been forked";
            clk[o.stream][o.stream] += 1;
            if (o.kind == OP_RECOMPUTE) recomputed[o.layer] += 1;
            if (o.kind == OP_POOL_EVAL) evaluated[o.layer] += 1;
            if (o.kind == OP_POST) {
                posted[o.layer] += 1;
                if (o.pair > 0) { if (o.pair >= L) return "plan: bad k_post pair"; posted[o.pair] += 1; }
                if (o.layer + 1 < L && n->L[o.layer + 1].kind == AEV_KIND_POOL && pool_eval_in_post(n, o.layer + 1) && !n->dense && !o.no_inline) evaluated[o.layer + 1] += 1;
            }
        }
    }
    for (int s = 1; s < ST_COUNT; ++s)
        if (clk[ST_CHAIN][s] != clk[s][s]) return "plan: the step does not end joined";
    for (int li = 1; li < L; ++li) {
        const bool in_tail = P.tail_from > 0 && li >= P.tail_from;
        if (n->L[li].kind == AEV_KIND_CONV) {
            const int want = (in_tail ? (P.defers_tail ? 0 : 1) + (w.tail_prev ? 1 : 0) : 1);
            if (recomputed[li] != want) return "plan: a conv layer is not recomputed exactly once";
            if (posted[li] != want) return "plan: a conv layer's k_post does not run exactly once";
        } else if (evaluated[li] != (in_tail ? (P.defers_tail ? 0 : 1) + (w.tail_prev ? 1 : 0) : 1)) return "plan: a pool layer is not evaluated exactly once";
    }
    return nullptr;
}

static hipStream_t plan_stream(aev_net *n, int st) { return st == ST_SWEEP ? n->stream2 : st == ST_AHEAD ? n->stream3 : st == ST_TAIL ? n->stream4 : n->stream; }
static hipEvent_t plan_event(aev_net *n, int ev)
{
    if (ev >= EVT_LA_A0) return n->ev_la_a[ev - EVT_LA_A0];
    if (ev >= EVT_SWEEP0) return n->ev_sweep[ev - EVT_SWEEP0];
    return ev == EVT_PREP ? n->ev_prep : ev == EVT_JOIN ? n->ev_join : ev == EVT_TAIL_A ? n->ev_tail_a : n->ev_la_b;
}

static int emit_step(aev_net *n, const StepPlan &P, const int *yxt, const long long *off, const int *qstep, int s0, int ns)
{
    struct Cleared { aev_net *n; Cleared(aev_net *m) : n(m) { n->chain_cleared = true; } ~Cleared() { n->chain_cleared = false; n->cur = nullptr; } } cleared(n);
    const int L = (int)n->L.size();
    IntegArgs head_args{};
    bool head_pending = false; // an INTEG waiting for the SWEEP it shares a launch with (in_head)
    for (const StepOp &o : P.ops) {
        switch (o.kind) {
        case OP_INTEG: {
            Layer &l = n->L[0];
            IntegArgs a{};
            fill_integ_args(n, a, yxt, off, qstep, s0, o.ahead);
            a.zero = n->d_tcount; a.nzero = AEV_TC_STRIDE * std::min(std::max(o.layer, 1), L); // (the step's scratch counters: tile-list sizes, work queues)
            n->cur_layer = 0;
            if (o.in_head && n->integ_fused) { head_args = a; head_pending = true; break; } // (launched with the sweep behind it: k_head)
            if (n->integ_fused) // (one launch; k_post of the input layer moves prev_ts on, see launch_post)
                LAUNCH(n, KC_INTEG_SWEEP, k_integ_step, dim3(cdiv(l.HW, 1024), ns), dim3(256), 0, a);
            else {
                LAUNCH(n, KC_INTEG_PREP, k_integ_prepare, dim3(ns), dim3(256), 0, a);
                LAUNCH(n, KC_INTEG_SWEEP, k_integ_sweep, dim3(cdiv(l.HW, 1024), ns), dim3(256), 0, a); // (four pixels per thread)
            }
            break;
        }
        case OP_SWEEP:
            TRY(launch_leak_sweep(n, o.layer, o.layer_hi, s0, ns, o.stream == ST_CHAIN ? nullptr : plan_stream(n, o.stream), o.with_copies, o.mode, o.bpc, o.parts, o.hi_parts,
                                  o.in_head && head_pending ? &head_args : nullptr));
            if (o.in_head) head_pending = false;
            break;
        case OP_RECOMPUTE:
            n->cur = o.stream == ST_CHAIN ? nullptr : plan_stream(n, o.stream);
            TRY(launch_conv_recompute(n, o.layer, s0, ns));
            n->cur = nullptr;
            break;
        case OP_POST:
            n->cur = o.stream == ST_CHAIN ? nullptr : plan_stream(n, o.stream);
            if (o.pair > 0) TRY(launch_post_pair(n, o.layer, o.pair, s0, ns));
            else TRY(launch_post(n, o.layer, s0, ns, o.advance ? const_cast<int *>(qstep) : nullptr, o.all_streams, !o.no_inline));
            n->cur = nullptr;
            break;
        case OP_POOL_EVAL:
            n->cur = o.stream == ST_CHAIN ? nullptr : plan_stream(n, o.stream);
            TRY(launch_pool_eval(n, o.layer, 0, s0, ns));
            n->cur = nullptr;
            break;
        case OP_ZERO_COUNTERS: // layers [layer, L) of d_tcount
            if (o.layer < L) HIPCHK(hipMemsetAsync(n->d_tcount + AEV_TC_STRIDE * o.layer, 0, (size_t)AEV_TC_STRIDE * (L - o.layer) * sizeof(int), plan_stream(n, o.stream)));
            break;
        case OP_DENSE_FILL: {
            Layer &c = n->L[o.layer];
            LAUNCH(n, KC_MISC, k_fill_map, dim3(std::min(cdiv(c.HW, 256), 64), ns), dim3(256), 0, c.mapA, c.mstride, c.HW, s0, n->d_active);
            if (conv_needs_list(n, c)) LAUNCH(n, KC_MISC, k_fill_all_sites, dim3(cdiv(c.HW, 256), ns), dim3(256), 0, c.sites, c.nsites, c.lstride, c.H, c.W, s0);
            break;
        }
        case OP_RECORD: HIPCHK(hipEventRecord(plan_event(n, o.ev), plan_stream(n, o.stream))); break;
        case OP_WAIT: HIPCHK(hipStreamWaitEvent(plan_stream(n, o.stream), plan_event(n, o.ev), 0)); break;
        default: return fail(AEV_E_HIP, "internal: unknown step operation");
        }
    }
    return AEV_OK;
}

// The chain of a net with a pool layer directly behind a pool layer: compute_all (conv2d.py:139-141, maxpool.py:163-165) layer by layer
// with the launch sequences of the per-layer API -- every layer reads the event list of the layer below from device memory.  Streams
// without events in the step are left alone (k_integ_prepare's activity flags: the leak sweep and the pool layers' lists skip them; an
// idle stream's conv layers see empty lists and empty maps).
static int launch_chain_generic(aev_net *n, const int *yxt, const long long *off, const int *qstep, int s0, int ns)
{
    if (n->dense) return fail(AEV_E_UNSUPPORTED, "dense-frame mode is not available for a chain with a pool layer directly behind a pool layer");
    const int L = (int)n->L.size();
    TRY(launch_integ(n, yxt, off, qstep, s0, ns));
    if (L > 1) TRY(launch_leak_sweep(n, 1, L - 1, s0, ns, nullptr, true));
    for (int li = 1; li < L; ++li) {
        const Layer &p = n->L[li - 1];
        if (n->L[li].kind == AEV_KIND_CONV) TRY(launch_conv(n, li, p.ev, p.nev, p.lstride, p.HW, s0, ns));
        else TRY(launch_pool(n, li, p.ev, p.nev, p.lstride, p.HW, s0, ns, n->d_active));
    }
    return AEV_OK;
}

// more: another queued step of the same run follows this one directly (in the same capture, or as the next eager launch).
static int launch_chain(aev_net *n, const int *yxt, const long long *off, const int *qstep, int s0, int ns, bool more = false)
{
    if (n->generic) { n->tail_pending = n->tail_sweeps4 = false; return launch_chain_generic(n, yxt, off, qstep, s0, ns); }
    const StepWhere w{qstep != nullptr, s0 == 0 && ns == n->S, more, n->tail_pending, n->tail_sweeps4};
    n->tail_pending = false; n->tail_sweeps4 = false; // (state between two steps of a run: consumed here, set again below)
    const StepPlan P = plan_step(n, w);
    if (P.error) return fail(P.error[0] == 'i' ? AEV_E_HIP : AEV_E_UNSUPPORTED, P.error);
    if (const char *bad = check_plan(n, P, w)) return fail(AEV_E_HIP, std::string("internal: ") + bad);
    if (n->sw.plan_dump > 0) { // AEV_PLAN_DUMP=<steps>: the plans of the first so many steps, one op per line
        static const char *kn[] = {"INTEG", "SWEEP", "POST", "RECOMPUTE", "POOL_EVAL", "DENSE_FILL", "ZERO_COUNTERS", "RECORD", "WAIT"};
        static const char *sn[] = {"CHAIN", "SWEEP", "AHEAD", "TAIL"};
        fprintf(stderr, "[aev plan] queued %d whole %d more %d tail_prev %d: %zu ops, defers tail %d (from layer %d)\n", (int)w.queued, (int)w.whole, (int)w.more, (int)w.tail_prev,
                P.ops.size(), (int)P.defers_tail, P.tail_from);
        for (const StepOp &o : P.ops)
            fprintf(stderr, "[aev plan]   %-6s %-13s layer %d..%d ev %d%s\n", sn[o.stream], kn[o.kind], o.layer, o.layer_hi, o.ev, o.all_streams ? " all-streams" : "");
        const_cast<Switches &>(n->sw).plan_dump--;
    }
    TRY(emit_step(n, P, yxt, off, qstep, s0, ns));
    n->tail_pending = P.defers_tail;
    n->tail_sweeps4 = P.tail_sweeps_on_tail_stream;
    return AEV_OK;
}

// Which recompute kernel serves a conv layer, and with which tiling.  Every choice computes the same bits (one arithmetic
// contract); the choice only decides speed:
//   first3x3 : 3x3 conv directly on the integration surface (conv1)                         -> k_conv_first3x3
//   patch_cfg: 3x3 conv after a conv or a 2x2 pool-over-conv, Cin % 4 == 0, Cout >= 32      -> k_conv_patch (spatial tiles)
//   one_cfg  : 1x1 conv, Cin % 16 == 0, Cout padded to a multiple of 128                    -> k_conv_1x1 (column tiles)
//   mfma_cfg : anything else with Cout >= 32                                                  -> k_conv_mfma (site list)
//   otherwise                                                                                 -> k_conv_valu
static int configure_conv_kernels(aev_net *n, const Layer &p, Layer &l)
{
    const int kh = l.kh, kw = l.kw, cin = l.cin, cout = l.C, oh = l.H, ow = l.W, S = n->S;
    const int nb = cdiv(cout, 32);
    if (cout >= 32 && l.K >= 8) {
        if (nb == 1) { l.mfma_cfg = 0; l.gz = 1; l.CoutP = 32; }
        else if (nb == 2) { l.mfma_cfg = 1; l.gz = 1; l.CoutP = 64; }
        else if (nb <= 8) { l.mfma_cfg = 2; l.gz = cdiv(nb, 4); l.CoutP = l.gz * 128; }
        else { l.mfma_cfg = 3; l.gz = cdiv(nb, 8); l.CoutP = l.gz * 256; }
        static const int nsw[4] = {4, 2, 1, 1};
        const int kcmax = nsw[l.mfma_cfg] == 4 ? 48 : 64;
        l.KC = 0;
        for (int d = kcmax; d >= 8; d -= 2) if (l.K % d == 0) { l.KC = d; break; }
        if (!l.KC) l.KC = std::min(kcmax, cdiv(l.K, 2) * 2);
        l.Krows = cdiv(l.K, l.KC) * l.KC;
    } else {
        l.mfma_cfg = -1; l.CoutP = cdiv(cout, 32) * 32; l.KC = 0; l.Krows = l.K;
    }
    l.TS = 16;
    while (l.TS > 1 && (size_t)2 * l.K * l.TS * sizeof(float) > 48 * 1024) l.TS >>= 1;
    if ((size_t)2 * l.K * l.TS * sizeof(float) > 60 * 1024) return fail(AEV_E_SHAPE, "cin*kh*kw too large");
    l.first3x3 = p.kind == AEV_KIND_INTEG && kh == 3 && kw == 3 && cout <= 1024;
    const bool padding_ok_1x1 = true; // a 1x1 kernel has no halo: VALID == SAME
    int psrc = -1; // the tile kernel is specialised on how the previous layer is read
    if (p.kind == AEV_KIND_CONV) psrc = 1;
    else if (p.kind == AEV_KIND_POOL && p.kh == 2 && p.kw == 2 && p.stride == 2 && n->L[n->L.size() - 2].kind == AEV_KIND_CONV) psrc = 0;
    else if (p.kind == AEV_KIND_POOL && p.pfm && n->L[n->L.size() - 2].kind == AEV_KIND_POOL) psrc = 0; // (a pool stack: read through the top copy)
    if (kh == 1 && kw == 1 && cin % 16 == 0 && l.mfma_cfg >= 0 && l.CoutP % 128 == 0 && psrc >= 0 && padding_ok_1x1) { // 1x1 MFMA kernel
        l.one_cfg = cin % 32 == 0 ? 32 : 16; l.patch_src = psrc; // 32 channels per chunk when possible: half the hand-overs, twice the time for a load to land
        for (int rs = 0; rs < (l.one_cfg == 32 ? 2 : 1); ++rs)
            HIPCHK(hipFuncSetAttribute(one_kernel(l.one_cfg, rs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)one_lds_bytes(l.one_cfg, rs)));
    }
    if (kh == 3 && kw == 3 && cin % 4 == 0 && l.mfma_cfg >= 0 && psrc >= 0) { // spatial-tile kernel
        l.patch_src = psrc;
        const int *cc = g_patch_cc, *cwv = g_patch_cw, *acap = g_patch_acap;
        int pc;
        // 32 couts: 8-channel chunks, 3 blocks per CU (measured 525 -> 439 us vs 16-channel chunks), the waves as (site group, 16-cout
        // half) -- instance 6; 64 couts: instance 2; more: instance 4 (one block per CU); cin % 8 != 0: the 4-channel-chunk instances
        if (nb == 1) pc = cin % 8 == 0 ? 6 : 1;
        else if (nb == 2) pc = cin % 8 == 0 ? 2 : 3;
        else pc = cin % 8 == 0 ? 4 : -1;
        // AEV_MATH_BF16X6: the split-bf16 kernel wherever its shape holds (16-channel chunks; 32, 64 or a multiple of 128 couts, Cout % 4 == 0)
        const bool x6_ok = n->conv_math == AEV_MATH_BF16X6 && cin % 16 == 0 && cout % 4 == 0 && (l.CoutP == 32 || l.CoutP == 64 || l.CoutP % 128 == 0);
        if (x6_ok) pc = l.CoutP <= 64 ? AEV_X6_CFG0 + 1 : AEV_X6_CFG0;
        double best = 1e30;
        int tile_cap = 1 << 30; // (few streams: at most so many sites per tile, see below)
        auto pick_tile = [&](int cfg) {
            best = 1e30;
            for (int th = 1; th <= std::min(oh, 64); ++th)
                for (int tw = 1; tw <= std::min(ow, 64); ++tw) {
                    const int sites = th * tw, pp = (th + 2) * (tw + 2);
                    if (sites > std::min(std::min(acap[cfg], 256), tile_cap) || tw + 2 > AEV_PW || (th + 2) * AEV_PW > 256) continue; // LDS patch: row stride AEV_PW, 256 slots
                    if (cfg == AEV_X6_CFG0 + 1 && cin > 16 && th > 9) continue; // (two blocks of that instance per CU: two patch images of <= 11 rows each, 74 KB)
                    if ((long long)cdiv(oh, th) * cdiv(ow, tw) > 1024) continue; // k_post keeps a stream's tile masks (32 B per tile) in LDS
                    // MFMA columns + halo + a per-tile fixed cost (chunk hand-overs, compaction): measured on conv3, 64-site tiles 595 us,
                    // 112-site tiles 533-557 us
                    // (exact for the ragged tiles at the bottom / right edge: 40 rows in tiles of 9 are 4 x 126 + 1 x 56 sites,
                    //  fewer padded site blocks than 5 x 112; measured on conv3: 9x14 beats 8x14 by 1 % of the step)
                    auto tile_cost = [&](int h, int w) { return cdiv(h * w, 32) * 32 + 0.15 * (h + 2) * (w + 2) + 64.0; };
                    const int fy = oh / th, ry = oh % th, fx = ow / tw, rx = ow % tw;
                    double cost = (double)fy * fx * tile_cost(th, tw);
                    if (ry) cost += (double)fx * tile_cost(ry, tw);
                    if (rx) cost += (double)fy * tile_cost(th, rx);
                    if (ry && rx) cost += tile_cost(ry, rx);
                    (void)pp;
                    if (cost < best) { best = cost; l.TH = th; l.TW = tw; }
                }
        };
        { // tests: AEV_CFG_<layer index>=<instance> overrides the instance choice (every instance computes the same bits)
            char key[32]; snprintf(key, sizeof(key), "AEV_CFG_%d", (int)n->L.size());
            const char *ov = getenv(key);
            if (ov && atoi(ov) >= 0 && atoi(ov) < AEV_X6_CFG0 && cin % cc[atoi(ov)] == 0 && l.CoutP % cwv[atoi(ov)] == 0 && !x6_ok) pc = atoi(ov);
        }
        if (pc >= 0) pick_tile(pc);
        // Latency mode: with few streams a launch cannot fill the chip, and what its duration then follows is the longest chain of
        // MFMAs ONE wave has to issue -- (sites of its blocks) x (couts of its block) x K.  So the work item shrinks to the smallest
        // the kernel has: 32 couts per block, tiles of at most 64 sites = ONE 16-site block per wave (v_mfma_f32_16x16x4_f32),
        // 16-channel chunks (a chunk's MFMAs, 2.2 us, then cover the next chunk's loads).  One stream, round 3: conv5 44 -> 26 us.
        {
            const long long blocks = best < 1e30 ? (long long)S * cdiv(oh, l.TH) * cdiv(ow, l.TW) * (l.CoutP / cwv[pc >= 0 ? pc : 0]) : 0;
            const bool few = pc >= 0 && pc < AEV_X6_CFG0 && best < 1e30 && cin % 8 == 0 && (n->tuning == 2 || (n->tuning == 0 && blocks < 192));
            if (few) {
                pc = cin % 16 == 0 ? 0 : 5; // 32 couts per block, 16 (or 8) channels per chunk
                if (S <= 16) tile_cap = 64; // (tiles of <= 64 sites up to 16 streams; measured: 1 stream -10 %, 8 -5 %, 32 / 64 streams +3 %)
                // ONE stream: instance 7 with tiles of 32 sites -- a wave's chain is one 16-site block x 16 couts, half of instance 0's (whose
                // 64-site tiles leave every wave 32 couts): 0.174 -> 0.165 ms per step; 2 / 3 streams +-0, 4 / 16 streams +3 % (AEV_HALF_MAX)
                if (S <= n->sw.half_max && pc == 0) { pc = 7; tile_cap = 32; }
                pick_tile(pc);
            }
        }
        if (best < 1e30) {
            l.patch_cfg = pc; l.tilesY = cdiv(oh, l.TH); l.tilesX = cdiv(ow, l.TW); l.patch_gz = l.CoutP / cwv[pc];
            // more than the default 64 KB of dynamic LDS: must be allowed per kernel
            HIPCHK(hipFuncSetAttribute(patch_kernel(pc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)patch_lds_bytes(pc)));
            l.tcap = S * l.tilesX * l.tilesY; // (segments by fill: AEV_TL_BUCKETS)
            TRY(dalloc(n, &l.tlist, (size_t)AEV_TL_BUCKETS * l.tcap)); TRY(dalloc(n, &l.tmask, (size_t)AEV_TL_BUCKETS * l.tcap * 4));
            if (n->L.size() >= 64) return fail(AEV_E_UNSUPPORTED, "more than 64 layers");
            l.tcount = n->d_tcount + AEV_TC_STRIDE * n->L.size();
            int nbk = 1, ncu = 256;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, patch_kernel(pc), g_patch_nt[pc], patch_lds_bytes(pc, l.TH, cin / 16)));
            if (pc >= AEV_X6_CFG0) nbk = std::min(nbk, (int)(160 * 1024 / patch_lds_bytes(pc, l.TH, cin / 16))); // (tools/probes/lds_occupancy.hip: the API over-counts at the edge)
            HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, n->device));
            l.patch_resident = std::max(1, nbk) * ncu;
            // the tile list by fill, largest tiles first, where a block takes few items: at most ~12 of the launch's (all tiles active).
            // With many small items per block the order of the appends is kept: consecutive items are then neighbouring tiles of one
            // stream, whose patches share lines (conv2 of the e-fcn: blocks 8 % slower on the sorted list).  AEV_TL_SEGMENTS=<n> overrides.
            // (k_post pays per segment: 8 where a block takes up to a dozen items, 32 where it takes up to six)
            const long long ipb = ((long long)l.tcap * l.patch_gz + l.patch_resident - 1) / l.patch_resident;
            // (an item per CU at most: nothing to balance -- two blocks on a CU already are: conv3 at 16 streams, 320 items on 512 blocks,
            //  35 us sorted against 47; few streams: k_post is most of a step, and its append pays per segment)
            l.tl_nb = (long long)l.tcap * l.patch_gz <= ncu ? 1 : (ipb <= 6 && S >= 128) ? 32 : ipb <= 12 ? 8 : 1;
            if (n->sw.tl_segments > 0) l.tl_nb = n->sw.tl_segments;
            if (n->sw.verbose) {
                fprintf(stderr, "[aev] layer %d: tile kernel cfg %d src %d, tile %dx%d (%dx%d tiles), %zu B LDS, %d blocks per CU, %d list segments\n", (int)n->L.size(), pc,
                        l.patch_src, l.TH, l.TW, l.tilesY, l.tilesX, patch_lds_bytes(pc, l.TH, cin / 16), nbk, l.tl_nb);
            }
        }
    }
    return AEV_OK;
}


// AEV_STAMPS diagnostic: per-phase cycles of the tile kernel's work items and the schedule of its blocks (last launch of the layer)
static void dump_stamps(aev_net *n)
{
    std::vector<unsigned long long> h(n->stamps_cnt);
    hipMemcpy(h.data(), n->d_stamps, n->stamps_cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum[8] = {0}, sum6 = 0, gap_a = 0, gap_b = 0; size_t nb = 0, n6 = 0;
    for (size_t b = 0; b < (size_t)n->stamp_items; ++b)
        if (h[b * 8 + 7] > 0) {
            for (int q = 0; q < 8; ++q) sum[q] += (double)h[b * 8 + q];
            sum[5] += (double)(h[b * 8 + 5] & 0xffff) - (double)h[b * 8 + 5]; gap_a += (double)((h[b * 8 + 5] >> 16) & 0xffffff); gap_b += (double)(h[b * 8 + 5] >> 40);
            ++nb; if (h[b * 8 + 6]) { sum6 += (double)h[b * 8 + 6]; ++n6; }
        }
    const unsigned long long *blk = h.data() + (size_t)n->stamp_items * 8;
    const size_t nblk = (n->stamps_cnt - (size_t)n->stamp_items * 8) / 4;
    unsigned long long t0 = ~0ull, t1 = 0, smax = 0, dmin = ~0ull, dmax = 0; double dsum = 0, csum = 0; size_t cntb = 0, early = 0;
    for (size_t b = 0; b < nblk; ++b) if (blk[b * 4]) t0 = std::min(t0, blk[b * 4]);
    std::map<unsigned long long, int> per_cu; // early blocks per (XCC, SE, SH, CU)
    for (size_t b = 0; b < nblk; ++b) if (blk[b * 4]) {
        const unsigned long long st = blk[b * 4] - t0, en = blk[b * 4 + 1] - t0, d = en - st;
        early += st < 2000; smax = std::max(smax, st); t1 = std::max(t1, en); dsum += (double)d; dmin = std::min(dmin, d); dmax = std::max(dmax, d); ++cntb;
        csum += (double)blk[b * 4 + 2];
        const unsigned long long hw = blk[b * 4 + 3];
        if (d > 2000) per_cu[((hw >> 32) & 0xf) << 16 | (hw & 0xff00)]++; // XCC_ID[3:0]; HW_ID: cu_id[11:8], sh_id[12], se_id[15:13]
    }
    if (cntb) fprintf(stderr, "[AEV_STAMPS] %zu blocks (%zu started within 20 us): last start +%.1f us, kernel span %.1f us, block duration min %.1f mean %.1f max %.1f us; shader clock %.0f MHz\n",
                      cntb, early, smax / 100.0, t1 / 100.0, dmin / 100.0, dsum / cntb / 100.0, dmax / 100.0, dsum > 0 ? csum / dsum * 100.0 : 0.0);
    if (!per_cu.empty()) { // how many working blocks (> 20 us) each CU held
        int hist[8] = {0};
        for (auto &kv : per_cu) hist[std::min(kv.second, 7)]++;
        fprintf(stderr, "[AEV_STAMPS] %zu CUs held working blocks; CUs with 1 / 2 / 3 / 4 / 5+ of them: %d / %d / %d / %d / %d\n", per_cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7]);
    }
    if (cntb) { // when the blocks became resident
        std::vector<unsigned long long> st;
        for (size_t b = 0; b < nblk; ++b) if (blk[b * 4]) st.push_back(blk[b * 4] - t0);
        std::sort(st.begin(), st.end());
        fprintf(stderr, "[AEV_STAMPS] start of the 10th / 25th / 50th / 75th / 90th percentile block: +%.1f / %.1f / %.1f / %.1f / %.1f us\n", st[st.size() / 10] / 100.0,
                st[st.size() / 4] / 100.0, st[st.size() / 2] / 100.0, st[st.size() * 3 / 4] / 100.0, st[st.size() * 9 / 10] / 100.0);
    }
    if (nb) fprintf(stderr, "[AEV_STAMPS] %zu items, cycles per item (wave 0): next item's site table %.0f, first chunk staging %.0f, gemm + hand-over %.0f, chunk barriers %.0f, "
                            "epilogue %.0f; top of the item (requests) %.0f, dispatch + offsets %.0f; mean site blocks per item %.2f, per wave 0 %.2f; item top to next item top %.0f (items that are not a block's last)\n", nb, sum[4] / nb, sum[0] / nb, sum[1] / nb, sum[2] / nb, sum[3] / nb, gap_a / nb, gap_b / nb, sum[5] / nb, sum[7] / nb, n6 ? sum6 / n6 : 0.0);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *aev_last_error(void) { return g_err.c_str(); }
int aev_version(void) { return 100; }
const char *aev_kernel_name(int k) { return k >= 0 && k < AEV_NUM_KERNELS ? g_kernel_names[k] : ""; }

int aev_device_count(int *n)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) { *n = 0; return fail(AEV_E_HIP, "no HIP device available: libaev_hip.so needs an MI355X"); }
    *n = c;
    return AEV_OK;
}

// Which HIP runtime this process has mapped, against the one the library was compiled for.  libamdhip64.so.7 is one SONAME for
// every ROCm 7.x: a process that has imported PyTorch first runs this library on torch's bundled runtime, whatever hipcc built it.
int aev_runtime_info(int *hip_compiled, int *hip_runtime, int *hip_driver, char *runtime_path, int path_capacity)
{
    if (hip_compiled) *hip_compiled = HIP_VERSION;
    int v = 0;
    if (hip_runtime) { *hip_runtime = hipRuntimeGetVersion(&v) == hipSuccess ? v : -1; }
    v = 0;
    if (hip_driver) { *hip_driver = hipDriverGetVersion(&v) == hipSuccess ? v : -1; }
    if (runtime_path && path_capacity > 0) {
        Dl_info di;
        runtime_path[0] = 0;
        if (dladdr((const void *)&hipRuntimeGetVersion, &di) && di.dli_fname) { strncpy(runtime_path, di.dli_fname, (size_t)path_capacity - 1); runtime_path[path_capacity - 1] = 0; }
    }
    return AEV_OK;
}

static int select_device(int device);
int aev_device_sync(int device)
{
    TRY(select_device(device));
    HIPCHK(hipDeviceSynchronize());
    return AEV_OK;
}

static int select_device(int device)
{
    int c = 0;
    TRY(aev_device_count(&c));
    if (device < 0 || device >= c || device >= AEV_MAX_DEVICES) return fail(AEV_E_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    return AEV_OK;
}

// HIP streams are recycled through a process-wide pool and never destroyed.  Destroying a handle's streams and creating the
// next handle's right away made the first hipGraphLaunch of the new handle's captured step crash inside the runtime (ROCm 7.0:
// deterministic for a given build and test order, gone as soon as any one of the three hipStreamDestroy calls is left out).
// A synchronised stream is as good as a new one.  The pool is keyed by the device index and guarded by a mutex: distinct
// handles may be driven from distinct threads (include/aev.h).
static std::mutex g_pool_mutex;
static std::vector<hipStream_t> g_stream_pool[AEV_MAX_DEVICES];
static int stream_acquire(int device, hipStream_t *st)
{
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        std::vector<hipStream_t> &pool = g_stream_pool[device];
        if (!pool.empty()) { *st = pool.back(); pool.pop_back(); return AEV_OK; }
    }
    HIPCHK(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    return AEV_OK;
}
static void stream_release(int device, hipStream_t st)
{
    if (!st) return;
    hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_stream_pool[device].push_back(st);
}

// Every kernel a captured step may contain is resolved once per device, outside any capture.  HIP loads code objects lazily,
// at a kernel's first use; a kernel whose first use in the process was a launch INSIDE a stream capture (k_leak_sweep_ahead in
// a test-suite run whose first queued many-stream run was a graph replay) made the first hipGraphLaunch of that graph crash
// inside the runtime (ROCm 7.0).
static std::once_flag g_warm_once[AEV_MAX_DEVICES];
static int warm_kernels_body()
{
    const void *fns[] = {
        (const void *)k_integ_prepare, (const void *)k_integ_peek, (const void *)k_integ_sweep, (const void *)k_integ_step, (const void *)k_head, (const void *)k_leak_sweep, (const void *)k_leak_sweep_ahead,
        (const void *)k_post, (const void *)k_post_pair, (const void *)k_post2<AEV_POST2_NT_FEW>, (const void *)k_post2_pair<AEV_POST2_NT_FEW>, (const void *)k_post2<AEV_POST2_NT_MANY>, (const void *)k_post2_pair<AEV_POST2_NT_MANY>, (const void *)k_compact, (const void *)k_conv_expand, (const void *)k_conv_valu, (const void *)k_conv_first3x3,
        (const void *)k_conv_mfma<4, 1, 1>, (const void *)k_conv_mfma<2, 2, 1>, (const void *)k_conv_mfma<1, 4, 1>, (const void *)k_conv_mfma<1, 4, 2>,
        (const void *)k_pool_eval_vec4, (const void *)k_pool_eval<true>, (const void *)k_pool_eval<false>,
        (const void *)k_pool_fill, (const void *)k_touch_mark, (const void *)k_touch_list, (const void *)k_tile_list, (const void *)k_getter,
        (const void *)k_fill_all_sites, (const void *)k_fill_map, (const void *)k_advance, (const void *)k_queue_begin, (const void *)k_fill_int};
    for (const void *f : fns) {
        hipFuncAttributes at;
        HIPCHK(hipFuncGetAttributes(&at, f));
    }
    for (int cfg = 0; cfg < AEV_NUM_PATCH_CFG; ++cfg) {
        hipFuncAttributes at;
        HIPCHK(hipFuncGetAttributes(&at, patch_kernel(cfg)));
    }
    for (int cc : {16, 32})
        for (int rs = 0; rs < (cc == 32 ? 2 : 1); ++rs) {
            hipFuncAttributes at;
            HIPCHK(hipFuncGetAttributes(&at, one_kernel(cc, rs)));
        }
    return AEV_OK;
}
static int warm_kernels(int device)
{
    int rc = AEV_OK;
    std::call_once(g_warm_once[device], [&]() { rc = warm_kernels_body(); }); // (a failure is reported by the first caller only; the launches that follow fail on their own)
    return rc;
}

int aev_net_create(int H, int W, double leak, int n_streams, int device, aev_net **out)
{
    if (!out) return fail(AEV_E_ARG, "out is NULL");
    if (H <= 0 || W <= 0 || H > 32767 || W > 32767) return fail(AEV_E_SHAPE, "surface size must be in 1..32767");
    if (n_streams <= 0 || n_streams > 65535) return fail(AEV_E_ARG, "n_streams must be in 1..65535");
    TRY(select_device(device));
    TRY(warm_kernels(device));
    aev_net *n = new aev_net();
    n->sw = Switches::from_env();
    n->H = H; n->W = W; n->S = n_streams; n->device = device; n->leak = leak;
    int rc = AEV_OK;
    auto body = [&]() -> int {
        HIPCHK(hipDeviceGetAttribute(&n->ncu, hipDeviceAttributeMultiprocessorCount, n->device));
        TRY(stream_acquire(n->device, &n->stream));
        TRY(stream_acquire(n->device, &n->stream2));
        TRY(stream_acquire(n->device, &n->stream3));
        TRY(stream_acquire(n->device, &n->stream4));
        HIPCHK(hipEventCreateWithFlags(&n->ev_prep, hipEventDisableTiming));
        HIPCHK(hipFuncSetAttribute((const void *)k_post, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024)); // LDS tables, see launch_post
        HIPCHK(hipFuncSetAttribute((const void *)k_post_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
        // The two plans of a queued step (plan_step): from 208 streams on the sweeps run on a second stream, a step ahead where they
        // can, and the 1x1 tail beside the next step's head (round 2); below, the bookkeeping runs ahead of the value chain and a step's
        // back half is a pipeline stage beside the next step's front half (round 3).
        // Measured ms per step at the end of round 3, fork plan / few-streams plan: 96 streams 0.977 / 0.888, 128 1.123 / 1.110,
        // 160 1.459 / 1.345, 192 1.593 / 1.534, 224 1.748 / 1.772, 256 1.907 / 2.023, 320 2.504 / 2.504.
        n->overlap = n->sw.overlap >= 0 ? n->sw.overlap : n->S >= 208;
        n->tuning = n->sw.tuning;
        const int S = n->S;
        TRY(dalloc(n, &n->d_prev_ts, S)); TRY(dalloc(n, &n->d_last_ts, S)); TRY(dalloc(n, &n->d_dl64, S));
        TRY(dalloc(n, &n->d_dlf, S)); TRY(dalloc(n, &n->d_active, S)); TRY(dalloc(n, &n->d_off, S + 1));
        TRY(dalloc(n, &n->d_qrun, 2)); HIPCHK(hipMemsetAsync(n->d_qrun, 0, 2 * sizeof(int), n->stream));
        TRY(dalloc(n, &n->d_dlf_next, S)); TRY(dalloc(n, &n->d_active_next, S));
        HIPCHK(hipMemsetAsync(n->d_dlf_next, 0, S * sizeof(float), n->stream)); HIPCHK(hipMemsetAsync(n->d_active_next, 0, S * sizeof(int), n->stream));
        HIPCHK(hipEventCreateWithFlags(&n->ev_la_b, hipEventDisableTiming));
        // On whenever the second stream is (>= 64 streams); AEV_LOOKAHEAD=0 / 1 forces it off / on.  Measured on the e-fcn at 256
        // streams: 2.16 -> 2.09 ms per step.  A sweep wave beside the MFMA waves of the one-block-per-CU tile kernels gets few
        // issue slots (6 us per 3072-element chunk beside conv5, 1.5 us beside conv3), and costs those kernels ~7 %.
        n->lookahead = n->sw.lookahead >= 0 ? n->sw.lookahead != 0 : n->overlap != 0;
        n->tail_overlap = n->sw.tail_overlap >= 0 ? n->sw.tail_overlap != 0 : n->lookahead;
        TRY(dalloc(n, &n->d_qstep, 1)); TRY(dalloc(n, &n->d_ninj, S)); TRY(dalloc(n, &n->d_stats, 64)); TRY(dalloc(n, &n->d_tcount, 64 * AEV_TC_STRIDE)); HIPCHK(hipMemsetAsync(n->d_tcount, 0, 64 * AEV_TC_STRIDE * sizeof(int), n->stream));
        HIPCHK(hipMemsetAsync(n->d_prev_ts, 0, S * sizeof(long long), n->stream)); HIPCHK(hipMemsetAsync(n->d_last_ts, 0, S * sizeof(long long), n->stream));
        HIPCHK(hipMemsetAsync(n->d_dl64, 0, S * sizeof(double), n->stream)); HIPCHK(hipMemsetAsync(n->d_dlf, 0, S * sizeof(float), n->stream));
        HIPCHK(hipMemsetAsync(n->d_active, 0, S * sizeof(int), n->stream)); HIPCHK(hipMemsetAsync(n->d_qstep, 0, sizeof(int), n->stream));
        HIPCHK(hipMemsetAsync(n->d_ninj, 0, S * sizeof(int), n->stream)); HIPCHK(hipMemsetAsync(n->d_stats, 0, 64 * sizeof(u64), n->stream));
        Layer l;
        l.kind = AEV_KIND_INTEG; l.C = 1; l.H = H; l.W = W; l.HW = H * W; l.CHW = l.HW;
        l.sstride = (long long)cdiv(l.CHW, 64) * 64; l.mstride = (long long)cdiv(l.HW, 64) * 64; l.lstride = l.HW;
        TRY(dalloc(n, &l.surf, (size_t)S * l.sstride)); TRY(dalloc(n, &l.owner, (size_t)S * l.HW));
        TRY(dalloc(n, &l.mapA, (size_t)S * l.mstride)); TRY(dalloc(n, &l.ev, (size_t)S * l.HW)); TRY(dalloc(n, &l.nev, S));
        HIPCHK(hipMemsetAsync(l.surf, 0, (size_t)S * l.sstride * sizeof(float), n->stream));
        HIPCHK(hipMemsetAsync(l.owner, 0xff, (size_t)S * l.HW * sizeof(int), n->stream));
        HIPCHK(hipMemsetAsync(l.mapA, 0, (size_t)S * l.mstride, n->stream));
        HIPCHK(hipMemsetAsync(l.nev, 0, S * sizeof(int), n->stream));
        n->L.push_back(l);
        { // (see warm_kernels) one real launch of the look-ahead sweep outside any capture: with an empty run it returns at once
            SweepArgs a{};
            memset(&a, 0, sizeof(a));
            a.qstep = n->d_qstep; a.qrun = n->d_qrun; a.dlf = n->d_dlf_next; a.active = n->d_active_next;
            hipLaunchKernelGGL(k_leak_sweep_ahead, dim3(1), dim3(256), 0, n->stream, a);
            HIPCHK(hipStreamSynchronize(n->stream));
        }
        return AEV_OK;
    };
    rc = body();
    if (rc < 0) { std::string keep = g_err; aev_net_destroy(n); g_err = keep; return rc; }
    *out = n;
    return AEV_OK;
}

void aev_net_destroy(aev_net *n)
{
    if (!n) return;
    hipSetDevice(n->device);
    if (n->stream) hipStreamSynchronize(n->stream);
    if (n->stream2) hipStreamSynchronize(n->stream2);
    if (n->stream3) hipStreamSynchronize(n->stream3);
    if (n->stream4) hipStreamSynchronize(n->stream4);
    // the captured graphs first: they were built from these events and this second stream
    for (auto &g : n->graphs) { hipGraphExecDestroy(g.exec); hipGraphDestroy(g.src); }
    n->graphs.clear();
    if (n->ev_prep) hipEventDestroy(n->ev_prep);
    for (hipEvent_t e : n->ev_la_a) if (e) hipEventDestroy(e);
    if (n->ev_la_b) hipEventDestroy(n->ev_la_b);
    if (n->ev_join) hipEventDestroy(n->ev_join);
    if (n->ev_tail_a) hipEventDestroy(n->ev_tail_a);
    for (hipEvent_t e : n->ev_sweep) if (e) hipEventDestroy(e);
    stream_release(n->device, n->stream2);
    stream_release(n->device, n->stream3);
    stream_release(n->device, n->stream4);
    for (void *p : n->allocs) hipFree(p);
    if (n->d_yxt) hipFree(n->d_yxt);
    for (auto &sl : n->slot) {
        if (sl.h_yxt) hipHostFree(sl.h_yxt);
        if (sl.h_off) hipHostFree(sl.h_off);
        if (sl.d_yxt) hipFree(sl.d_yxt);
        if (sl.d_off) hipFree(sl.d_off);
        if (sl.done) hipEventDestroy(sl.done);
    }
    if (n->d_qyxt) hipFree(n->d_qyxt);
    if (n->d_qoff) hipFree(n->d_qoff);
    if (n->d_inj) hipFree(n->d_inj);
    if (n->d_scratch) hipFree(n->d_scratch);
    if (n->d_stamps) hipFree(n->d_stamps);
    if (n->d_pstamps) hipFree(n->d_pstamps);
    stream_release(n->device, n->stream);
    delete n;
}

static void drop_graph(aev_net *n)
{
    for (auto &g : n->graphs) { hipGraphExecDestroy(g.exec); hipGraphDestroy(g.src); }
    n->graphs.clear();
}

int aev_net_add_conv(aev_net *n, const float *kernel_hwio, const float *bias, int kh, int kw, int cin, int cout,
                     int stride, double alpha, int padding)
{
    if (!n || !kernel_hwio) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    const Layer &p = n->L.back();
    if (kh <= 0 || kw <= 0 || kh > 255 || kw > 255 || cout <= 0) return fail(AEV_E_SHAPE, "bad kernel shape");
    if (cin != p.C) return fail(AEV_E_SHAPE, "kernel in-channels do not match the previous layer's channels");
    if (stride != 1) return fail(AEV_E_STRIDE, "async conv supports stride 1 only (as the reference, SURVEY Q3)");
    int pt = 0, pl = 0, oh, ow;
    if (padding == AEV_PAD_VALID) {
        if (p.H < kh || p.W < kw) return fail(AEV_E_SHAPE, "kernel larger than the input map");
        oh = p.H - kh + 1; ow = p.W - kw + 1;
    } else if (padding == AEV_PAD_SAME) { // conv2d.py:38-54 with stride 1
        oh = p.H; ow = p.W;
        const int ph = kh - 1, pw = kw - 1;
        pt = ph / 2; pl = pw / 2;
        if (ph == 0 && pw > 0) return fail(AEV_E_SHAPE, "SAME padding with kh == 1 < kw is not padded by the reference (conv2d.py:69)");
    } else
        return fail(AEV_E_ARG, "'padding' must be either 'SAME' or 'VALID'");
    { SrcView chk; TRY(make_view(n, (int)n->L.size() - 1, &chk)); }
    drop_graph(n);
    const int S = n->S;
    Layer l;
    l.kind = AEV_KIND_CONV; l.C = cout; l.H = oh; l.W = ow; l.HW = oh * ow; l.CHW = cout * l.HW;
    // (+ one site behind the map: the sink the MFMA kernels' scatter sends the padding rows of a tile's last site block to)
    l.sstride = (long long)cdiv(l.CHW + cdiv(cout, 32) * 32, 64) * 64; l.mstride = (long long)cdiv(l.HW, 64) * 64; l.lstride = l.HW;
    l.kh = kh; l.kw = kw; l.cin = cin; l.K = cin * kh * kw; l.pt = pt; l.pl = pl; l.stride = 1; l.alpha = (float)alpha;
    TRY(configure_conv_kernels(n, p, l));
    TRY(dalloc(n, &l.fm, (size_t)S * l.sstride)); TRY(dalloc(n, &l.act, (size_t)S * l.sstride));
    TRY(dalloc(n, &l.init_fm, (size_t)S * l.sstride));
    TRY(dalloc(n, &l.wt, (size_t)(l.Krows + 64) * l.CoutP)); // + 64 rows: the tile kernel's last 16-byte pass over a weight slice may read past it (eight-wave instances: up to 56 rows)
    TRY(dalloc(n, &l.bias, l.CoutP));
    TRY(dalloc(n, &l.mapA, (size_t)S * l.mstride)); TRY(dalloc(n, &l.mapB, (size_t)S * l.mstride));
    if ((cout & (cout - 1)) == 0 && cout >= 4 && cout <= 256) { TRY(dalloc(n, &l.dflag, (size_t)S * l.mstride)); HIPCHK(hipMemsetAsync(l.dflag, 1, (size_t)S * l.mstride, n->stream)); } // "may decay" flags (SweepSeg::flag)
    TRY(dalloc(n, &l.ev, (size_t)S * l.HW)); TRY(dalloc(n, &l.nev, S));
    TRY(dalloc(n, &l.sites, (size_t)S * l.HW)); TRY(dalloc(n, &l.nsites, S));
    HIPCHK(hipMemsetAsync(l.fm, 0, (size_t)S * l.sstride * sizeof(float), n->stream));
    HIPCHK(hipMemsetAsync(l.act, 0, (size_t)S * l.sstride * sizeof(float), n->stream));
    HIPCHK(hipMemsetAsync(l.mapA, 0, (size_t)S * l.mstride, n->stream));
    HIPCHK(hipMemsetAsync(l.mapB, 0, (size_t)S * l.mstride, n->stream));
    HIPCHK(hipMemsetAsync(l.nev, 0, S * sizeof(int), n->stream)); HIPCHK(hipMemsetAsync(l.nsites, 0, S * sizeof(int), n->stream));
    // weights: [kh,kw,cin,cout] -> Wt[position in the fmaf chain (aev_device.hpp: chain_pos)][cout], zero padded (conv2d.py:26 keeps [cout][k])
    std::vector<float> wt((size_t)l.Krows * l.CoutP, 0.f), bb(l.CoutP, 0.f);
    for (int dy = 0; dy < kh; ++dy)
        for (int dx = 0; dx < kw; ++dx)
            for (int c = 0; c < cin; ++c)
                for (int co = 0; co < cout; ++co)
                    wt[(size_t)chain_pos(c, dy * kw + dx, cin, kh * kw) * l.CoutP + co] = kernel_hwio[(((size_t)dy * kw + dx) * cin + c) * cout + co];
    if (bias) for (int co = 0; co < cout; ++co) bb[co] = bias[co];
    HIPCHK(hipMemcpy(l.wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(l.bias, bb.data(), bb.size() * sizeof(float), hipMemcpyHostToDevice));
    if (l.patch_cfg >= AEV_X6_CFG0) { // the x6 kernel's weights: three bf16 parts of every weight (exact: 8 + 8 + 8 bits of the significand), in its k order
        const int nch = cin / 16;
        const size_t wpart = (size_t)nch * AEV_X6_KB * l.CoutP * 32;
        std::vector<unsigned short> w3(3 * wpart, 0);
        auto bits = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
        auto flt = [](unsigned u) { float f; memcpy(&f, &u, 4); return f; };
        for (int c = 0; c < nch; ++c)
            for (int p = 0; p < AEV_X6_KB; ++p)
                for (int co = 0; co < cout; ++co)
                    for (int gq = 0; gq < 4; ++gq) {
                        const int tap = x6_tap(p, gq >> 1);
                        if (tap < 0) continue; // (the tenth tap slot: zeros)
                        for (int e = 0; e < 8; ++e) {
                            const int ch = 16 * c + 8 * (gq & 1) + e;
                            const float w = kernel_hwio[((size_t)tap * cin + ch) * cout + co];
                            const unsigned hi = bits(w) & 0xffff0000u;
                            const float r = w - flt(hi);
                            const unsigned mid = bits(r) & 0xffff0000u;
                            const unsigned lo = bits(r - flt(mid));
                            const size_t at = (((size_t)(c * AEV_X6_KB + p)) * l.CoutP + co) * 32 + gq * 8 + e;
                            w3[at] = (unsigned short)(hi >> 16); w3[wpart + at] = (unsigned short)(mid >> 16); w3[2 * wpart + at] = (unsigned short)(lo >> 16);
                        }
                    }
        TRY(dalloc(n, &l.wt3, w3.size()));
        HIPCHK(hipMemcpy(l.wt3, w3.data(), w3.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    n->L.push_back(l);
    choose_plan(n); n->la_gate = lookahead_layers(n, &n->la_state); tail_layers(n); tail_layers_few(n);
    const int li = (int)n->L.size() - 1;
    // initial state: dense conv of the previous layer's current feature map (conv2d.py:59-61)
    hipLaunchKernelGGL(k_fill_all_sites, dim3(cdiv(l.HW, 256), S), dim3(256), 0, n->stream, l.sites, l.nsites, l.lstride, l.H, l.W, 0);
    HIPCHK(hipMemsetAsync(l.mapA, 1, (size_t)S * l.mstride, n->stream)); // every site active (tile kernel reads the map)
    int rc = launch_conv_recompute(n, li, 0, S);
    HIPCHK(hipMemsetAsync(l.mapA, 0, (size_t)S * l.mstride, n->stream));
    if (rc < 0) { n->L.pop_back(); choose_plan(n); n->la_gate = lookahead_layers(n, &n->la_state); tail_layers(n); tail_layers_few(n); return rc; }
    HIPCHK(hipMemsetAsync(l.act, 0, (size_t)S * l.sstride * sizeof(float), n->stream)); // conv2d.py:62-63
    HIPCHK(hipMemcpyAsync(l.init_fm, l.fm, (size_t)S * l.sstride * sizeof(float), hipMemcpyDeviceToDevice, n->stream));
    HIPCHK(hipMemsetAsync(l.nsites, 0, S * sizeof(int), n->stream));
    HIPCHK(hipStreamSynchronize(n->stream));
    HIPCHK(hipGetLastError());
    return li;
}

int aev_net_add_pool(aev_net *n, int kh, int kw, int stride)
{
    if (!n) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    const Layer &p = n->L.back();
    if (kh <= 0 || kw <= 0 || kh * kw > 255) return fail(AEV_E_SHAPE, "bad pool window");
    if (!(stride == kh && stride == kw)) return fail(AEV_E_STRIDE, "This method only support stride equal to 1 or to the kernel's dimensions.");
    if (p.H % kh || p.W % kw) return fail(AEV_E_SHAPE, "pooled dimensions must be divisible by the window (SURVEY Q7)");
    // A pool layer directly behind a pool layer (MaxPoolLayer takes any prev_layer, maxpool.py:14-40): it reads the layer below through
    // that layer's copy of its winners, so the layer below must keep one (it does whenever its own stack stands on a conv layer), and the
    // chain is stepped layer by layer from now on (aev_net::generic).
    if (p.kind == AEV_KIND_POOL && !p.pfm) return fail(AEV_E_UNSUPPORTED, "a pool layer behind a pool layer needs a conv layer at the bottom of the pool stack");
    if (p.kind == AEV_KIND_POOL && n->dense) return fail(AEV_E_UNSUPPORTED, "dense-frame mode is not available for a chain with a pool layer directly behind a pool layer");
    drop_graph(n);
    if (p.kind == AEV_KIND_POOL) { n->generic = true; n->overlap = 0; n->lookahead = false; n->tail_overlap = false; }
    const int S = n->S;
    Layer l;
    l.kind = AEV_KIND_POOL; l.C = p.C; l.H = p.H / kh; l.W = p.W / kw; l.HW = l.H * l.W; l.CHW = l.C * l.HW;
    l.sstride = (long long)cdiv(l.CHW, 64) * 64; l.mstride = (long long)cdiv(l.HW, 64) * 64; l.lstride = l.HW;
    l.kh = kh; l.kw = kw; l.stride = stride;
    TRY(dalloc(n, &l.idx, (size_t)S * l.sstride)); TRY(dalloc(n, &l.init_idx, (size_t)S * l.sstride));
    TRY(dalloc(n, &l.pending, (size_t)S * l.mstride)); TRY(dalloc(n, &l.key, (size_t)S * l.HW));
    TRY(dalloc(n, &l.ev, (size_t)S * l.HW)); TRY(dalloc(n, &l.nev, S)); TRY(dalloc(n, &l.nevw, S));
    // (every initialisation on the handle's stream: hipMemset runs on the NULL stream, which the handle's non-blocking stream
    //  does not wait for -- the plain evaluation below raced with the zeroing of the copy it writes, and once in ~1500 nets the
    //  tail of the copy ended up zeroed: a conv layer behind the pool then saw zeros there until the windows' next evaluation)
    HIPCHK(hipMemsetAsync(l.nevw, 0, (size_t)S * sizeof(int), n->stream));
    HIPCHK(hipMemsetAsync(l.idx, 0, (size_t)S * l.sstride, n->stream)); HIPCHK(hipMemsetAsync(l.pending, 0, (size_t)S * l.mstride, n->stream));
    if (p.kind == AEV_KIND_CONV || (p.kind == AEV_KIND_POOL && p.pfm)) { // the copy of the conv below at the (composed) argmax sites (pool_copy_fill)
        TRY(dalloc(n, &l.pfm, (size_t)S * l.sstride)); TRY(dalloc(n, &l.pact, (size_t)S * l.sstride));
        if ((l.C & (l.C - 1)) == 0 && l.C >= 4 && l.C <= 256) { // "may decay" flags of the copy: one byte per window
            TRY(dalloc(n, &l.dflag, (size_t)S * l.mstride)); HIPCHK(hipMemsetAsync(l.dflag, 1, (size_t)S * l.mstride, n->stream));
        }
        HIPCHK(hipMemsetAsync(l.pfm, 0, (size_t)S * l.sstride * sizeof(float), n->stream)); HIPCHK(hipMemsetAsync(l.pact, 0, (size_t)S * l.sstride * sizeof(float), n->stream));
    }
    hipLaunchKernelGGL(k_fill_int, dim3(256), dim3(256), 0, n->stream, l.key, (long long)S * l.HW, INT_MAX);
    n->L.push_back(l);
    choose_plan(n); n->la_gate = lookahead_layers(n, &n->la_state); tail_layers(n); tail_layers_few(n);
    const int li = (int)n->L.size() - 1;
    // initial argmax: first-occurrence argmax of the previous surface (maxpool.py:31-35)
    hipLaunchKernelGGL(k_fill_all_sites, dim3(cdiv(l.HW, 256), S), dim3(256), 0, n->stream, l.ev, l.nev, l.lstride, l.H, l.W, 0);
    int rc = launch_pool_eval(n, li, 1, 0, S);
    if (rc < 0) { // (the handle stays what it was before the call: a failed pool-behind-pool add must not leave it on the generic chain)
        n->L.pop_back();
        n->generic = false;
        for (int lj = 2; lj < (int)n->L.size(); ++lj) n->generic |= pool_behind_pool(n, lj);
        choose_plan(n); n->la_gate = lookahead_layers(n, &n->la_state); tail_layers(n); tail_layers_few(n);
        return rc;
    }
    HIPCHK(hipMemcpyAsync(l.init_idx, l.idx, (size_t)S * l.sstride, hipMemcpyDeviceToDevice, n->stream));
    HIPCHK(hipMemsetAsync(l.nev, 0, S * sizeof(int), n->stream));
    n->L[li].copy_dirty = false; // the plain evaluation of every window above has written the whole copy
    HIPCHK(hipStreamSynchronize(n->stream));
    HIPCHK(hipGetLastError());
    return li;
}

int aev_net_num_layers(const aev_net *n) { return n ? (int)n->L.size() : 0; }
int aev_net_num_streams(const aev_net *n) { return n ? n->S : 0; }

int aev_layer_shape(const aev_net *n, int layer, int32_t *chw)
{
    if (!n || layer < 0 || layer >= (int)n->L.size()) return fail(AEV_E_ARG, "bad layer index");
    chw[0] = n->L[layer].C; chw[1] = n->L[layer].H; chw[2] = n->L[layer].W;
    return AEV_OK;
}

static int reset_layer(aev_net *n, int li, int s0, int ns)
{
    Layer &l = n->L[li];
    hipStream_t st = n->stream;
    const size_t fo = (size_t)s0 * l.sstride, fn = (size_t)ns * l.sstride;
    if (l.kind == AEV_KIND_INTEG) {
        HIPCHK(hipMemsetAsync(n->d_prev_ts + s0, 0, ns * sizeof(long long), st)); // integration.py:49
        HIPCHK(hipMemsetAsync(l.surf + fo, 0, fn * sizeof(float), st));           // integration.py:50
    } else if (l.kind == AEV_KIND_CONV) {
        HIPCHK(hipMemcpyAsync(l.fm + fo, l.init_fm + fo, fn * sizeof(float), hipMemcpyDeviceToDevice, st)); // conv2d.py:100
        HIPCHK(hipMemsetAsync(l.act + fo, 0, fn * sizeof(float), st));                                      // conv2d.py:101
        HIPCHK(hipMemsetAsync(l.mapB + (size_t)s0 * l.mstride, 0, (size_t)ns * l.mstride, st));
        HIPCHK(hipMemsetAsync(l.nsites + s0, 0, ns * sizeof(int), st));
        if (l.dflag) HIPCHK(hipMemsetAsync(l.dflag + (size_t)s0 * l.mstride, 1, (size_t)ns * l.mstride, st));
    } else {
        HIPCHK(hipMemcpyAsync(l.idx + fo, l.init_idx + fo, fn, hipMemcpyDeviceToDevice, st));                  // maxpool.py:86
        HIPCHK(hipMemsetAsync(l.pending + (size_t)s0 * l.mstride, 0, (size_t)ns * l.mstride, st)); // maxpool.py:87
    }
    if (l.mapA) HIPCHK(hipMemsetAsync(l.mapA + (size_t)s0 * l.mstride, 0, (size_t)ns * l.mstride, st));
    HIPCHK(hipMemsetAsync(l.nev + s0, 0, ns * sizeof(int), st));
    pool_copy_touch(n, li);
    return AEV_OK;
}

int aev_layer_reset(aev_net *n, int layer, int stream)
{
    if (!n || layer < 0 || layer >= (int)n->L.size() || stream < -1 || stream >= n->S) return fail(AEV_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(n->device));
    TRY(reset_layer(n, layer, stream < 0 ? 0 : stream, stream < 0 ? n->S : 1));
    HIPCHK(hipStreamSynchronize(n->stream));
    return AEV_OK;
}

int aev_net_reset(aev_net *n, int stream)
{
    if (!n || stream < -1 || stream >= n->S) return fail(AEV_E_ARG, "bad stream index");
    HIPCHK(hipSetDevice(n->device));
    for (int li = 0; li < (int)n->L.size(); ++li) TRY(reset_layer(n, li, stream < 0 ? 0 : stream, stream < 0 ? n->S : 1));
    HIPCHK(hipStreamSynchronize(n->stream));
    return AEV_OK;
}

static int ensure_buf(int **p, size_t *cap, size_t need)
{
    if (*cap >= need) return AEV_OK;
    if (*p) hipFree(*p);
    *p = nullptr; *cap = 0;
    size_t c = std::max<size_t>(need, 1024);
    HIPCHK(hipMalloc((void **)p, c * sizeof(int)));
    *cap = c;
    return AEV_OK;
}

static int ensure_scratch(aev_net *n, size_t floats)
{
    if (n->scratch_cap >= floats) return AEV_OK;
    if (n->d_scratch) hipFree(n->d_scratch);
    n->d_scratch = nullptr; n->scratch_cap = 0;
    HIPCHK(hipMalloc((void **)&n->d_scratch, floats * sizeof(float)));
    n->scratch_cap = floats;
    return AEV_OK;
}

int aev_layer_step(aev_net *n, int layer, int stream, const int32_t *ev_a, const int32_t *ev_b, int cnt, double *delta_leak)
{
    if (!n || layer < 0 || layer >= (int)n->L.size() || stream < 0 || stream >= n->S || !delta_leak || cnt < 0)
        return fail(AEV_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(n->device));
    Layer &l = n->L[layer];
    hipStream_t st = n->stream;
    if (l.kind == AEV_KIND_INTEG) {
        if (cnt == 0) return fail(AEV_E_ARG, "the integration layer needs at least one event (np.max of an empty array)");
        for (int i = 0; i < cnt; ++i)
            if (ev_a[3 * i] < 0 || ev_a[3 * i] >= l.H || ev_a[3 * i + 1] < 0 || ev_a[3 * i + 1] >= l.W)
                return fail(AEV_E_BOUNDS, "event coordinate outside the surface");
        TRY(ensure_buf(&n->d_yxt, &n->yxt_cap, (size_t)cnt * 3));
        std::vector<long long> off(n->S + 1, 0);
        for (int s = stream + 1; s <= n->S; ++s) off[s] = cnt;
        HIPCHK(hipMemcpyAsync(n->d_yxt, ev_a, (size_t)cnt * 3 * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(n->d_off, off.data(), (n->S + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
        TRY(launch_integ(n, n->d_yxt, n->d_off, nullptr, stream, 1));
        HIPCHK(hipMemcpyAsync(delta_leak, n->d_dl64 + stream, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        return AEV_OK;
    }
    const Layer &p = n->L[layer - 1];
    std::vector<int> packed(std::max(cnt, 1));
    for (int i = 0; i < cnt; ++i) {
        if (ev_a[i] < 0 || ev_a[i] >= p.H || ev_b[i] < 0 || ev_b[i] >= p.W) return fail(AEV_E_BOUNDS, "event coordinate outside the previous layer's map");
        packed[i] = pack_yx(ev_a[i], ev_b[i]);
    }
    TRY(ensure_buf(&n->d_inj, &n->inj_cap, (size_t)cnt));
    HIPCHK(hipMemcpyAsync(n->d_inj, packed.data(), (size_t)cnt * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(n->d_ninj + stream, &cnt, sizeof(int), hipMemcpyHostToDevice, st));
    pool_copy_touch(n, layer); // the per-layer API does not maintain the pool layers' copies
    if (l.kind == AEV_KIND_CONV) {
        const float dlf = (float)*delta_leak; // conv2d.py:115: the float64 scalar meets a float32 array
        const int one = 1;
        HIPCHK(hipMemcpyAsync(n->d_dlf + stream, &dlf, sizeof(float), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(n->d_active + stream, &one, sizeof(int), hipMemcpyHostToDevice, st));
        // ... nor the "may decay" flags: every site is flagged for the sweep below (stale flags would hide decay written by an
        // earlier per-layer step) and again AFTER the recompute -- the sweep un-flags the all-zero sites, including the ones the
        // recompute is about to give a decay, and k_compact (unlike the chain's k_post) does not re-flag them.  The next sweep
        // un-flags lazily whatever really is zero.
        if (l.dflag) HIPCHK(hipMemsetAsync(l.dflag + (size_t)stream * l.mstride, 1, l.mstride, st));
        TRY(launch_leak_sweep(n, layer, layer, stream, 1));
        TRY(launch_conv(n, layer, n->d_inj, n->d_ninj, 0, cnt, stream, 1));
        if (l.dflag) HIPCHK(hipMemsetAsync(l.dflag + (size_t)stream * l.mstride, 1, l.mstride, st));
    } else {
        TRY(launch_pool(n, layer, n->d_inj, n->d_ninj, 0, cnt, stream, 1, nullptr));
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

int aev_layer_num_events(aev_net *n, int layer, int stream, int *cnt)
{
    if (!n || layer < 0 || layer >= (int)n->L.size() || stream < 0 || stream >= n->S || !cnt) return fail(AEV_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(n->device));
    HIPCHK(hipStreamSynchronize(n->stream));
    HIPCHK(hipMemcpy(cnt, n->L[layer].nev + stream, sizeof(int), hipMemcpyDeviceToHost));
    return AEV_OK;
}

int aev_layer_events(aev_net *n, int layer, int stream, int32_t *y, int32_t *x, int capacity, int *cnt)
{
    int c = 0;
    TRY(aev_layer_num_events(n, layer, stream, &c));
    if (c > capacity) return fail(AEV_E_ARG, "event buffers too small");
    std::vector<int> pk(std::max(c, 1));
    const Layer &l = n->L[layer];
    HIPCHK(hipMemcpy(pk.data(), l.ev + (size_t)stream * l.lstride, (size_t)c * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < c; ++i) { y[i] = unpack_y(pk[i]); x[i] = unpack_x(pk[i]); }
    *cnt = c;
    return AEV_OK;
}

static int run_getter(aev_net *n, int layer, int which, int hwc, int s0, int ns, float *d_out)
{
    GetArgs g;
    if (pool_behind_pool(n, layer)) TRY(pool_stack_clean(n, layer)); // (read through its copy, see make_view)
    TRY(make_view(n, layer, &g.self));
    g.which = which; g.hwc = hwc; g.s0 = s0; g.out = d_out;
    n->cur_layer = layer;
    const int gx = std::min(cdiv(n->L[layer].CHW, 256), 1024);
    LAUNCH(n, KC_MISC, k_getter, dim3(gx, ns), dim3(256), 0, g);
    return AEV_OK;
}

int aev_layer_read(aev_net *n, int layer, int stream, int which, void *host_out)
{
    if (!n || layer < 0 || layer >= (int)n->L.size() || stream < 0 || stream >= n->S || !host_out) return fail(AEV_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(n->device));
    const Layer &l = n->L[layer];
    HIPCHK(hipStreamSynchronize(n->stream));
    if (which >= AEV_SURFACE && which <= AEV_FEATUREMAP) {
        TRY(ensure_scratch(n, l.CHW));
        TRY(run_getter(n, layer, which, 0, stream, 1, n->d_scratch));
        HIPCHK(hipMemcpyAsync(host_out, n->d_scratch, (size_t)l.CHW * sizeof(float), hipMemcpyDeviceToHost, n->stream));
        HIPCHK(hipStreamSynchronize(n->stream));
        return AEV_OK;
    }
    if (which == AEV_RAW_ACT && l.kind == AEV_KIND_CONV) {
        std::vector<float> hwc(l.CHW);
        HIPCHK(hipMemcpy(hwc.data(), l.act + (size_t)stream * l.sstride, (size_t)l.CHW * sizeof(float), hipMemcpyDeviceToHost));
        float *o = (float *)host_out; // device keeps [H][W][C]; the caller sees [C,H,W]
        for (int p = 0; p < l.HW; ++p) for (int c = 0; c < l.C; ++c) o[(size_t)c * l.HW + p] = hwc[(size_t)p * l.C + c];
        return AEV_OK;
    }
    if (which == AEV_IDX_MAX && l.kind == AEV_KIND_POOL) {
        std::vector<uint8_t> b(l.CHW);
        HIPCHK(hipMemcpy(b.data(), l.idx + (size_t)stream * l.sstride, l.CHW, hipMemcpyDeviceToHost));
        int32_t *o = (int32_t *)host_out; // [H][W][C] bytes -> [C*H*W] int32 (maxpool.py:33-35)
        for (int p = 0; p < l.HW; ++p) for (int c = 0; c < l.C; ++c) o[(size_t)c * l.HW + p] = b[(size_t)p * l.C + c];
        return AEV_OK;
    }
    if (which == AEV_RECOMPUTE && l.kind == AEV_KIND_POOL) {
        HIPCHK(hipMemcpy(host_out, l.pending + (size_t)stream * l.mstride, l.HW, hipMemcpyDeviceToHost));
        return AEV_OK;
    }
    return fail(AEV_E_ARG, "this map does not exist for this layer kind");
}

static int check_rows(const aev_net *n, const int32_t *yxt, long long rows)
{
    for (long long i = 0; i < rows; ++i)
        if (yxt[3 * i] < 0 || yxt[3 * i] >= n->H || yxt[3 * i + 1] < 0 || yxt[3 * i + 1] >= n->W)
            return fail(AEV_E_BOUNDS, "event coordinate outside the surface");
    return AEV_OK;
}

int aev_net_step(aev_net *n, const int32_t *yxt, const int32_t *offsets)
{
    if (!n || !yxt || !offsets) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    if (offsets[0] != 0) return fail(AEV_E_ARG, "offsets[0] must be 0");
    for (int s = 1; s <= n->S; ++s)
        if (offsets[s] < offsets[s - 1]) return fail(AEV_E_ARG, "offsets must be non-decreasing");
    const long long rows = offsets[n->S];
    TRY(check_rows(n, yxt, rows));
    int most = 0;
    for (int s = 0; s < n->S; ++s) most = std::max(most, offsets[s + 1] - offsets[s]);
    n->integ_fused = most <= AEV_INTEG_FUSED_MAX;
    aev_net::Slot &sl = n->slot[n->cur_slot];
    n->cur_slot ^= 1;
    if (!sl.done) {
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(hipHostMalloc((void **)&sl.h_off, (n->S + 1) * sizeof(long long)));
        HIPCHK(hipMalloc((void **)&sl.d_off, (n->S + 1) * sizeof(long long)));
    } else
        HIPCHK(hipEventSynchronize(sl.done)); // the step that used this slot two calls ago has consumed it
    if (sl.cap < (size_t)rows * 3) {
        const size_t c = std::max<size_t>((size_t)rows * 3 * 2, 4096);
        if (sl.h_yxt) hipHostFree(sl.h_yxt);
        if (sl.d_yxt) hipFree(sl.d_yxt);
        sl.h_yxt = nullptr; sl.d_yxt = nullptr; sl.cap = 0;
        HIPCHK(hipHostMalloc((void **)&sl.h_yxt, c * sizeof(int)));
        HIPCHK(hipMalloc((void **)&sl.d_yxt, c * sizeof(int)));
        sl.cap = c;
    }
    memcpy(sl.h_yxt, yxt, (size_t)rows * 3 * sizeof(int));
    for (int s = 0; s <= n->S; ++s) sl.h_off[s] = offsets[s];
    HIPCHK(hipMemcpyAsync(sl.d_yxt, sl.h_yxt, (size_t)rows * 3 * sizeof(int), hipMemcpyHostToDevice, n->stream));
    HIPCHK(hipMemcpyAsync(sl.d_off, sl.h_off, (n->S + 1) * sizeof(long long), hipMemcpyHostToDevice, n->stream));
    TRY(ensure_stamps(n));
    TRY(ensure_chain_events(n));
    TRY(pool_copies_clean(n));
    TRY(launch_chain(n, sl.d_yxt, sl.d_off, nullptr, 0, n->S));
    HIPCHK(hipEventRecord(sl.done, n->stream));
    n->host_steps += 1;
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

int aev_net_queue_upload(aev_net *n, const int32_t *yxt, const int64_t *offsets, int T)
{
    if (!n || !yxt || !offsets || T <= 0) return fail(AEV_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(n->device));
    const long long cells = (long long)T * n->S;
    if (offsets[0] != 0) return fail(AEV_E_ARG, "offsets[0] must be 0");
    for (long long i = 1; i <= cells; ++i) if (offsets[i] < offsets[i - 1]) return fail(AEV_E_ARG, "offsets must be non-decreasing");
    const long long rows = offsets[cells];
    TRY(check_rows(n, yxt, rows));
    long long most = 0;
    for (long long i = 0; i < cells; ++i) most = std::max<long long>(most, offsets[i + 1] - offsets[i]);
    n->queue_fused_ok = most <= AEV_INTEG_FUSED_MAX;
    HIPCHK(hipStreamSynchronize(n->stream));
    if (n->d_qyxt) hipFree(n->d_qyxt);
    if (n->d_qoff) hipFree(n->d_qoff);
    n->d_qyxt = nullptr; n->d_qoff = nullptr; n->qT = 0;
    HIPCHK(hipMalloc((void **)&n->d_qyxt, std::max<long long>(rows, 1) * 3 * sizeof(int)));
    HIPCHK(hipMalloc((void **)&n->d_qoff, (cells + 1) * sizeof(long long)));
    HIPCHK(hipMemcpy(n->d_qyxt, yxt, (size_t)rows * 3 * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(n->d_qoff, offsets, (cells + 1) * sizeof(long long), hipMemcpyHostToDevice));
    n->qT = T;
    drop_graph(n); // the captured graph holds the old queue pointers
    return AEV_OK;
}

// The feeder on the device: S whole samples -> cropped, shifted, chunked, step-major event queue.
int aev_net_queue_upload_samples(aev_net *n, const int32_t *x, const int32_t *y, const int32_t *ts, const int64_t *sample_off, int example_h,
                                 int example_w, int batch_event_size, int batch_event_usec, int *T_out)
{
    if (!n || !x || !y || !ts || !sample_off || !T_out) return fail(AEV_E_ARG, "null argument");
    if ((batch_event_usec <= 0 && batch_event_size <= 0)) return fail(AEV_E_ARG, "batch_event_size or batch_event_usec must be positive");
    HIPCHK(hipSetDevice(n->device));
    const int S = n->S;
    if (sample_off[0] != 0) return fail(AEV_E_ARG, "sample_off[0] must be 0");
    for (int s = 1; s <= S; ++s) if (sample_off[s] < sample_off[s - 1]) return fail(AEV_E_ARG, "sample offsets must be non-decreasing");
    const long long rows = sample_off[S];
    if (rows <= 0) return fail(AEV_E_ARG, "no events");
    long long tcap = 1;
    for (int s = 0; s < S; ++s) {
        const long long a = sample_off[s], m = sample_off[s + 1] - a;
        if (m == 0) continue;
        for (long long i = a + 1; i < a + m; ++i) if (ts[i] < ts[i - 1]) return fail(AEV_E_ARG, "a sample's timestamps must be non-decreasing");
        tcap = std::max(tcap, batch_event_usec > 0 ? ((long long)ts[a + m - 1] - ts[a]) / batch_event_usec + 2 : (m + batch_event_size - 1) / batch_event_size);
    }
    if (tcap * S > (1ll << 28)) return fail(AEV_E_SHAPE, "too many steps");
    int *d_raw = nullptr, *d_c = nullptr, *d_meta = nullptr; long long *d_soff = nullptr;
    HIPCHK(hipStreamSynchronize(n->stream));
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void **)&d_raw, (size_t)rows * 3 * sizeof(int))); HIPCHK(hipMalloc((void **)&d_c, (size_t)rows * 4 * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_soff, (S + 1) * sizeof(long long)));
        const size_t meta = (size_t)2 * S + 1 + (size_t)S * (tcap + 1);
        HIPCHK(hipMalloc((void **)&d_meta, meta * sizeof(int)));
        HIPCHK(hipMemcpy(d_raw, x, (size_t)rows * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_raw + rows, y, (size_t)rows * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_raw + 2 * rows, ts, (size_t)rows * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_soff, sample_off, (S + 1) * sizeof(long long), hipMemcpyHostToDevice));
        HIPCHK(hipMemsetAsync(d_meta, 0, meta * sizeof(int), n->stream));
        FeedArgs a{};
        a.x = d_raw; a.y = d_raw + rows; a.ts = d_raw + 2 * rows; a.soff = d_soff;
        a.crop = (example_h != n->H || example_w != n->W) ? 1 : 0; a.new_h = n->H; a.new_w = n->W;             // runner.py:26-27
        a.size = batch_event_usec > 0 ? 0 : batch_event_size; a.usec = batch_event_usec > 0 ? batch_event_usec : 0;
        a.H = n->H; a.W = n->W; a.Tcap = (int)tcap;
        a.cx = d_c; a.cy = d_c + rows; a.cts = d_c + 2 * rows; a.chunk = d_c + 3 * rows;
        a.n_in = d_meta; a.nchunks = d_meta + S; a.bad = d_meta + 2 * S; a.cstart = d_meta + 2 * S + 1;
        hipLaunchKernelGGL(k_feed_prepare, dim3(S), dim3(1024), 0, n->stream, a);
        std::vector<int> h(2 * S + 1);
        HIPCHK(hipMemcpyAsync(h.data(), d_meta, h.size() * sizeof(int), hipMemcpyDeviceToHost, n->stream));
        HIPCHK(hipStreamSynchronize(n->stream));
        if (h[2 * S]) return fail(AEV_E_BOUNDS, "event coordinate outside the surface");
        long long kept = 0; int T = 0;
        for (int s = 0; s < S; ++s) { kept += h[s]; T = std::max(T, h[S + s]); }
        if (T == 0) return fail(AEV_E_ARG, "no event is left after the crop");
        if (n->d_qyxt) hipFree(n->d_qyxt);
        if (n->d_qoff) hipFree(n->d_qoff);
        n->d_qyxt = nullptr; n->d_qoff = nullptr; n->qT = 0;
        HIPCHK(hipMalloc((void **)&n->d_qyxt, std::max<long long>(kept, 1) * 3 * sizeof(int)));
        HIPCHK(hipMalloc((void **)&n->d_qoff, ((long long)T * S + 1) * sizeof(long long)));
        hipLaunchKernelGGL(k_feed_offsets, dim3(1), dim3(1024), 0, n->stream, S, T, (int)tcap, a.cstart, a.nchunks, n->d_qoff);
        long long most = 0;
        for (int s = 0; s < S; ++s) most = std::max<long long>(most, h[s]);
        hipLaunchKernelGGL(k_feed_scatter, dim3(std::max(1, std::min(cdiv(most, 256), 1024)), S), dim3(256), 0, n->stream, a, S, n->d_qoff, n->d_qyxt);
        HIPCHK(hipStreamSynchronize(n->stream));
        HIPCHK(hipGetLastError());
        n->qT = T;
        *T_out = T;
        // (np.array_split: chunks of ceil(kept / chunks) events at most; time bins: unknown without a look at the queue)
        long long most_chunk = 0;
        for (int s = 0; s < S; ++s) if (h[S + s] > 0) most_chunk = std::max<long long>(most_chunk, (h[s] + h[S + s] - 1) / h[S + s]);
        n->queue_fused_ok = batch_event_usec <= 0 && most_chunk <= AEV_INTEG_FUSED_MAX;
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_raw); hipFree(d_c); hipFree(d_meta); hipFree(d_soff);
    drop_graph(n); // the captured graphs hold the old queue pointers
    return rc;
}

int aev_net_queue_download(aev_net *n, int32_t *yxt, long long row_capacity, int64_t *offsets, int *T_out)
{
    if (!n || !T_out) return fail(AEV_E_ARG, "null argument");
    if (!n->d_qyxt) return fail(AEV_E_ARG, "no event queue uploaded");
    HIPCHK(hipSetDevice(n->device));
    HIPCHK(hipStreamSynchronize(n->stream));
    const long long cells = (long long)n->qT * n->S;
    long long rows = 0;
    HIPCHK(hipMemcpy(&rows, n->d_qoff + cells, sizeof(long long), hipMemcpyDeviceToHost));
    *T_out = n->qT;
    if (offsets) HIPCHK(hipMemcpy(offsets, n->d_qoff, (cells + 1) * sizeof(long long), hipMemcpyDeviceToHost));
    if (yxt) {
        if (rows > row_capacity) return fail(AEV_E_ARG, "event buffer too small");
        HIPCHK(hipMemcpy(yxt, n->d_qyxt, (size_t)rows * 3 * sizeof(int), hipMemcpyDeviceToHost));
    }
    return AEV_OK;
}

// (8 steps per graph: no 5-9 us replay gap between them, and the tail of every step but the graph's last runs beside the head
//  of the next, see launch_chain; 4 -> 8: -0.3 % on the e-fcn at 256 streams, 12: the same again)

// validate, point the device step counter at t0, capture the step graph on first use
static int queue_prepare(aev_net *n, int t0, int T, int use_graph)
{
    if (!n || !n->d_qyxt) return fail(AEV_E_ARG, "no event queue uploaded");
    if (t0 < 0 || T < 0 || t0 + T > n->qT) return fail(AEV_E_ARG, "queue range out of bounds");
    HIPCHK(hipSetDevice(n->device));
    // (kernel arguments of fill kernels instead of copies from host memory: nothing to keep alive, no reason to wait for the stream below)
    hipLaunchKernelGGL(k_queue_begin, dim3(1), dim3(1), 0, n->stream, n->d_qstep, n->d_qrun, t0, t0 + T);
    TRY(ensure_stamps(n));
    TRY(ensure_chain_events(n));
    TRY(pool_copies_clean(n)); // (before any capture)
    n->integ_fused = n->queue_fused_ok;
    if (use_graph && !n->profiling && n->graphs.empty()) HIPCHK(hipStreamSynchronize(n->stream)); // (before the captures below; a run that replays what is captured starts without a host wait)
    if (n->sw.verbose && n->graphs.empty())
        fprintf(stderr, "[aev] queued steps: second stream %d, look-ahead sweeps %s (layers 1..%d, own state of layer %d), tail beside the next step's head %s (layers %d.., behind layer %d), %d steps per graph\n",
                n->overlap, n->lookahead && n->la_gate > 0 ? "on" : "off", n->la_gate, n->conv_impl == 0 ? n->la_state : 0,
                n->lookahead && n->la_gate > 0 && n->overlap && n->tail_overlap && n->tail_from > 0 && n->conv_impl == 0 ? "on" : "off", n->tail_from, n->tail_feed, n->sw.graph_steps);
    // One graph per step, and one of AEV_GRAPH_STEPS steps for long runs: a step reads its events through the device step
    // counter, so steps chain inside a graph as they do across replays, without the 5-9 us between two graph launches.
    auto capture = [&](int steps, bool incoming, bool defers, hipGraph_t *src, hipGraphExec_t *exec) -> int {
        HIPCHK(hipStreamBeginCapture(n->stream, hipStreamCaptureModeThreadLocal));
        int rc = AEV_OK;
        // (the tail that comes in belongs to a step of the graph replayed before this one: everything it depends on is complete,
        //  there is no event of this capture to wait for)
        n->tail_pending = incoming; n->tail_sweeps4 = incoming;
        for (int k = 0; k < steps && rc >= 0; ++k) {
            rc = launch_chain(n, n->d_qyxt, n->d_qoff, n->d_qstep, 0, n->S, k + 1 < steps || defers);
            if (!chain_advances(n)) hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, n->stream, n->d_qstep, n->d_stats + 63);
        }
        const bool deferred = n->tail_pending;
        n->tail_pending = n->tail_sweeps4 = false;
        const hipError_t e = hipStreamEndCapture(n->stream, src);
        if (rc < 0) return rc;
        if (deferred != defers) { if (e == hipSuccess) hipGraphDestroy(*src); return AEV_E_UNSUPPORTED; } // (this net's steps do not overlap: no such variant)
        if (e != hipSuccess) return fail(AEV_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        HIPCHK(hipGraphInstantiate(exec, *src, nullptr, nullptr, 0));
        return AEV_OK;
    };
    // (all of them captured with the first graph run -- every length the queue could ever replay -- not by the first run that
    //  needs one: a warm-up run of a few steps then leaves nothing to capture inside a timed run)
    if (use_graph && !n->profiling && n->graphs.empty()) {
        bool overlaps = true; // (found out by the first variant that tries to leave a tail behind)
        for (int steps = n->sw.graph_steps; steps >= 1; steps /= 2) {
            if (steps > 1 && n->qT < steps) continue;
            for (int v = 0; v < 4; ++v) { // plain, deferring, incoming, both
                const bool defers = (v & 1) != 0, incoming = (v & 2) != 0;
                if ((incoming || defers) && (!overlaps || n->qT < steps + 1)) continue;
                aev_net::StepGraph g{steps, incoming, defers, nullptr, nullptr};
                const int rc = capture(steps, incoming, defers, &g.src, &g.exec);
                if (rc == AEV_E_UNSUPPORTED && (incoming || defers)) { overlaps = false; continue; }
                TRY(rc);
                n->graphs.push_back(g);
            }
        }
    }
    return AEV_OK;
}

static int queue_launch(aev_net *n, int T, int use_graph)
{
    if (T > 0 && n->lookahead && n->la_gate > 0 && !n->dense && !first_step_sweeps_itself(n)) {
        // Prologue of a run with the look-ahead (the condition is launch_chain's `la`) whose first step does not sweep for itself: the first
        // step's sweeps of the look-ahead layers, from its delta_leak as k_integ_peek computes it -- what the previous step does for every
        // other step of the run.
        IntegArgs a{};
        fill_integ_args(n, a, n->d_qyxt, n->d_qoff, n->d_qstep, 0, true);
        n->cur_layer = 0;
        LAUNCH(n, KC_INTEG_PREP, k_integ_peek, dim3(n->S), dim3(256), 0, a);
        // (one launch: layers 1 .. la_gate with the pool copies behind them, and the own state of the layer behind the gate)
        if (n->la_state > n->la_gate && n->conv_impl == 0) TRY(launch_leak_sweep(n, 1, n->la_state, 0, n->S, nullptr, true, 3, 1, 3, 1));
        else TRY(launch_leak_sweep(n, 1, n->la_gate, 0, n->S, nullptr, true, 3));
    }
    if (use_graph && !n->profiling) {
        // the longest graphs that still fit, each in the variant its place in the run asks for (plain ones if the net has no other)
        bool overlaps = false;
        for (auto &g : n->graphs) overlaps |= g.defers;
        auto find = [&](int steps, bool incoming, bool defers) -> const aev_net::StepGraph * {
            for (auto &g : n->graphs) if (g.steps == steps && g.incoming == incoming && g.defers == defers) return &g;
            return nullptr;
        };
        int t = 0;
        bool pending = false;
        while (t < T) {
            int steps = 0;
            for (auto &g : n->graphs) if (!g.incoming && !g.defers && g.steps <= T - t) { steps = g.steps; break; } // (longest first)
            if (!steps) return fail(AEV_E_HIP, "internal: no step graph");
            const bool defers = overlaps && t + steps < T;
            const aev_net::StepGraph *g = find(steps, pending, defers);
            if (!g) return fail(AEV_E_HIP, "internal: step graph variant missing");
            HIPCHK(hipGraphLaunch(g->exec, n->stream));
            pending = defers;
            t += steps;
        }
    } else {
        for (int t = 0; t < T; ++t) {
            TRY(launch_chain(n, n->d_qyxt, n->d_qoff, n->d_qstep, 0, n->S, t + 1 < T));
            if (!chain_advances(n)) LAUNCH(n, KC_MISC, k_advance, dim3(1), dim3(1), 0, n->d_qstep, n->d_stats + 63);
        }
    }
    n->host_steps += T;
    return AEV_OK;
}

static int queue_run(aev_net *n, int t0, int T, int use_graph)
{
    if (stamps_wanted(n)) use_graph = 0; // the stamps of a replayed graph would not be cleared per launch
    TRY(queue_prepare(n, t0, T, use_graph));
    TRY(queue_launch(n, T, use_graph));
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

int aev_net_queue_run(aev_net *n, int t0, int T, int use_graph) { return queue_run(n, t0, T, use_graph); }

int aev_net_queue_run_timed(aev_net *n, int t0, int T, int use_graph, float *ms)
{
    if (!n || !ms) return fail(AEV_E_ARG, "null argument");
    if (stamps_wanted(n)) use_graph = 0;
    TRY(queue_prepare(n, t0, T, use_graph));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, n->stream));
    const int rc = queue_launch(n, T, use_graph);
    HIPCHK(hipEventRecord(b, n->stream));
    HIPCHK(hipEventSynchronize(b));
    HIPCHK(hipEventElapsedTime(ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    HIPCHK(hipGetLastError());
    return rc;
}

int aev_net_queue_profile(aev_net *n, int t0, int T, float *ms_by_kernel, int64_t *launches_by_kernel)
{
    if (!n || !ms_by_kernel || !launches_by_kernel) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    const int L = (int)n->L.size();
    n->profiling = true;
    n->prof.clear();
    int rc = queue_run(n, t0, T, 0);
    hipStreamSynchronize(n->stream);
    n->profiling = false;
    for (int k = 0; k < L * AEV_NUM_KERNELS; ++k) { ms_by_kernel[k] = 0.f; launches_by_kernel[k] = 0; }
    for (ProfRec &r : n->prof) {
        float ms = 0.f;
        const int slot = std::min(std::max(r.layer, 0), L - 1) * AEV_NUM_KERNELS + r.cls;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { ms_by_kernel[slot] += ms; launches_by_kernel[slot] += 1; }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    n->prof.clear();
    return rc;
}

int aev_net_sync(aev_net *n)
{
    if (!n) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    HIPCHK(hipStreamSynchronize(n->stream));
    if (n->d_stamps) dump_stamps(n);
    if (n->d_pstamps) { // AEV_POST_STAMPS: k_post2's phases, block 0, mean microseconds per launch of each layer's bookkeeping
        std::vector<unsigned long long> h(64 * 16);
        hipMemcpy(h.data(), n->d_pstamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        static const char *ph[12] = {"setup", "maps (own share)", "maps (block)", "event list", "window rows", "window list", "dilation", "touch bytes", "tile masks", "tile list", "site list", "pool in place"};
        for (int li = 0; li < (int)n->L.size() && li < 64; ++li) {
            const unsigned long long cnt = h[li * 16 + 15];
            if (!cnt) continue;
            double tot = 0;
            std::string line;
            for (int q = 0; q < 12; ++q) { const double us = h[li * 16 + q] / 100.0 / cnt; tot += us; if (us > 0.005) { char b[64]; snprintf(b, sizeof(b), " %s %.2f", ph[q], us); line += b; } }
            fprintf(stderr, "[AEV_POST_STAMPS] layer %d: %llu launches, %.2f us per launch (block 0):%s\n", li, cnt, tot, line.c_str());
        }
        hipMemsetAsync(n->d_pstamps, 0, h.size() * sizeof(unsigned long long), n->stream);
    }
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

int aev_net_output_device(aev_net *n, void *device_out)
{
    if (!n || !device_out) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    TRY(run_getter(n, (int)n->L.size() - 1, AEV_FEATUREMAP, 1, 0, n->S, (float *)device_out));
    HIPCHK(hipStreamSynchronize(n->stream));
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

int aev_net_output(aev_net *n, float *host_out)
{
    if (!n || !host_out) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    const size_t cnt = (size_t)n->S * n->L.back().CHW;
    TRY(ensure_scratch(n, cnt));
    TRY(aev_net_output_device(n, n->d_scratch));
    HIPCHK(hipMemcpy(host_out, n->d_scratch, cnt * sizeof(float), hipMemcpyDeviceToHost));
    return AEV_OK;
}

int aev_net_stats(aev_net *n, int64_t *sites, int64_t *events_in, int64_t *steps)
{
    if (!n) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    HIPCHK(hipStreamSynchronize(n->stream));
    u64 h[64];
    HIPCHK(hipMemcpy(h, n->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    if (sites) for (size_t i = 0; i < n->L.size() && i < 62; ++i) sites[i] = (int64_t)h[i];
    if (events_in) *events_in = (int64_t)h[62];
    if (steps) *steps = n->host_steps;
    return AEV_OK;
}

int aev_net_stats_reset(aev_net *n)
{
    if (!n) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    HIPCHK(hipStreamSynchronize(n->stream));
    HIPCHK(hipMemsetAsync(n->d_stats, 0, 64 * sizeof(u64), n->stream));
    n->host_steps = 0;
    return AEV_OK;
}

int aev_net_set_tuning(aev_net *n, int tuning)
{
    if (!n || tuning < 0 || tuning > 2) return fail(AEV_E_ARG, "tuning must be 0 (auto), 1 (throughput) or 2 (few streams)");
    if (n->L.size() > 1) return fail(AEV_E_ARG, "set the tuning before adding layers");
    n->tuning = tuning;
    return AEV_OK;
}

int aev_net_set_conv_math(aev_net *n, int math)
{
    if (!n || (math != AEV_MATH_F32 && math != AEV_MATH_BF16X6)) return fail(AEV_E_ARG, "math must be AEV_MATH_F32 (0) or AEV_MATH_BF16X6 (1)");
    if (n->L.size() > 1) return fail(AEV_E_ARG, "set the conv arithmetic before adding layers");
    n->conv_math = math;
    return AEV_OK;
}

int aev_net_set_dense(aev_net *n, int dense)
{
    if (!n || dense < 0 || dense > 2) return fail(AEV_E_ARG, "dense must be 0 (asynchronous), 1 (dense frame) or 2 (dense frame, activation again after every pool)");
    if (dense && n->generic) return fail(AEV_E_UNSUPPORTED, "dense-frame mode is not available for a chain with a pool layer directly behind a pool layer");
    n->dense = dense;
    drop_graph(n);
    return AEV_OK;
}

int aev_net_set_conv_impl(aev_net *n, int impl)
{
    if (!n || impl < 0 || impl > 2) return fail(AEV_E_ARG, "impl must be 0, 1 or 2");
    n->conv_impl = impl;
    drop_graph(n);
    return AEV_OK;
}

// ---- stand-alone ops ---------------------------------------------------------------------------
int aev_im2col_event(const float *image, int C, int H, int W, const int32_t *ey, const int32_t *ex, int N, int kh,
                     int kw, int stride, int chan_as_cols, float *out_cols, int32_t *out_y, int32_t *out_x,
                     int *n_out, int device)
{
    if (!image || !out_cols || !out_y || !out_x || !n_out || N < 0 || (N > 0 && (!ey || !ex))) return fail(AEV_E_ARG, "null argument");
    if (!(stride == 1 || (stride == kh && stride == kw)))
        return fail(AEV_E_STRIDE, "This method only support stride equal to 1 or to the kernel's dimensions.");
    if (C <= 0 || kh <= 0 || kw <= 0 || H < kh || W < kw || H > 32767 || W > 32767) return fail(AEV_E_SHAPE, "bad image / kernel shape");
    if (stride != 1 && (H % kh || W % kw)) return fail(AEV_E_SHAPE, "pooled dimensions must be divisible by the window (SURVEY Q7)");
    for (int i = 0; i < N; ++i)
        if (ey[i] < 0 || ey[i] >= H || ex[i] < 0 || ex[i] >= W) return fail(AEV_E_BOUNDS, "event coordinate outside the image");
    TRY(select_device(device));
    const int oh = (H - kh) / stride + 1, ow = (W - kw) / stride + 1, kk = kh * kw, rows = C * kk;
    std::vector<int> packed(std::max(N, 1));
    for (int i = 0; i < N; ++i) packed[i] = pack_yx(ey[i], ex[i]);
    float *d_img = nullptr, *d_cols = nullptr;
    int *d_ev = nullptr, *d_n = nullptr, *d_key = nullptr, *d_list = nullptr, *d_nl = nullptr;
    int rc = AEV_OK;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void **)&d_img, (size_t)C * H * W * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_ev, std::max(N, 1) * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_n, sizeof(int))); HIPCHK(hipMalloc((void **)&d_nl, sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_key, (size_t)oh * ow * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_list, (size_t)oh * ow * sizeof(int)));
        HIPCHK(hipMemcpy(d_img, image, (size_t)C * H * W * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ev, packed.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_n, &N, sizeof(int), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fill_int, dim3(64), dim3(256), 0, 0, d_key, (long long)oh * ow, INT_MAX);
        MarkArgs m{};
        m.g.mode = stride == 1 ? 0 : 1; m.g.kh = kh; m.g.kw = kw; m.g.oh = oh; m.g.ow = ow; m.g.ih = H; m.g.iw = W;
        m.in_ev = d_ev; m.in_n = d_n; m.in_stride = 0; m.key = d_key; m.pending = nullptr; m.mstride = 0; m.s0 = 0;
        const int slots = stride == 1 ? kk : 1;
        hipLaunchKernelGGL(k_touch_mark, dim3(std::max(1, std::min(cdiv((long long)N * slots, 256), 1024)), 1), dim3(256), 0, 0, m);
        ListArgs q{};
        q.g = m.g; q.in_ev = d_ev; q.in_n = d_n; q.in_stride = 0; q.key = d_key; q.pending = nullptr; q.mstride = 0; q.s0 = 0;
        q.list = d_list; q.nlist = d_nl; q.lstride = 0; q.stat = nullptr; q.active = nullptr;
        hipLaunchKernelGGL(k_touch_list, dim3(1), dim3(1024), 0, 0, q);
        int nl = 0;
        HIPCHK(hipMemcpy(&nl, d_nl, sizeof(int), hipMemcpyDeviceToHost));
        if (nl > 0) {
            HIPCHK(hipMalloc((void **)&d_cols, (size_t)nl * rows * sizeof(float)));
            hipLaunchKernelGGL(k_op_gather, dim3(std::min(cdiv((long long)nl * rows, 256), 4096)), dim3(256), 0, 0, d_img, C, H, W,
                               kh, kw, stride, chan_as_cols, d_list, d_nl, d_cols);
            HIPCHK(hipMemcpy(out_cols, d_cols, (size_t)nl * rows * sizeof(float), hipMemcpyDeviceToHost));
            std::vector<int> pk(nl);
            HIPCHK(hipMemcpy(pk.data(), d_list, (size_t)nl * sizeof(int), hipMemcpyDeviceToHost));
            for (int i = 0; i < nl; ++i) { out_y[i] = unpack_y(pk[i]); out_x[i] = unpack_x(pk[i]); }
        }
        HIPCHK(hipGetLastError());
        *n_out = nl;
        return AEV_OK;
    };
    rc = body();
    hipFree(d_img); hipFree(d_cols); hipFree(d_ev); hipFree(d_n); hipFree(d_key); hipFree(d_list); hipFree(d_nl);
    return rc;
}

int aev_min_argmax(const float *mx, const float *mn, int R, int M, int32_t *argmax, int32_t *not_argmin, int device)
{
    if (!mx || !mn || !argmax || !not_argmin) return fail(AEV_E_ARG, "null argument");
    if (R <= 0 || M < 0) return fail(AEV_E_SHAPE, "bad matrix shape");
    if (M == 0) return AEV_OK;
    TRY(select_device(device));
    float *d_mx = nullptr, *d_mn = nullptr; int *d_a = nullptr, *d_b = nullptr;
    auto body = [&]() -> int {
        const size_t bytes = (size_t)R * M * sizeof(float);
        HIPCHK(hipMalloc((void **)&d_mx, bytes)); HIPCHK(hipMalloc((void **)&d_mn, bytes));
        HIPCHK(hipMalloc((void **)&d_a, M * sizeof(int))); HIPCHK(hipMalloc((void **)&d_b, M * sizeof(int)));
        HIPCHK(hipMemcpy(d_mx, mx, bytes, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_mn, mn, bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_op_min_argmax, dim3(std::min(cdiv(M, 256), 2048)), dim3(256), 0, 0, d_mx, d_mn, R, M, d_a, d_b);
        HIPCHK(hipMemcpy(argmax, d_a, M * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(not_argmin, d_b, M * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipGetLastError());
        return AEV_OK;
    };
    int rc = body();
    hipFree(d_mx); hipFree(d_mn); hipFree(d_a); hipFree(d_b);
    return rc;
}

// ---- the conv ops the reference's callers import by name (conv2d.py:144-230, functional.py:4-34) ------------------
// kernel [Cout][C][kh][kw] (the layout conv2d.py:26 keeps) -> Wt[chain position][CoutP]
static void chain_weights(const float *kernel_oihw, int Cout, int C, int kh, int kw, int CoutP, std::vector<float> &wt)
{
    const int kk = kh * kw;
    wt.assign((size_t)C * kk * CoutP, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int c = 0; c < C; ++c)
            for (int r = 0; r < kk; ++r)
                wt[(size_t)chain_pos(c, r, C, kk) * CoutP + co] = kernel_oihw[((size_t)co * C + c) * kk + r];
}

int aev_conv2d_event(const float *image, int C, int H, int W, const int32_t *ey, const int32_t *ex, int N, const float *kernel_oihw,
                     const float *bias, int Cout, int kh, int kw, int stride, float *out_conv, int32_t *out_y, int32_t *out_x,
                     int *n_out, int device)
{
    if (!image || !kernel_oihw || !out_conv || !out_y || !out_x || !n_out || N < 0 || (N > 0 && (!ey || !ex))) return fail(AEV_E_ARG, "null argument");
    if (!(stride == 1 || (stride == kh && stride == kw)))
        return fail(AEV_E_STRIDE, "This method only support stride equal to 1 or to the kernel's dimensions.");
    if (C <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || H < kh || W < kw || H > 32767 || W > 32767) return fail(AEV_E_SHAPE, "bad image / kernel shape");
    if (stride != 1 && (H % kh || W % kw)) return fail(AEV_E_SHAPE, "pooled dimensions must be divisible by the window (SURVEY Q7)");
    for (int i = 0; i < N; ++i)
        if (ey[i] < 0 || ey[i] >= H || ex[i] < 0 || ex[i] >= W) return fail(AEV_E_BOUNDS, "event coordinate outside the image");
    TRY(select_device(device));
    const int oh = (H - kh) / stride + 1, ow = (W - kw) / stride + 1, kk = kh * kw, CoutP = cdiv(Cout, 32) * 32;
    std::vector<int> packed(std::max(N, 1));
    for (int i = 0; i < N; ++i) packed[i] = pack_yx(ey[i], ex[i]);
    std::vector<float> wt;
    chain_weights(kernel_oihw, Cout, C, kh, kw, CoutP, wt);
    float *d_img = nullptr, *d_out = nullptr, *d_wt = nullptr, *d_bias = nullptr;
    int *d_ev = nullptr, *d_n = nullptr, *d_key = nullptr, *d_list = nullptr, *d_nl = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void **)&d_img, (size_t)C * H * W * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_wt, wt.size() * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_ev, std::max(N, 1) * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_n, sizeof(int))); HIPCHK(hipMalloc((void **)&d_nl, sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_key, (size_t)oh * ow * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_list, (size_t)oh * ow * sizeof(int)));
        HIPCHK(hipMemcpy(d_img, image, (size_t)C * H * W * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
        if (bias) { HIPCHK(hipMalloc((void **)&d_bias, Cout * sizeof(float))); HIPCHK(hipMemcpy(d_bias, bias, Cout * sizeof(float), hipMemcpyHostToDevice)); }
        HIPCHK(hipMemcpy(d_ev, packed.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_n, &N, sizeof(int), hipMemcpyHostToDevice));
        // the index half of im2col_event (cutils.pyx:73-112): first-touch ordered list of the affected output sites
        hipLaunchKernelGGL(k_fill_int, dim3(64), dim3(256), 0, 0, d_key, (long long)oh * ow, INT_MAX);
        MarkArgs m{};
        m.g.mode = stride == 1 ? 0 : 1; m.g.kh = kh; m.g.kw = kw; m.g.oh = oh; m.g.ow = ow; m.g.ih = H; m.g.iw = W;
        m.in_ev = d_ev; m.in_n = d_n; m.in_stride = 0; m.key = d_key; m.pending = nullptr; m.mstride = 0; m.s0 = 0;
        const int slots = stride == 1 ? kk : 1;
        hipLaunchKernelGGL(k_touch_mark, dim3(std::max(1, std::min(cdiv((long long)N * slots, 256), 1024)), 1), dim3(256), 0, 0, m);
        ListArgs q{};
        q.g = m.g; q.in_ev = d_ev; q.in_n = d_n; q.in_stride = 0; q.key = d_key; q.pending = nullptr; q.mstride = 0; q.s0 = 0;
        q.list = d_list; q.nlist = d_nl; q.lstride = 0; q.stat = nullptr; q.active = nullptr;
        hipLaunchKernelGGL(k_touch_list, dim3(1), dim3(1024), 0, 0, q);
        int nl = 0;
        HIPCHK(hipMemcpy(&nl, d_nl, sizeof(int), hipMemcpyDeviceToHost));
        if (nl > 0) {
            HIPCHK(hipMalloc((void **)&d_out, (size_t)nl * Cout * sizeof(float)));
            OpConvArgs a{};
            a.img = d_img; a.B = 1; a.C = C; a.H = H; a.W = W; a.pt = 0; a.pl = 0; a.wt = d_wt; a.bias = d_bias;
            a.Cout = Cout; a.CoutP = CoutP; a.kh = kh; a.kw = kw; a.stride = stride; a.oh = oh; a.ow = ow; a.list = d_list; a.n = nl; a.out = d_out;
            hipLaunchKernelGGL(k_op_conv, dim3(std::min(cdiv((long long)nl * Cout, 256), 8192)), dim3(256), 0, 0, a);
            HIPCHK(hipMemcpy(out_conv, d_out, (size_t)nl * Cout * sizeof(float), hipMemcpyDeviceToHost));
            std::vector<int> pk(nl);
            HIPCHK(hipMemcpy(pk.data(), d_list, (size_t)nl * sizeof(int), hipMemcpyDeviceToHost));
            for (int i = 0; i < nl; ++i) { out_y[i] = unpack_y(pk[i]); out_x[i] = unpack_x(pk[i]); }
        }
        HIPCHK(hipGetLastError());
        *n_out = nl;
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_img); hipFree(d_out); hipFree(d_wt); hipFree(d_bias); hipFree(d_ev); hipFree(d_n); hipFree(d_key); hipFree(d_list); hipFree(d_nl);
    return rc;
}

int aev_conv2d(const float *image, int B, int C, int H, int W, const float *kernel_oihw, const float *bias, int Cout, int kh, int kw,
               int pad_top, int pad_bottom, int pad_left, int pad_right, int stride, float *out, int device)
{
    if (!image || !kernel_oihw || !out) return fail(AEV_E_ARG, "null argument");
    if (B <= 0 || C <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad_top < 0 || pad_bottom < 0 || pad_left < 0 || pad_right < 0)
        return fail(AEV_E_SHAPE, "bad image / kernel shape");
    const int Hp = H + pad_top + pad_bottom, Wp = W + pad_left + pad_right;
    if (Hp < kh || Wp < kw) return fail(AEV_E_SHAPE, "kernel larger than the (padded) image");
    TRY(select_device(device));
    const int oh = (Hp - kh) / stride + 1, ow = (Wp - kw) / stride + 1, CoutP = cdiv(Cout, 32) * 32;
    std::vector<float> wt;
    chain_weights(kernel_oihw, Cout, C, kh, kw, CoutP, wt);
    float *d_img = nullptr, *d_out = nullptr, *d_wt = nullptr, *d_bias = nullptr;
    auto body = [&]() -> int {
        const size_t in_sz = (size_t)B * C * H * W, out_sz = (size_t)B * Cout * oh * ow;
        HIPCHK(hipMalloc((void **)&d_img, in_sz * sizeof(float))); HIPCHK(hipMalloc((void **)&d_out, out_sz * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_wt, wt.size() * sizeof(float)));
        HIPCHK(hipMemcpy(d_img, image, in_sz * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
        if (bias) { HIPCHK(hipMalloc((void **)&d_bias, Cout * sizeof(float))); HIPCHK(hipMemcpy(d_bias, bias, Cout * sizeof(float), hipMemcpyHostToDevice)); }
        OpConvArgs a{};
        a.img = d_img; a.B = B; a.C = C; a.H = H; a.W = W; a.pt = pad_top; a.pl = pad_left; a.wt = d_wt; a.bias = d_bias;
        a.Cout = Cout; a.CoutP = CoutP; a.kh = kh; a.kw = kw; a.stride = stride; a.oh = oh; a.ow = ow; a.list = nullptr; a.n = 0; a.out = d_out;
        hipLaunchKernelGGL(k_op_conv, dim3(std::min(cdiv((long long)out_sz, 256), 16384)), dim3(256), 0, 0, a);
        HIPCHK(hipMemcpy(out, d_out, out_sz * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(hipGetLastError());
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_img); hipFree(d_out); hipFree(d_wt); hipFree(d_bias);
    return rc;
}

int aev_im2col(const void *image, int elem_bytes, int B, int C, int H, int W, int kh, int kw, int stride, void *out_cols, int device)
{
    if (!image || !out_cols) return fail(AEV_E_ARG, "null argument");
    if (elem_bytes != 4 && elem_bytes != 8) return fail(AEV_E_ARG, "elements must be 4 or 8 bytes wide");
    if (B <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || H < kh || W < kw) return fail(AEV_E_SHAPE, "bad image / kernel shape");
    TRY(select_device(device));
    const int oh = (H - kh) / stride + 1, ow = (W - kw) / stride + 1;
    void *d_img = nullptr, *d_out = nullptr;
    auto body = [&]() -> int {
        const size_t in_sz = (size_t)B * C * H * W * elem_bytes, out_n = (size_t)C * kh * kw * B * oh * ow;
        HIPCHK(hipMalloc(&d_img, in_sz)); HIPCHK(hipMalloc(&d_out, out_n * elem_bytes));
        HIPCHK(hipMemcpy(d_img, image, in_sz, hipMemcpyHostToDevice));
        const dim3 grid(std::min(cdiv((long long)out_n, 256), 16384));
        if (elem_bytes == 4) hipLaunchKernelGGL(k_op_im2col<uint32_t>, grid, dim3(256), 0, 0, (const uint32_t *)d_img, B, C, H, W, kh, kw, stride, oh, ow, (uint32_t *)d_out);
        else hipLaunchKernelGGL(k_op_im2col<u64>, grid, dim3(256), 0, 0, (const u64 *)d_img, B, C, H, W, kh, kw, stride, oh, ow, (u64 *)d_out);
        HIPCHK(hipMemcpy(out_cols, d_out, out_n * elem_bytes, hipMemcpyDeviceToHost));
        HIPCHK(hipGetLastError());
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_img); hipFree(d_out);
    return rc;
}

// ---- next row: YOLO decode + NMS ----------------------------------------------------------------
static int detect_launch(hipStream_t st, const float *d_pred, long long pstride, int S, int gh, int gw, int C, int B, int h_image,
                         int w_image, float conf_thr, int use_nms, float iou_thr, float *d_out, int *d_order, int *d_np)
{
    const int nb = gh * gw * B;
    if (S <= 0 || gh <= 0 || gw <= 0 || C <= 0 || B <= 0 || nb > 4096) return fail(AEV_E_SHAPE, "bad detection grid");
    DetectArgs a;
    a.pred = d_pred; a.pstride = pstride; a.gh = gh; a.gw = gw; a.C = C; a.B = B; a.h_image = h_image; a.w_image = w_image;
    a.conf_thr = conf_thr; a.use_nms = use_nms; a.iou_thr = iou_thr; a.out = d_out; a.order = d_order; a.npicked = d_np;
    const size_t sh = ((size_t)7 * nb + 16) * sizeof(float);
    hipLaunchKernelGGL(k_yolo_detect, dim3(S), dim3(256), sh, st, a);
    return AEV_OK;
}

int aev_yolo_detect(const float *pred, int S, int gh, int gw, int C, int B, int h_image, int w_image, float conf_thr, int use_nms,
                    float iou_thr, float *out, int32_t *order, int32_t *n_picked, int device)
{
    if (!pred || !out) return fail(AEV_E_ARG, "null argument");
    TRY(select_device(device));
    const int nb = gh * gw * B;
    const long long pstride = (long long)gh * gw * (C + 5 * B);
    float *d_pred = nullptr, *d_out = nullptr; int *d_order = nullptr, *d_np = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void **)&d_pred, (size_t)S * pstride * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_out, (size_t)S * nb * 7 * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_order, (size_t)S * nb * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_np, (size_t)S * sizeof(int)));
        HIPCHK(hipMemcpy(d_pred, pred, (size_t)S * pstride * sizeof(float), hipMemcpyHostToDevice));
        TRY(detect_launch(0, d_pred, pstride, S, gh, gw, C, B, h_image, w_image, conf_thr, use_nms, iou_thr, d_out, d_order, d_np));
        HIPCHK(hipMemcpy(out, d_out, (size_t)S * nb * 7 * sizeof(float), hipMemcpyDeviceToHost));
        if (order) HIPCHK(hipMemcpy(order, d_order, (size_t)S * nb * sizeof(int), hipMemcpyDeviceToHost));
        if (n_picked) HIPCHK(hipMemcpy(n_picked, d_np, (size_t)S * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipGetLastError());
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_pred); hipFree(d_out); hipFree(d_order); hipFree(d_np);
    return rc;
}

int aev_nms(const float *boxes, const float *scores, const uint8_t *valid, int nbatch, int n, float iou_thr, int32_t *order,
            int32_t *n_picked, int device)
{
    if (!boxes || !scores || !order || !n_picked) return fail(AEV_E_ARG, "null argument");
    if (nbatch <= 0 || n < 0 || n > 4096) return fail(AEV_E_SHAPE, "bad box count");
    if (n == 0) { for (int b = 0; b < nbatch; ++b) n_picked[b] = 0; return AEV_OK; }
    TRY(select_device(device));
    float *d_b = nullptr, *d_s = nullptr; uint8_t *d_v = nullptr; int *d_o = nullptr, *d_n = nullptr;
    auto body = [&]() -> int {
        const size_t tot = (size_t)nbatch * n;
        HIPCHK(hipMalloc((void **)&d_b, tot * 4 * sizeof(float))); HIPCHK(hipMalloc((void **)&d_s, tot * sizeof(float)));
        HIPCHK(hipMalloc((void **)&d_o, tot * sizeof(int))); HIPCHK(hipMalloc((void **)&d_n, nbatch * sizeof(int)));
        HIPCHK(hipMemcpy(d_b, boxes, tot * 4 * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_s, scores, tot * sizeof(float), hipMemcpyHostToDevice));
        if (valid) { HIPCHK(hipMalloc((void **)&d_v, tot)); HIPCHK(hipMemcpy(d_v, valid, tot, hipMemcpyHostToDevice)); }
        hipLaunchKernelGGL(k_nms, dim3(nbatch), dim3(256), ((size_t)7 * n + 16) * sizeof(float), 0, d_b, d_s, d_v, n, iou_thr, d_o, d_n);
        HIPCHK(hipMemcpy(order, d_o, tot * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(n_picked, d_n, nbatch * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipGetLastError());
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_b); hipFree(d_s); hipFree(d_v); hipFree(d_o); hipFree(d_n);
    return rc;
}

static int net_detect(aev_net *n, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr, float *d_out)
{
    const Layer &l = n->L.back();
    if (l.C != num_classes + 5 * num_bbox) return fail(AEV_E_SHAPE, "last layer channels != num_classes + 5*num_bbox");
    TRY(ensure_scratch(n, (size_t)n->S * l.CHW));
    TRY(run_getter(n, (int)n->L.size() - 1, AEV_FEATUREMAP, 1, 0, n->S, n->d_scratch)); // [S][H][W][C], event_numpy.py:79
    TRY(detect_launch(n->stream, n->d_scratch, l.CHW, n->S, l.H, l.W, num_classes, num_bbox, n->H, n->W, conf_thr, use_nms,
                      iou_thr, d_out, nullptr, nullptr));
    return AEV_OK;
}

int aev_net_detect_device(aev_net *n, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr, void *device_out)
{
    if (!n || !device_out) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    TRY(net_detect(n, num_classes, num_bbox, conf_thr, use_nms, iou_thr, (float *)device_out));
    HIPCHK(hipStreamSynchronize(n->stream));
    HIPCHK(hipGetLastError());
    return AEV_OK;
}

int aev_net_detect(aev_net *n, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr, float *host_out)
{
    if (!n || !host_out) return fail(AEV_E_ARG, "null argument");
    HIPCHK(hipSetDevice(n->device));
    const Layer &l = n->L.back();
    const size_t cnt = (size_t)n->S * l.H * l.W * num_bbox * 7;
    float *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, std::max<size_t>(cnt, 1) * sizeof(float)));
    int rc = aev_net_detect_device(n, num_classes, num_bbox, conf_thr, use_nms, iou_thr, d_out);
    if (rc == AEV_OK && hipMemcpy(host_out, d_out, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(AEV_E_HIP, "copy of detections failed");
    hipFree(d_out);
    return rc;
}

// ---- multi-GPU: the RCCL gather of detections ---------------------------------------------------------
#include "aev_comm.hpp"

// ---- next row: N-MNIST / N-Caltech101 decoder -------------------------------------------------------
int aev_decode_nformat(const uint8_t *raw, long long nbytes, int32_t *x, int32_t *y, int32_t *ts, int32_t *p, long long *n_out,
                       int device)
{
    if (!raw || !x || !y || !ts || !p || !n_out || nbytes < 0) return fail(AEV_E_ARG, "null argument");
    const long long n = nbytes / 5;
    *n_out = 0;
    if (n == 0) return AEV_OK;
    if (n > (1ll << 30)) return fail(AEV_E_SHAPE, "file too large");
    TRY(select_device(device));
    const int nblk = cdiv(n, 1024);
    uint8_t *d_raw = nullptr; int *d_bo = nullptr, *d_bt = nullptr, *d_x = nullptr, *d_y = nullptr, *d_t = nullptr, *d_p = nullptr;
    long long *d_tot = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void **)&d_raw, (size_t)n * 5)); HIPCHK(hipMalloc((void **)&d_bo, nblk * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_bt, nblk * sizeof(int))); HIPCHK(hipMalloc((void **)&d_tot, sizeof(long long)));
        HIPCHK(hipMalloc((void **)&d_x, (size_t)n * sizeof(int))); HIPCHK(hipMalloc((void **)&d_y, (size_t)n * sizeof(int)));
        HIPCHK(hipMalloc((void **)&d_t, (size_t)n * sizeof(int))); HIPCHK(hipMalloc((void **)&d_p, (size_t)n * sizeof(int)));
        HIPCHK(hipMemcpy(d_raw, raw, (size_t)n * 5, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ndec_count, dim3(nblk), dim3(1024), 0, 0, d_raw, n, d_bo, d_bt);
        hipLaunchKernelGGL(k_ndec_scan, dim3(1), dim3(1024), 0, 0, d_bo, d_bt, nblk, d_tot);
        hipLaunchKernelGGL(k_ndec_emit, dim3(nblk), dim3(1024), 0, 0, d_raw, n, d_bo, d_bt, d_x, d_y, d_t, d_p);
        long long m = 0;
        HIPCHK(hipMemcpy(&m, d_tot, sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(x, d_x, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(y, d_y, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ts, d_t, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(p, d_p, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipGetLastError());
        *n_out = m;
        return AEV_OK;
    };
    const int rc = body();
    hipFree(d_raw); hipFree(d_bo); hipFree(d_bt); hipFree(d_tot); hipFree(d_x); hipFree(d_y); hipFree(d_t); hipFree(d_p);
    return rc;
}

} // extern "C"

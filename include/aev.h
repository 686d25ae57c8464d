/*
 * aev.h -- C ABI of libaev_hip.so: the MI355X (gfx950) implementation of the asynchronous
 * (event-driven) sparse conv / max-pool forward path of marcocannici/async-ev-cnn.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; every output buffer is
 * allocated by the caller; the library never returns memory it owns.  Device state lives
 * behind an opaque handle.  One handle = one chain of layers replicated over `n_streams`
 * independent event streams that advance together (one launch sequence serves all streams).
 *
 * Each entry point names the reference interface it replaces (file:line relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Return value: AEV_OK (0) or a negative AEV_E_* code; aev_last_error() gives the
 * thread-local message of the last failure.  Calls on one handle must be serialised by
 * the caller (the reference's layer graph is not re-entrant either); distinct handles may
 * be driven from distinct threads.
 */
#ifndef AEV_H
#define AEV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AEV_OK 0
#define AEV_E_STRIDE (-1)      /* NotImplementedError: stride not 1 and not == kernel (cutils.pyx:89) */
#define AEV_E_SHAPE (-2)       /* ValueError: inconsistent shapes (incl. pooled dim not divisible) */
#define AEV_E_BOUNDS (-3)      /* IndexError: event coordinate outside the map (UB in the reference, cutils.pyx:26-28) */
#define AEV_E_ALLOC (-4)       /* MemoryError: host or device allocation failed */
#define AEV_E_ARG (-5)         /* ValueError: bad argument (unknown padding: conv2d.py:56; bad index; ...) */
#define AEV_E_HIP (-6)         /* RuntimeError: HIP runtime error / no usable GPU */
#define AEV_E_UNSUPPORTED (-7) /* NotImplementedError: valid in the reference but not built here -- see "Chains" below */

/* `which` of aev_layer_read */
#define AEV_SURFACE 0     /* Layer.surface()      layer.py:45  float32 [C,H,W] */
#define AEV_LAYER_ACTFN 1 /* Layer.layer_actfn()  layer.py:52  float32 [C,H,W] */
#define AEV_CONV_ACTFN 2  /* Layer.conv_actfn()   layer.py:59  float32 [C,H,W] */
#define AEV_FEATUREMAP 3  /* Layer.featuremap()   layer.py:77-80 float32 [C,H,W] */
#define AEV_RAW_ACT 4     /* Conv2DLayer._conv_actfn (conv2d.py:62-63) float32 [C,H,W] */
#define AEV_IDX_MAX 5     /* MaxPoolLayer._idx_max[0] (maxpool.py:33-35) int32 [C*H*W] */
#define AEV_RECOMPUTE 6   /* MaxPoolLayer._recompute_coords (maxpool.py:36) uint8 [H,W] */

#define AEV_PAD_VALID 0
#define AEV_PAD_SAME 1

typedef struct aev_net aev_net;

const char *aev_last_error(void);
int aev_version(void);
/* Number of usable HIP devices (0 with AEV_E_HIP when there is none). */
int aev_device_count(int *n);
/* Which HIP runtime the process has mapped against the one the library was built for (diagnostics; needs no GPU):
 * hip_compiled = HIP_VERSION of the hipcc that built the library, hip_runtime / hip_driver = hipRuntimeGetVersion /
 * hipDriverGetVersion of the mapped libamdhip64 (-1 if the call fails), runtime_path = the file that library was
 * loaded from.  Any pointer may be NULL.  (The reference has no counterpart: it has no device code.) */
int aev_runtime_info(int *hip_compiled, int *hip_runtime, int *hip_driver, char *runtime_path, int path_capacity);
/* hipDeviceSynchronize on `device`: every stream of every handle on it has drained (what bench.py brackets its timed region with). */
int aev_device_sync(int device);

/* ---- the two native ops of the reference: src/libs/cutils.pyx ------------------------------ */

/* Replaces im2col_event, cutils.pyx:29-134.  Host pointers; the gather runs on `device`.
 *  image      float32 [C,H,W] C-contiguous (already padded by the caller, conv2d.py:108-110)
 *  ey, ex     int32 [N] event coordinates in image space
 *  out_cols   float32, column-major, capacity rows * outH*outW (* C if chan_as_cols):
 *             chan_as_cols == 0: rows = C*kh*kw, row = c*kh*kw + dy*kw + dx (cutils.pyx:125), column = site
 *             chan_as_cols != 0: rows = kh*kw, column = C*site + c        (cutils.pyx:122-123)
 *  out_y/x    int32, capacity outH*outW, sites in first-touch order (cutils.pyx:108-112)
 *  n_out      number of distinct sites
 * Only stride == 1 or stride == kh == kw (cutils.pyx:78-89), else AEV_E_STRIDE. */
int aev_im2col_event(const float *image, int C, int H, int W, const int32_t *ey, const int32_t *ex, int N, int kh,
                     int kw, int stride, int chan_as_cols, float *out_cols, int32_t *out_y, int32_t *out_x,
                     int *n_out, int device);

/* Replaces min_argmax, cutils.pyx:139-179.  mx, mn: float32 column-major [R,M];
 * argmax, not_argmin: int32 [M]. */
int aev_min_argmax(const float *mx, const float *mn, int R, int M, int32_t *argmax, int32_t *not_argmin,
                   int device);

/* ---- free functions the reference's callers import by name --------------------------------------- */

/* Replaces conv2d_event, src/layers/conv2d.py:144-181 (imported by src/models/frame_numpy.py:5 together with conv2d):
 * the index half of im2col_event followed by kernel_rows . img_cols (+ bias) on the listed sites.  Host pointers.
 *  image        float32 [C,H,W] (padded by the caller, as for im2col_event)
 *  kernel_oihw  float32 [Cout,C,kh,kw] (the layout conv2d.py:26 keeps); bias float32 [Cout] or NULL
 *  out_conv     float32, capacity Cout * outH*outW; receives the [Cout, n] matrix, C order, row stride n (conv2d.py:175-179)
 *  out_y/x      int32, capacity outH*outW, first-touch order; n_out = n
 * float32 fmaf chain in the contract's order, bias added afterwards (the reference leaves the order to BLAS). */
int aev_conv2d_event(const float *image, int C, int H, int W, const int32_t *ey, const int32_t *ex, int N,
                     const float *kernel_oihw, const float *bias, int Cout, int kh, int kw, int stride, float *out_conv,
                     int32_t *out_y, int32_t *out_x, int *n_out, int device);

/* Replaces the dense conv2d, src/layers/conv2d.py:184-230 (conv state init conv2d.py:59-60; the dense NumPy model
 * src/models/frame_numpy.py:74).  image float32 [B,C,H,W]; zero padding given explicitly (the caller evaluates the SAME
 * rule of conv2d.py:199-213); any stride >= 1.  out float32 [B,Cout,outH,outW]. */
int aev_conv2d(const float *image, int B, int C, int H, int W, const float *kernel_oihw, const float *bias, int Cout, int kh,
               int kw, int pad_top, int pad_bottom, int pad_left, int pad_right, int stride, float *out, int device);

/* Replaces the dense im2col, src/layers/functional.py:4-34: image [B,C,H,W] of 4- or 8-byte elements ->
 * out_cols [C*kh*kw][B*outH*outW] C order, row = c*kh*kw + dy*kw + dx, column = b*outH*outW + oy*outW + ox. */
int aev_im2col(const void *image, int elem_bytes, int B, int C, int H, int W, int kh, int kw, int stride, void *out_cols,
               int device);

/* ---- the stateful layer chain: src/layers/{integration,conv2d,maxpool}.py ------------------ */

/* Chains.  A handle is ONE linear chain: every layer is appended behind the current last layer.
 *   * A pool layer directly behind a pool layer (MaxPoolLayer takes any prev_layer, src/layers/maxpool.py:14-40) is accepted when the
 *     stack stands on a conv layer: every pool layer of the stack keeps a copy of its winners (values + activation flags), the next pool
 *     layer pools over that copy, and the conv layer behind the stack reads the top copy.  Such a chain is stepped layer by layer (no
 *     look-ahead sweeps, no deferred tail, no dense frames: aev_net_set_dense answers AEV_E_UNSUPPORTED); results are those of the
 *     reference (tests/golden/exact_poolstack*.npz, float_poolstack32.npz).  A pool stack directly on the integration layer keeps no
 *     copy to pool over: AEV_E_UNSUPPORTED.
 *   * A layer attached to a layer that is NOT the chain's last one (Conv2DLayer / MaxPoolLayer keep whatever prev_layer they are given,
 *     src/layers/conv2d.py:24, maxpool.py:21, so two layers may share one predecessor) is permanently outside this library:
 *     AEV_E_UNSUPPORTED (NotImplementedError in the Python classes).  Device state, event lists and the step plan belong to one chain;
 *     and the reference itself has no use for it: compute_all recurses into prev_layer (conv2d.py:139-140, maxpool.py:163-164), so each
 *     head of a branch would integrate the step's events into the shared layers once more.  A caller who wants a branch builds a second
 *     handle with the shared prefix (the streams are independent).
 * Neither occurs in the reference's own models (src/models/event_numpy.py:53-73 builds a chain; configs/efcn_event.yml:15) or in its test
 * (src/scripts/test_correctness.py:27-31).  Also rejected, as in the reference: conv stride != 1 (AEV_E_STRIDE; the reference passes the
 * stride into the padding slot) and pooled dimensions the window does not divide (AEV_E_SHAPE; out-of-bounds access in the
 * reference). */

/* Replaces IntegrationLayer.__init__, integration.py:12-28: creates a chain whose layer 0 is
 * the leaky integration surface [1,H,W], replicated over n_streams independent streams. */
int aev_net_create(int H, int W, double leak, int n_streams, int device, aev_net **out);
void aev_net_destroy(aev_net *net);

/* Replaces Conv2DLayer.__init__, conv2d.py:15-66.  Appends an async conv + leaky-ReLU layer after
 * the current last layer.  kernel_hwio: float32 [kh,kw,cin,cout] (the layout conv2d.py:19 takes);
 * bias float32 [cout] or NULL; stride must be 1; padding AEV_PAD_VALID / AEV_PAD_SAME.
 * The initial feature map is the dense convolution of the previous layer's CURRENT feature map
 * (conv2d.py:59-61), the initial decay map is zero (conv2d.py:62-63).  Returns the layer index. */
int aev_net_add_conv(aev_net *net, const float *kernel_hwio, const float *bias, int kh, int kw, int cin, int cout,
                     int stride, double alpha, int padding);

/* Replaces MaxPoolLayer.__init__, maxpool.py:14-40.  Needs stride == kh == kw and pooled dims
 * divisible by k.  Returns the layer index. */
int aev_net_add_pool(aev_net *net, int kh, int kw, int stride);

int aev_net_num_layers(const aev_net *net);
int aev_net_num_streams(const aev_net *net);
/* Layer.out_shape(), layer.py:66-70: chw[3] = {C,H,W}. */
int aev_layer_shape(const aev_net *net, int layer, int32_t *chw);

/* Layer.reset() for every layer (integration.py:48-51, conv2d.py:99-103, maxpool.py:84-90), as
 * the model does before a new sample (event_numpy.py:96-98).  stream = -1: all streams. */
int aev_net_reset(aev_net *net, int stream);
/* Layer.reset() of one layer only (same reference lines). */
int aev_layer_reset(aev_net *net, int layer, int stream);

/* Layer.compute(events, delta_leak) of ONE layer of ONE stream with caller-provided events
 * (integration.py:53-91, conv2d.py:105-137, maxpool.py:105-161), the way
 * test_correctness.py:35-39 drives the layers.
 *   layer 0      : ev_a = int32 [n][3] rows (y,x,ts), ev_b unused; *delta_leak is an OUTPUT.
 *   other layers : ev_a = y int32 [n], ev_b = x int32 [n] in the previous layer's output space;
 *                  *delta_leak is an INPUT.
 * Synchronous. The produced events are read back with aev_layer_events. */
int aev_layer_step(aev_net *net, int layer, int stream, const int32_t *ev_a, const int32_t *ev_b, int n,
                   double *delta_leak);

/* The events the layer emitted in its last step (the first return value of Layer.compute):
 * integration / conv: row-major order (np.where, integration.py:83, conv2d.py:131);
 * pool: first-touch order of [incoming events, then pending windows] (maxpool.py:123-126,154). */
int aev_layer_num_events(aev_net *net, int layer, int stream, int *n);
int aev_layer_events(aev_net *net, int layer, int stream, int32_t *y, int32_t *x, int capacity, int *n);

/* Dense getters (layer.py:45-80; conv2d.py:80-94; maxpool.py:42-79).  host_out sized per AEV_* above. */
int aev_layer_read(aev_net *net, int layer, int stream, int which, void *host_out);

/* Layer.compute_all over the whole chain for ALL streams at once (event_numpy.py:75-79).
 *   yxt      int32 [offsets[S]][3] rows (y,x,ts), the streams' batches concatenated
 *   offsets  int32 [S+1]; stream s owns rows offsets[s] .. offsets[s+1]-1; an empty batch leaves that
 *            stream untouched (no call is made for it in the reference either)
 * Host pointers, copied into pinned staging before the call returns (the caller may reuse its buffers at once);
 * asynchronous on the handle's HIP stream (aev_net_sync waits).  Two staging slots let the host prepare batch t+1
 * while batch t is running. */
int aev_net_step(aev_net *net, const int32_t *yxt, const int32_t *offsets);

/* Device-resident event queue: T consecutive steps for all streams, uploaded once.
 *   yxt int32 [offsets[T*S]][3]; offsets int64 [T*S+1], step t / stream s owns rows
 *   offsets[t*S+s] .. offsets[t*S+s+1]-1.  Replaces any previous queue. */
int aev_net_queue_upload(aev_net *net, const int32_t *yxt, const int64_t *offsets, int T);
/* The feeder on the device.  Replaces Runner.data_transform (src/libs/runner.py:23-33: ts -= ts[0]; utils.center_crop,
 * src/libs/utils.py:4-35, when the sample's sensor size differs from the handle's surface -- its swapped-offset quirk kept -- ; stack
 * (y, x, ts)) and the batching rule of Runner.run (runner.py:65-72) for one WHOLE sample per stream, and builds the event queue from it:
 *   x, y, ts    int32 per event, the samples concatenated (host pointers); a sample's timestamps non-decreasing
 *   sample_off  int64 [S+1]: stream s owns events sample_off[s] .. sample_off[s+1]-1
 *   example_h/w the samples' sensor size (args.example_h / example_w); crop to the handle's H x W if it differs
 *   batch_event_usec > 0: time bins of that width (np.digitize, runner.py:66-69); else chunks of ~batch_event_size events
 *                         (np.array_split into ceil(n / size) parts, runner.py:71-72)
 *   *T_out      number of queue steps = the most chunks any stream has (a stream with fewer is idle in the remaining steps)
 * Replaces any previous queue; run it with aev_net_queue_run. */
int aev_net_queue_upload_samples(aev_net *net, const int32_t *x, const int32_t *y, const int32_t *ts, const int64_t *sample_off,
                                 int example_h, int example_w, int batch_event_size, int batch_event_usec, int *T_out);
/* Copy of the current event queue (tests, inspection): *T_out steps; offsets int64 [T*S+1] and yxt int32 [rows][3] may be NULL. */
int aev_net_queue_download(aev_net *net, int32_t *yxt, long long row_capacity, int64_t *offsets, int *T_out);
/* Run queue steps t0 .. t0+T-1 back to back (asynchronous, no host round trip between steps).
 * use_graph != 0 replays captured hipGraphs (of 32, 16, 8, 4, 2 and one steps: the longest that still fit).
 * Within one call steps overlap.  With many streams (from 232 on, where the last tile-kernel layer's work items fill the rounds of its
 * persistent grid -- 256, 384, 512 ... e-fcn streams; aev_hip.hip: choose_plan): the leak extrapolation of the first conv layers for
 * step t+1 (conv2d.py:115)
 * runs during step t, beside its deep layers -- the device knows the next step's timestamps -- and the trailing 1x1 conv layers of
 * step t are computed while step t+1 works on its first layers.  Below: the bookkeeping of a step (event lists, window lists, touch
 * maps: it does not depend on the values the step computes) runs ahead of the conv / pool evaluations on a second HIP stream, and the
 * back half of step t (from the last conv layer of the spatial-tile kernel on) is a pipeline stage beside the front half of step
 * t+1.  The last step of a call neither looks ahead nor leaves anything behind, so the
 * state a call leaves is the state the same steps leave through aev_net_step, bit for bit, and any entry point may follow. */
int aev_net_queue_run(aev_net *net, int t0, int T, int use_graph);
/* Same, bracketed by HIP events on the handle's stream; *ms = elapsed device time. Synchronous. */
int aev_net_queue_run_timed(aev_net *net, int t0, int T, int use_graph, float *ms);
/* Same steps run eagerly with HIP events around every kernel launch.  ms_by_kernel and launches_by_kernel are
 * [num_layers][AEV_NUM_KERNELS] matrices: summed duration / number of launches per (layer, kernel class); the one
 * leak sweep that serves every conv layer is booked on layer 0.  Synchronous. */
#define AEV_NUM_KERNELS 12
int aev_net_queue_profile(aev_net *net, int t0, int T, float *ms_by_kernel, int64_t *launches_by_kernel);
const char *aev_kernel_name(int kernel_class);

int aev_net_sync(aev_net *net);

/* The model output: last layer's featuremap().transpose(1,2,0) (event_numpy.py:79,101), for every
 * stream: float32 [S][H][W][C].  host_out: host pointer (synchronous copy). */
int aev_net_output(aev_net *net, float *host_out);
/* Same into a DEVICE pointer (e.g. a torch tensor's data_ptr used as an RCCL send buffer);
 * enqueued on the handle's stream, then the stream is synchronised. */
int aev_net_output_device(aev_net *net, void *device_out);

/* Work counters since creation / last aev_net_stats_reset, summed over streams:
 *   sites[l]  conv: output sites recomputed; pool: windows re-evaluated; integration: events emitted
 *   events_in total input events consumed by layer 0.            sites: int64 [num_layers]. */
int aev_net_stats(aev_net *net, int64_t *sites, int64_t *events_in, int64_t *steps);
int aev_net_stats_reset(aev_net *net);

/* Kernel-configuration policy, to be set before any layer is added: 0 = auto (chosen per layer from the stream count:
 * many streams -> large cout blocks / column tiles across streams; few streams -> small blocks that spread one tile over
 * many CUs), 1 = always the throughput configurations, 2 = always the few-streams configurations.  Results are
 * identical in every mode (same arithmetic contract); the switch exists so that both families are testable at any size. */
int aev_net_set_tuning(aev_net *net, int tuning);

/* Dense-frame mode (BASELINE config 5 comparator; src/models/frame_numpy.py:62-88 is the reference's dense model): when
 * set, every step recomputes EVERY site of every layer from the current input surface through the same kernels (no
 * leak extrapolation, every pooling window re-evaluated).  Used to report the async-vs-dense speed-up on the same device.
 *   0  asynchronous (default)
 *   1  dense, activation once -- the semantics of the event path and of the TF comparator (test_correctness.py:73-80):
 *      same result as the asynchronous mode up to float rounding
 *   2  dense, activation applied again after every pool layer, as frame_numpy.py:77-78 does: what the
 *      reference's dense NumPy model returns */
int aev_net_set_dense(aev_net *net, int dense);

/* Arithmetic of the 3x3 tile-kernel conv layers with Cin % 16 == 0 and Cout >= 64, to be set before any layer is added (the reference
 * leaves the contraction to BLAS, conv2d.py:175 -- "summation order unspecified"):
 *   0  AEV_MATH_F32 (default)  f32 MFMA, the fmaf chain of the arithmetic contract: bit-identical to the CPU oracle
 *   1  AEV_MATH_BF16X6         every f32 operand split EXACTLY into three bf16 parts, six bf16 products accumulated in f32 on the bf16
 *                              MFMA (csrc/aev_conv_x6.hpp): 2.4x the MFMA rate; error against float64 no larger than the f32 chain's
 *                              (tools/probes/mfma_bf16x6.hip), but not bit-identical to it -- float maps agree with the oracle to north_star's
 *                              1e-5, a sign decision on a value within ~1e-6 of zero may differ; exact for small-integer / dyadic operands. */
#define AEV_MATH_F32 0
#define AEV_MATH_BF16X6 1
int aev_net_set_conv_math(aev_net *net, int math);

/* Implementation selector for the conv site-recompute kernel: 0 = auto, 1 = VALU kernel only,
 * 2 = f32 MFMA kernel wherever its shape constraints hold.  Both are HIP kernels with identical results. */
int aev_net_set_conv_impl(aev_net *net, int impl);

/* ---- next row after the path: YOLO decode + NMS -------------------------------- */

/* Replaces draw_bboxes up to (not including) the drawing, src/libs/viz.py:125-164: convert_bboxes (viz.py:27-46, sqrt
 * form), class = argmax(label * conf), valid = conf > conf_thr, optional apply_nms (src/libs/utils.py:38-118).
 *   pred  float32 [S][grid_h][grid_w][C + 5*B]  (the model output, event_numpy.py:101), host pointer
 *   out   float32 [S][nb = grid_h*grid_w*B][7] = x, y, w, h (pixels, centre form), conf, class index, valid (0/1);
 *         with NMS the boxes that were not picked are zeroed as viz.py:154-164 does
 *   order int32 [S][nb] (nullable): original indices in the order NMS picked them; n_picked int32 [S] (nullable)
 * NMS visits by descending score, equal scores by descending index. */
int aev_yolo_detect(const float *pred, int S, int grid_h, int grid_w, int num_classes, int num_bbox, int h_image, int w_image,
                    float conf_thr, int use_nms, float iou_thr, float *out, int32_t *order, int32_t *n_picked, int device);
/* Replaces apply_nms, src/libs/utils.py:38-118, on caller-provided boxes: boxes float32 [nbatch][n][4] (centre x, y, w, h),
 * scores float32 [nbatch][n], valid uint8 [nbatch][n] (nullable) -> order int32 [nbatch][n], n_picked int32 [nbatch]. */
int aev_nms(const float *boxes, const float *scores, const uint8_t *valid, int nbatch, int n, float iou_thr, int32_t *order,
            int32_t *n_picked, int device);
/* The same decode + NMS applied to the handle's current output for every stream; image size = the input surface.
 * host_out / device_out: float32 [S][nb][7].  The device form feeds the RCCL gather of detections. */
int aev_net_detect(aev_net *net, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr, float *host_out);
int aev_net_detect_device(aev_net *net, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr,
                          void *device_out);

/* ---- multi-GPU (BASELINE config 4): the one collective of the path ------------------------------------
 * The reference has no distributed code (no NCCL / MPI call sites); north_star asks for independent streams sharded over
 * the GPUs of a node "with RCCL over xGMI only for the final gather of predictions".  One process per GPU; each rank owns its handle
 * and its streams, nothing is exchanged during compute.  RCCL (librccl.so.1 of the ROCm the library was built with) is bound at run
 * time by the first call below, so a rank needs no PyTorch and runs on the HIP runtime the library was built for.  The 128-byte
 * id is ncclUniqueId: rank 0 creates it and hands it to the other ranks over any side channel (bench.py: a TCP socket on
 * MASTER_ADDR : MASTER_PORT + 1; async_ev_cnn_amd/comm.py). */
typedef struct aev_comm aev_comm;
#define AEV_COMM_ID_BYTES 128
#define AEV_REDUCE_SUM 0
#define AEV_REDUCE_MAX 1
int aev_comm_unique_id(uint8_t *id /* [AEV_COMM_ID_BYTES] */);
/* ncclCommInitRank on `device` (collective: every rank of the world calls it with the same id). */
int aev_comm_create(int rank, int world, const uint8_t *id, int device, aev_comm **out);
void aev_comm_destroy(aev_comm *comm);
int aev_comm_rank(const aev_comm *comm);
int aev_comm_world(const aev_comm *comm);
/* All-reduce of n <= 4096 host doubles in place (what bench.py needs around its timed region: MAX of the ranks' elapsed times, SUM of
 * their event counts; with n = 1 it doubles as a barrier).  Synchronous. */
int aev_comm_allreduce(aev_comm *comm, double *values, int n, int op);
/* The gather of detections: aev_net_detect of this rank's streams, all-gathered over the communicator.  Ranks may hold different
 * numbers of streams (stream i -> rank i mod world): every rank pads to the largest count.
 *   counts     int32 [world]: streams of every rank;  *s_max_out = the largest of them (must be <= s_cap)
 *   host_out   float32 [world][*s_max_out][grid_h * grid_w * num_bbox][7] (capacity world * s_cap * ...); rows behind a rank's count are 0
 * Synchronous; identical on every rank. */
int aev_net_gather_detections(aev_net *net, aev_comm *comm, int num_classes, int num_bbox, float conf_thr, int use_nms, float iou_thr,
                              int s_cap, float *host_out, int32_t *counts, int *s_max_out);

/* ---- next row before the path: N-MNIST / N-Caltech101 recordings ------------------- */

/* Replaces NReader.read_example, src/readers/file_reader.py:30-58, on a byte buffer (host pointers): 5 bytes per record
 * (x, y, p<<7 | ts[22:16], ts[15:8], ts[7:0]); records with y == 240 are timestamp-overflow markers (+2^13 from there on)
 * and are dropped.  x, y, ts, p: int32, capacity nbytes/5; *n_out = number of events. */
int aev_decode_nformat(const uint8_t *raw, long long nbytes, int32_t *x, int32_t *y, int32_t *ts, int32_t *p, long long *n_out,
                       int device);

#ifdef __cplusplus
}
#endif
#endif /* AEV_H */

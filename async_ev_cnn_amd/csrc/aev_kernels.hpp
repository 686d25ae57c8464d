// aev_kernels.hpp -- the gfx950 kernels of the asynchronous conv / max-pool path.
// Included once by aev_hip.hip.  Every kernel cites the reference lines whose effect it produces.
//
// Launch convention: blockIdx.y (+ s0) is the event stream; blockIdx.x walks the work items of that
// stream in a grid-stride loop whose bound is a device-side count, so every wave terminates whatever
// the counts are and the launch geometry never depends on data (hipGraph-friendly).
#pragma once
#include <climits>
#include "aev_device.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Loads through a block-uniform base pointer + an unsigned 32-bit element offset: hipcc selects the
// `global_load v, v_off, s[base:base+1]` form (no 64-bit VALU address arithmetic per element).
__device__ __forceinline__ float ld_u32(const float *base, unsigned off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)(off * 4u));
}
__device__ __forceinline__ f32x4 ld_u32x4(const float *base, unsigned off)
{
    return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + (size_t)(off * 4u));
}
__device__ __forceinline__ int ld_u8(const uint8_t *base, unsigned off) { return base[(size_t)off]; }
__device__ __forceinline__ void st_u32(float *base, unsigned off, float v)
{
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (size_t)(off * 4u)) = v;
}

// ------------------------------------------------------------------------------------------------
// Integration layer (src/layers/integration.py:53-91)
// ------------------------------------------------------------------------------------------------
struct IntegArgs {
    const int *yxt;        // [rows][3] (y,x,ts)
    const long long *off;  // row offsets, [qbase*S + s]
    const int *qstep;      // device step counter of the event queue (nullptr: base 0)
    int S, s0;
    int H, W, HW;
    double leak;
    float *surf;           // [S][sstride]
    long long sstride;
    int *owner;            // [S][HW] index of the LAST event of this batch on the pixel, -1 otherwise
    long long *prev_ts;    // [S]
    long long *last_ts;    // [S]
    double *dl64;          // [S] delta_leak (integration.py:60)
    float *dlf;            // [S] the same rounded once to float32
    int *active;           // [S] 1 if the stream has events in this step
    int *zero; int nzero;  // scratch counters of the step (the tile kernels' list sizes / work queues), cleared here: the first kernel of a step
    uint8_t *evmap;        // [S][mstride] output event byte map (1 = event)
    long long mstride;
    u64 *stat_events;
    // Look-ahead (queued steps only, else nullptr): delta_leak and activity of the NEXT queue step, for the leak sweeps that
    // run a step early beside the deep conv layers (k_leak_sweep_ahead).  qrun = {first step, end} of the run being executed:
    // the next step is looked at only if it belongs to the same run (nothing can come between two steps of a run).
    const int *qrun;
    float *dlf_next;       // [S]
    int *active_next;      // [S]
};

// One block per stream: last timestamp, delta_leak, last-writer table; and delta_leak of the next queue step (look-ahead).
__global__ __launch_bounds__(256) void k_integ_prepare(IntegArgs a)
{
    const int s = blockIdx.x + a.s0;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && a.zero)
        for (int i = tid; i < a.nzero; i += (int)blockDim.x) a.zero[i] = 0;
    const int q = a.qstep ? *a.qstep : 0;
    const long long base = (long long)q * a.S;
    const long long r0 = a.off[base + s];
    const int n = (int)(a.off[base + s + 1] - r0);
    const bool ahead = a.qrun != nullptr && q + 1 < a.qrun[1];
    long long r1 = 0;
    int n1 = 0;
    if (ahead) { r1 = a.off[base + a.S + s]; n1 = (int)(a.off[base + a.S + s + 1] - r1); }
    if (n <= 0 && n1 <= 0) { // (block-uniform)
        if (tid == 0) { a.active[s] = 0; if (a.active_next) a.active_next[s] = 0; }
        return;
    }
    __shared__ int red[8];
    int m = INT_MIN, m1 = INT_MIN;
    for (int i = tid; i < n; i += 256) m = max(m, a.yxt[(r0 + i) * 3 + 2]);
    for (int i = tid; i < n1; i += 256) m1 = max(m1, a.yxt[(r1 + i) * 3 + 2]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { m = max(m, __shfl_xor(m, d, 64)); m1 = max(m1, __shfl_xor(m1, d, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = m; red[4 + (tid >> 6)] = m1; }
    __syncthreads();
    if (tid == 0) {
        long long prev = a.prev_ts[s];
        if (n > 0) {
            const long long last = max(max(red[0], red[1]), max(red[2], red[3]));
            const double dl = (double)(last - prev) * a.leak; // integration.py:60
            a.dl64[s] = dl;
            a.dlf[s] = (float)dl;
            a.last_ts[s] = last;
            a.prev_ts[s] = last; // integration.py:86
            a.active[s] = 1;
            atomicAdd(a.stat_events, (u64)n);
            prev = last;
        } else
            a.active[s] = 0;
        if (a.active_next) {
            if (n1 > 0) { // the next step's integration.py:60, from the same operands its own k_integ_prepare will use
                const long long last1 = max(max(red[4], red[5]), max(red[6], red[7]));
                a.dlf_next[s] = (float)((double)(last1 - prev) * a.leak);
                a.active_next[s] = 1;
            } else
                a.active_next[s] = 0;
        }
    }
    // integration.py:71 is a fancy-index "+=": duplicates do not accumulate, the last one wins
    int *own = a.owner + (long long)s * a.HW;
    for (int i = tid; i < n; i += 256) {
        const int y = a.yxt[(r0 + i) * 3], x = a.yxt[(r0 + i) * 3 + 1];
        atomicMax(&own[y * a.W + x], i);
    }
}

// Prologue of a queue run with the look-ahead: delta_leak / activity of the run's FIRST step into dlf_next / active_next, from the
// operands its k_integ_prepare will use (nothing is changed) -- the sweeps of the look-ahead layers for that step are then
// launched ahead of it like those of every later step (k_leak_sweep_ahead), and no step has to ask whether it was swept.
__global__ __launch_bounds__(256) void k_integ_peek(IntegArgs a)
{
    const int s = blockIdx.x + a.s0;
    const int tid = threadIdx.x;
    const long long base = (long long)(*a.qstep) * a.S;
    const long long r0 = a.off[base + s];
    const int n = (int)(a.off[base + s + 1] - r0);
    if (n <= 0) { if (tid == 0) a.active_next[s] = 0; return; } // (block-uniform)
    __shared__ int red[4];
    int m = INT_MIN;
    for (int i = tid; i < n; i += 256) m = max(m, a.yxt[(r0 + i) * 3 + 2]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        const long long last = max(max(red[0], red[1]), max(red[2], red[3]));
        a.dlf_next[s] = (float)((double)(last - a.prev_ts[s]) * a.leak); // integration.py:60, as k_integ_prepare
        a.active_next[s] = 1;
    }
}

// Dense sweep of the surface: leak, clamp, add the events, clamp, event mask -> byte map.  Four pixels per thread (16-byte
// accesses of the surface and the last-writer table, one 4-byte store of the event map).
__device__ __forceinline__ float integ_pixel(const IntegArgs &a, int s, long long r0, float v, float dl, int o, bool &ev)
{
    const bool was_pos = v > 0.0f;            // :63
    v = __fsub_rn(v, dl);                     // :65
    bool neg = v <= 0.0f;                     // :66
    if (neg) v = 0.0f;                        // :68
    if (o >= 0) {
        const int ts = a.yxt[(r0 + o) * 3 + 2];
        const double add = 1.0 - (double)(a.last_ts[s] - (long long)ts) * a.leak; // :71 (float64 term)
        v = (float)((double)v + add);
    }
    if (v <= 0.0f) { v = 0.0f; neg = true; }  // :72-74
    ev = (was_pos && neg) || (o >= 0);        // :78-80
    return v;
}
__global__ __launch_bounds__(256) void k_integ_sweep(IntegArgs a)
{
    const int s = blockIdx.y + a.s0;
    if (!a.active[s]) return;
    const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= a.HW) return;
    float *S = a.surf + (long long)s * a.sstride;
    int *own = a.owner + (long long)s * a.HW;
    uint8_t *E = a.evmap + (long long)s * a.mstride;
    const float dl = a.dlf[s];
    const long long base = a.qstep ? (long long)(*a.qstep) * a.S : 0;
    const long long r0 = a.off[base + s];
    if ((a.HW & 3) == 0) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        f32x4 v = *reinterpret_cast<const f32x4 *>(S + p0);
        const i32x4 o = *reinterpret_cast<const i32x4 *>(own + p0);
        unsigned evb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ev;
            v[j] = integ_pixel(a, s, r0, v[j], dl, o[j], ev);
            evb |= ev ? 1u << (8 * j) : 0u;
        }
        *reinterpret_cast<f32x4 *>(S + p0) = v;
        if ((o[0] & o[1] & o[2] & o[3]) >= 0) *reinterpret_cast<i32x4 *>(own + p0) = (i32x4){-1, -1, -1, -1}; // (some entry >= 0: sign bit of the AND clear)
        *reinterpret_cast<unsigned *>(E + p0) = evb;
    } else {
        for (int j = 0; j < 4 && p0 + j < a.HW; ++j) {
            bool ev;
            const int o = own[p0 + j];
            S[p0 + j] = integ_pixel(a, s, r0, S[p0 + j], dl, o, ev);
            if (o >= 0) own[p0 + j] = -1;
            E[p0 + j] = ev ? 1 : 0;
        }
    }
}

// k_integ_prepare + k_integ_sweep in ONE launch, for steps of at most AEV_INTEG_FUSED_MAX events per stream (the chain's steps: the
// first two launches of every step were a dependent pair).  A block sweeps 1024 pixels of one stream and finds what it needs from
// the step's events itself: the last timestamp (hence delta_leak -- every block of a stream computes the same value from the same
// operands) and, in LDS instead of the global last-writer table, the last event on each of ITS pixels.  Block 0 of a stream writes
// the per-stream results (delta_leak, activity, the look-ahead's values for the next step).  prev_ts is read by every block of the
// stream and therefore NOT written here: k_post of the input layer (one block per stream, the next kernel of the chain that is one)
// moves it on (PostArgs::prev_ts).
#define AEV_INTEG_FUSED_MAX 2048
__device__ __forceinline__ void integ_step_body(const IntegArgs &a, int bx)
{
    const int s = blockIdx.y + a.s0;
    const int tid = threadIdx.x;
    const bool first = bx == 0;
    if (first && blockIdx.y == 0 && a.zero)
        for (int i = tid; i < a.nzero; i += 256) a.zero[i] = 0;
    const int q = a.qstep ? *a.qstep : 0;
    const long long base = (long long)q * a.S;
    const long long r0 = a.off[base + s];
    const int n = (int)(a.off[base + s + 1] - r0);
    const bool ahead = first && a.qrun != nullptr && q + 1 < a.qrun[1];
    long long r1 = 0;
    int n1 = 0;
    if (ahead) { r1 = a.off[base + a.S + s]; n1 = (int)(a.off[base + a.S + s + 1] - r1); }
    if (n <= 0 && n1 <= 0) { // (block-uniform)
        if (first && tid == 0) { a.active[s] = 0; if (a.active_next) a.active_next[s] = 0; }
        return;
    }
    __shared__ int red[8];
    __shared__ int own_l[1024]; // index of the LAST event of this step on the block's pixel, -1: none (integration.py:71: the last one wins)
    const int pb = bx * 1024;
    for (int i = tid; i < 1024; i += 256) own_l[i] = -1;
    __syncthreads();
    int m = INT_MIN, m1 = INT_MIN;
    for (int i = tid; i < n; i += 256) {
        const int y = a.yxt[(r0 + i) * 3], x = a.yxt[(r0 + i) * 3 + 1];
        m = max(m, a.yxt[(r0 + i) * 3 + 2]);
        const int p = y * a.W + x - pb;
        if ((unsigned)p < 1024u) atomicMax(&own_l[p], i);
    }
    for (int i = tid; i < n1; i += 256) m1 = max(m1, a.yxt[(r1 + i) * 3 + 2]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { m = max(m, __shfl_xor(m, d, 64)); m1 = max(m1, __shfl_xor(m1, d, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = m; red[4 + (tid >> 6)] = m1; }
    __syncthreads();
    const long long prev0 = a.prev_ts[s];
    const long long last = max(max(red[0], red[1]), max(red[2], red[3]));
    const double dl64 = (double)(last - prev0) * a.leak; // integration.py:60
    if (first && tid == 0) {
        long long prev = prev0;
        if (n > 0) {
            a.dl64[s] = dl64;
            a.dlf[s] = (float)dl64;
            a.last_ts[s] = last; // (prev_ts <- last_ts: k_post of the input layer, integration.py:86)
            a.active[s] = 1;
            atomicAdd(a.stat_events, (u64)n);
            prev = last;
        } else
            a.active[s] = 0;
        if (a.active_next) {
            if (n1 > 0) { // the next step's integration.py:60, from the same operands its own launch will use
                const long long last1 = max(max(red[4], red[5]), max(red[6], red[7]));
                a.dlf_next[s] = (float)((double)(last1 - prev) * a.leak);
                a.active_next[s] = 1;
            } else
                a.active_next[s] = 0;
        }
    }
    if (n <= 0) return;
    const int p0 = pb + tid * 4;
    if (p0 >= a.HW) return;
    float *S = a.surf + (long long)s * a.sstride;
    uint8_t *E = a.evmap + (long long)s * a.mstride;
    const float dl = (float)dl64;
    auto pixel = [&](float v, int o, bool &ev) __attribute__((always_inline)) { // as integ_pixel, with the block's own last timestamp
        const bool was_pos = v > 0.0f;            // :63
        v = __fsub_rn(v, dl);                     // :65
        bool neg = v <= 0.0f;                     // :66
        if (neg) v = 0.0f;                        // :68
        if (o >= 0) {
            const int ts = a.yxt[(r0 + o) * 3 + 2];
            const double add = 1.0 - (double)(last - (long long)ts) * a.leak; // :71 (float64 term)
            v = (float)((double)v + add);
        }
        if (v <= 0.0f) { v = 0.0f; neg = true; }  // :72-74
        ev = (was_pos && neg) || (o >= 0);        // :78-80
        return v;
    };
    if ((a.HW & 3) == 0) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(S + p0);
        unsigned evb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ev;
            v[j] = pixel(v[j], own_l[tid * 4 + j], ev);
            evb |= ev ? 1u << (8 * j) : 0u;
        }
        *reinterpret_cast<f32x4 *>(S + p0) = v;
        *reinterpret_cast<unsigned *>(E + p0) = evb;
    } else {
        for (int j = 0; j < 4 && p0 + j < a.HW; ++j) {
            bool ev;
            S[p0 + j] = pixel(S[p0 + j], own_l[tid * 4 + j], ev);
            E[p0 + j] = ev ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(256) void k_integ_step(IntegArgs a) { integ_step_body(a, (int)blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// Leak extrapolation of every conv layer of every stream + sign-change detection
// (src/layers/conv2d.py:113,115,125-127).  Pure streaming: 12 B per state element.
// ------------------------------------------------------------------------------------------------
#define AEV_MAX_SEGS 16
#define AEV_SWEEP_CHUNK 4096    // elements per block
#ifndef AEV_SWEEP_CHUNK_LA
#define AEV_SWEEP_CHUNK_LA 3072 // elements per work item of the look-ahead kernel
#endif
struct SweepSeg {
    float *fm;
    const float *act;
    uint8_t *changed;      // [S][mstride] byte map: 1 = some channel of the site changed sign (nullptr: a pool layer's copy, nobody asks)
    long long sstride, mstride;
    int CHW, C, chunk0;
    // "may decay" byte per site (nullptr: none; only for power-of-two C <= 256): 0 = every channel's decay is exactly 0, the
    // site is not even looked at.  Set by whoever writes decay values (k_post for recomputed sites, k_pool_eval for the
    // windows it rewrites), cleared here, lazily, when a flagged site turns out to be all zero.
    uint8_t *flag; long long fstride; int cshift;
};
struct SweepArgs {
    SweepSeg seg[AEV_MAX_SEGS];
    int nseg, s0;
    const float *dlf;
    const int *active;
    // Look-ahead (queued steps, k_leak_sweep_ahead): the sweep is for step *qstep + 1 and runs only if that step belongs to the
    // run being executed (qrun = {first step, end}); `prologue`: for step *qstep itself, before the run's first step.
    const int *qstep, *qrun;
    int prologue;
    int chunks, ns; // k_leak_sweep_ahead: chunks per stream over all segments, streams
};

// One chunk (IT * 1024 elements: IT float4s per thread) of one segment of stream s, in three parts without any block-wide
// synchronisation: sweep_flags requests the chunk's "may decay" bytes, sweep_issue the decay and feature values of the flagged
// sites, sweep_finish does the arithmetic and the stores.  NO load sits under a condition: hipcc answers a guarded load with
// a branch and, at the join, an s_waitcnt vmcnt(0) -- the loads of a thread then complete one by one.  A lane with nothing to
// fetch loads the first bytes of the segment instead (one hot line) and ignores them.
// The arithmetic of one float4 of the sweep, written for few instructions (the look-ahead kernel runs one wave per SIMD, beside
// MFMA waves: every instruction counts).
//   all_zero: the four decay values are +-0.0 <=> the OR of their bit patterns has nothing but (possibly) the sign bit set;
//   leak4   : r = f - (d * dl), two roundings (conv2d.py:115); returns whether some channel MAY have changed sign
//             ((f >= 0) != (r >= 0), conv2d.py:113,126) -- only if |f| <= |d * dl| (or something is NaN): otherwise r keeps f's
//             sign and is not zero.  The exact test runs under that (rare) condition only.
__device__ __forceinline__ bool all_zero(const f32x4 &d)
{
    return ((__float_as_uint(d.x) | __float_as_uint(d.y) | __float_as_uint(d.z) | __float_as_uint(d.w)) << 1) == 0u;
}
__device__ __forceinline__ bool leak4(const f32x4 &d, const f32x4 &f, float dl, f32x4 &r)
{
    const float mx = __fmul_rn(d.x, dl), my = __fmul_rn(d.y, dl), mz = __fmul_rn(d.z, dl), mw = __fmul_rn(d.w, dl);
    r.x = __fsub_rn(f.x, mx); r.y = __fsub_rn(f.y, my); r.z = __fsub_rn(f.z, mz); r.w = __fsub_rn(f.w, mw);
    return !(fabsf(f.x) > fabsf(mx)) || !(fabsf(f.y) > fabsf(my)) || !(fabsf(f.z) > fabsf(mz)) || !(fabsf(f.w) > fabsf(mw));
}
__device__ __forceinline__ bool flipped4(const f32x4 &f, const f32x4 &r)
{
    return (f.x >= 0.0f) != (r.x >= 0.0f) || (f.y >= 0.0f) != (r.y >= 0.0f) || (f.z >= 0.0f) != (r.z >= 0.0f) || (f.w >= 0.0f) != (r.w >= 0.0f);
}

template <int IT>
__device__ __forceinline__ void sweep_flags(const SweepSeg &sg, int s, int chunk, int (&fl)[IT])
{
    const bool has = sg.flag != nullptr;
    const uint8_t *fp = has ? sg.flag + (long long)s * sg.fstride : reinterpret_cast<const uint8_t *>(sg.act); // (no flags: a dummy read)
    const int e0 = chunk * (IT * 1024);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = e0 + (it * 256 + threadIdx.x) * 4;
        const bool in = has && e + 4 <= sg.CHW;
        const int v = fp[in ? e >> sg.cshift : 0];
        fl[it] = in ? v : 1;
    }
}
template <int IT>
__device__ __forceinline__ void sweep_issue(const SweepSeg &sg, int s, int chunk, const int (&fl)[IT], f32x4 (&dv)[IT], f32x4 (&fv)[IT])
{
    const float *fm = sg.fm + (long long)s * sg.sstride;
    const float *act = sg.act + (long long)s * sg.sstride;
    const int e0 = chunk * (IT * 1024);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = e0 + (it * 256 + threadIdx.x) * 4;
        const int ea = (fl[it] && e + 4 <= sg.CHW) ? e : 0;
        // streamed once per step and not re-read soon: non-temporal accesses keep the weights / patches in L2
        dv[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(act + ea));
        fv[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(fm + ea));
    }
}
template <int IT>
__device__ __forceinline__ void sweep_finish(const SweepSeg &sg, int s, int chunk, float dl, const int (&fl)[IT], const f32x4 (&dv)[IT],
                                             const f32x4 (&fv)[IT])
{
    float *fm = sg.fm + (long long)s * sg.sstride;
    const float *act = sg.act + (long long)s * sg.sstride;
    const int e0 = chunk * (IT * 1024);
    // A site whose decay is exactly 0 does not move (fm - 0*dl == fm, no sign change): its feature values are neither
    // read nor written.  Decay is zero for whole sites (all channels) wherever nothing is active in the receptive field,
    // i.e. for contiguous 64+ byte runs of the channel-fastest state -- about half of conv1 on N-Caltech101-like streams.
    // (A flagged site whose decay turns out to be zero -- the lazy un-flagging below has not caught up with it -- costs one
    //  read of its feature values in vain.)
    bool live[IT]; // this lane's float4 was fetched and some decay in it is not zero
    uint8_t *flag = sg.flag ? sg.flag + (long long)s * sg.fstride : nullptr;
    const int lane = threadIdx.x & 63, gsz = flag ? 1 << (sg.cshift - 2) : 1;
    const u64 grp = (gsz == 64 ? ~0ull : ((1ull << gsz) - 1)) << (lane & ~(gsz - 1));
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = e0 + (it * 256 + threadIdx.x) * 4;
        const f32x4 d = dv[it];
        const bool got = fl[it] && e + 4 <= sg.CHW;
        const bool z = all_zero(d);
        live[it] = got && !z;
        if (flag) { // a flagged site whose decay turned out to be all zero: un-flag it (the lanes of a site are neighbours in the wave)
            const u64 bal = __ballot(got && z);
            if ((bal & grp) == grp && (lane & (gsz - 1)) == 0) flag[e >> sg.cshift] = 0;
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = e0 + (it * 256 + threadIdx.x) * 4;
        if (live[it]) {
            const f32x4 d = dv[it], f = fv[it];
            f32x4 r;
            const bool maybe = leak4(d, f, dl, r);
            __builtin_nontemporal_store(r, reinterpret_cast<f32x4 *>(fm + e));
            if (maybe && sg.changed) { // rare
                const float fa[4] = {f.x, f.y, f.z, f.w}, ra[4] = {r.x, r.y, r.z, r.w};
                for (int j = 0; j < 4; ++j)
                    if ((fa[j] >= 0.0f) != (ra[j] >= 0.0f)) sg.changed[(long long)s * sg.mstride + (e + j) / sg.C] = 1; // same-value stores: no atomic needed
            }
        } else if (e < sg.CHW && e + 4 > sg.CHW) { // the last, partial float4 of a segment whose size is not a multiple of 4
            for (int j = 0; j < 4; ++j)
                if (e + j < sg.CHW) {
                    const float f = fm[e + j];
                    const float r = __fsub_rn(f, __fmul_rn(act[e + j], dl));
                    fm[e + j] = r;
                    if (sg.changed && (f >= 0.0f) != (r >= 0.0f)) sg.changed[(long long)s * sg.mstride + (e + j) / sg.C] = 1;
                }
        }
    }
}

constexpr int SWEEP_IT = AEV_SWEEP_CHUNK / 1024;       // float4s per thread and chunk
constexpr int SWEEP_IT_LA = AEV_SWEEP_CHUNK_LA / 1024; // ... of the look-ahead kernel
// chunk bx (over all segments) of stream s with this delta_leak
__device__ __forceinline__ void leak_sweep_body(const SweepArgs &a, int s, int bx, float dl)
{
    int g = 0;
    while (g + 1 < a.nseg && bx >= a.seg[g + 1].chunk0) ++g;
    // (the segment descriptor is copied out of the kernel arguments: indexed in place, every field is re-fetched -- a scalar
    //  load and a wait -- at every use)
    const SweepSeg sg = a.seg[g];
    const int chunk = bx - sg.chunk0;
    int fl[SWEEP_IT];
    f32x4 dv[SWEEP_IT], fv[SWEEP_IT];
    sweep_flags<SWEEP_IT>(sg, s, chunk, fl);
    sweep_issue<SWEEP_IT>(sg, s, chunk, fl, dv, fv);
    sweep_finish<SWEEP_IT>(sg, s, chunk, dl, fl, dv, fv);
}
__global__ __launch_bounds__(256) void k_leak_sweep(SweepArgs a)
{
    const int s = blockIdx.y + a.s0;
    if (!a.active[s]) return;
    leak_sweep_body(a, s, (int)blockIdx.x, a.dlf[s]);
}
// k_integ_step and the leak sweep of a step's head in ONE launch (the few-streams plan: the sweep was the second launch of every step's
// dependent chain): blocks [0, nib) of a stream are k_integ_step's, the others sweep -- and find delta_leak themselves, from the operands
// k_integ_step uses (the step's last timestamp: a scan of its <= AEV_INTEG_FUSED_MAX events; prev_ts, which nothing in this launch writes),
// so that no block of the launch depends on another.
__global__ __launch_bounds__(256) void k_head(IntegArgs ia, SweepArgs sa, int nib)
{
    if ((int)blockIdx.x < nib) { integ_step_body(ia, (int)blockIdx.x); return; }
    const int s = blockIdx.y + ia.s0;
    const int tid = threadIdx.x;
    const int q = ia.qstep ? *ia.qstep : 0;
    const long long r0 = ia.off[(long long)q * ia.S + s];
    const int n = (int)(ia.off[(long long)q * ia.S + s + 1] - r0);
    if (n <= 0) return; // (block-uniform: a stream without events in the step is not swept)
    __shared__ int red[4];
    int m = INT_MIN;
    for (int i = tid; i < n; i += 256) m = max(m, ia.yxt[(r0 + i) * 3 + 2]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    const long long last = max(max(red[0], red[1]), max(red[2], red[3]));
    const float dl = (float)((double)(last - ia.prev_ts[s]) * ia.leak); // integration.py:60, as k_integ_step
    leak_sweep_body(sa, s, (int)blockIdx.x - nib, dl);
}

// The same sweep for the NEXT queue step (a.dlf / a.active = the next step's delta_leak / activity from k_integ_prepare), run
// beside the MFMA-bound deep conv layers of the current step, once nothing in the current step touches these segments any
// more: the HBM-bound half of the step and the MFMA-bound half overlap.  Few persistent blocks (the grid is a small multiple
// of the CU count): a large grid would take the CUs' wave slots and registers from the tile kernel's one-per-CU blocks, which
// could then not become resident (measured: the co-running conv kernel lost more than the sweep gained); beside conv4 / conv5
// of the e-fcn (414 of a SIMD's 512 registers per lane) there is room for ONE wave of up to 96 registers per SIMD.
// One wave per SIMD has nothing to hide its latencies behind: neither the memory's (~5 us beside the tile kernel) nor its own
// instructions' (a dependent instruction every ~8 cycles).  So the inner loop
//   * runs over the consecutive chunks of ONE segment of ONE stream (everything but the chunk offset is loop-invariant and
//     sits in SGPRs; addresses are a base + a 32-bit offset that advances by a constant): ~80 instructions per chunk;
//   * is a software pipeline with STATIC register slots (two value slots, two flag slots, unrolled twice): while chunk t
//     is finished, the values of chunk t+1 and the flag bytes of chunk t+2 are in flight.  (Rotating the slots with register
//     moves does not work: a move of a register with a load in flight waits for the load.)
template <int IT>
__device__ __forceinline__ void sweep_run_fast(const SweepSeg &sg, int s, int c, int run, float dl)
{
    float *fm = sg.fm + (long long)s * sg.sstride;
    const float *act = sg.act + (long long)s * sg.sstride;
    uint8_t *flag = sg.flag + (long long)s * sg.fstride;
    uint8_t *changed = sg.changed ? sg.changed + (long long)s * sg.mstride : nullptr;
    constexpr unsigned STEP = IT * 1024;
    const unsigned off0 = (unsigned)c * STEP + threadIdx.x * 4; // element offset of this thread's first float4 of chunk c
    const int cs = sg.cshift;
    const int lane = threadIdx.x & 63, gsz = 1 << (cs - 2);
    const u64 grp = (gsz == 64 ? ~0ull : ((1ull << gsz) - 1)) << (lane & ~(gsz - 1));
    int P[2][IT];               // flag bytes
    f32x4 Vd[2][IT], Vf[2][IT]; // decay / feature values
    unsigned Vok[2];            // bit it: the float4 was fetched
    auto ldq = [&](const float *base, unsigned off) __attribute__((always_inline)) {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + (size_t)(off * 4u)));
    };
    auto F = [&](int t, auto ps) __attribute__((always_inline)) { // flag bytes of chunk c + t (a chunk beyond the run: byte 0, ignored)
        constexpr int p = decltype(ps)::value;
        const unsigned o = off0 + (unsigned)t * STEP;
#pragma unroll
        for (int it = 0; it < IT; ++it) P[p][it] = flag[t < run ? (o + it * 1024) >> cs : 0];
    };
    auto I = [&](int t, auto ps, auto vs) __attribute__((always_inline)) { // values of chunk c + t where flagged
        constexpr int p = decltype(ps)::value, v = decltype(vs)::value;
        const unsigned o = off0 + (unsigned)t * STEP;
        unsigned ok = 0;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool on = t < run && P[p][it] != 0;
            const unsigned oa = on ? o + it * 1024 : 0u;
            Vd[v][it] = ldq(act, oa);
            Vf[v][it] = ldq(fm, oa);
            ok |= on ? 1u << it : 0u;
        }
        Vok[v] = ok;
    };
    auto X = [&](int t, auto vs) __attribute__((always_inline)) { // arithmetic, stores, marks of chunk c + t
        constexpr int v = decltype(vs)::value;
        const unsigned o = off0 + (unsigned)t * STEP;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const f32x4 d = Vd[v][it], f = Vf[v][it];
            const bool got = (Vok[v] >> it) & 1;
            const bool z = all_zero(d);
            const u64 bal = __ballot(got && z); // a flagged site whose decay turned out to be all zero: un-flag it
            if (bal != 0 && (bal & grp) == grp && (lane & (gsz - 1)) == 0) flag[(o + it * 1024) >> cs] = 0;
            if (got && !z) {
                f32x4 r;
                const bool maybe = leak4(d, f, dl, r);
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(fm) + (size_t)((o + it * 1024) * 4u)));
                if (maybe && changed && flipped4(f, r)) changed[(o + it * 1024) >> cs] = 1; // (rare; the four channels of a float4 belong to one site: C >= 4)
            }
        }
    };
    constexpr std::integral_constant<int, 0> _0{};
    constexpr std::integral_constant<int, 1> _1{};
    F(0, _0); F(1, _1);
    I(0, _0, _0);
    for (int t = 0; t < run; t += 2) {
        F(t + 2, _0); I(t + 1, _1, _1); X(t, _0);
        F(t + 3, _1); I(t + 2, _0, _0); X(t + 1, _1); // (t + 1 == run: every float4 of the slot is "not fetched")
    }
}

__global__ __launch_bounds__(256) void k_leak_sweep_ahead(SweepArgs a)
{
    constexpr int IT = SWEEP_IT_LA;
    if (!a.prologue && !(*a.qstep + 1 < a.qrun[1])) return; // the next step is not part of this run: its run's prologue sweeps for it
    // A block takes a contiguous range of the work items (stream-major, then segment, then chunk) and walks it in runs of
    // consecutive chunks of one segment of one stream.  (At most two segments per launch: a conv layer and the pool copy
    // behind it -- addressed with constant indices, so that their descriptors stay in SGPRs.)
    const int total = a.chunks * a.ns;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    int w = (int)blockIdx.x * per;
    const int wend = min(w + per, total);
    while (w < wend) {
        const int sl = w / a.chunks, c0 = w - sl * a.chunks;
        const bool second = a.nseg > 1 && c0 >= a.seg[1].chunk0;
        const SweepSeg sg = second ? a.seg[1] : a.seg[0];
        const int c = c0 - sg.chunk0;
        const int nch = (sg.CHW + IT * 1024 - 1) / (IT * 1024), nfull = sg.CHW / (IT * 1024);
        const int run = min(wend - w, nch - c);
        const int s = sl + a.s0;
        if (a.active[s]) {
            const float dl = a.dlf[s];
            // full chunks of a flagged power-of-two segment (C >= 4): the pipelined loop; a partial last chunk, or a segment
            // without flags: the generic path, chunk by chunk
            int fast = 0;
            if (sg.flag != nullptr && sg.cshift >= 2) fast = max(0, min(c + run, nfull) - c);
            if (fast > 0) sweep_run_fast<IT>(sg, s, c, fast, dl);
            for (int k = c + fast; k < c + run; ++k) {
                int fl[IT];
                f32x4 dv[IT], fv[IT];
                sweep_flags<IT>(sg, s, k, fl);
                sweep_issue<IT>(sg, s, k, fl, dv, fv);
                sweep_finish<IT>(sg, s, k, dl, fl, dv, fv);
            }
        }
        w += run;
    }
}

// ------------------------------------------------------------------------------------------------
// Byte map -> ordered lists.  One 1024-thread block per stream, 32 sites per thread and pass.
//   listA  <- sites set in A          (conv: sites to recompute; any order would do, row-major is free)
//   listAB <- sites set in A or B     (events out, row-major = np.where order: conv2d.py:129-131, integration.py:83)
// Clears A and B for the next step.  Event maps are BYTE maps (0/1): they are written by plain same-value
// stores from thousands of threads; bit maps would need contended atomics (measured: 60x slower pool step).
// ------------------------------------------------------------------------------------------------
struct CompactArgs {
    uint8_t *mapA, *mapB; // [S][mstride]; mapB may be nullptr
    long long mstride;     // multiple of 32
    int HW, W, s0;
    int *listA, *nA;   // may be nullptr
    int *listAB, *nAB; // may be nullptr
    long long lstride; // per-stream stride of the lists
    u64 *stat;         // += nA if listA else nAB
};

// w holds 8 bytes (each 0 or 1) of sites p0 .. p0+7
__device__ __forceinline__ int *expand_bytes(u64 w, int p0, int W, int *dst)
{
    if (!w) return dst;
    const int y0 = p0 / W, x0 = p0 - y0 * W; // one integer division per non-empty group of 8 sites (not one per site)
    while (w) {
        const int b = __builtin_ctzll(w) >> 3;
        int y = y0, x = x0 + b;
        while (x >= W) { x -= W; ++y; }
        *dst++ = pack_yx(y, x);
        w &= ~(0xffull << (b * 8));
    }
    return dst;
}

// ... and a copy of the entries below index `cap` into `mir` (LDS), `idx` = list index of dst[0]
__device__ __forceinline__ int *expand_bytes_mirror(u64 w, int p0, int W, int *dst, int *mir, int idx, int cap)
{
    if (!w) return dst;
    const int y0 = p0 / W, x0 = p0 - y0 * W;
    while (w) {
        const int b = __builtin_ctzll(w) >> 3;
        int y = y0, x = x0 + b;
        while (x >= W) { x -= W; ++y; }
        const int pk = pack_yx(y, x);
        *dst++ = pk;
        if (idx < cap) mir[idx] = pk;
        ++idx;
        w &= ~(0xffull << (b * 8));
    }
    return dst;
}

__global__ __launch_bounds__(1024) void k_compact(CompactArgs a)
{
    __shared__ int lds[20];
    const int s = blockIdx.x + a.s0;
    uint8_t *A = a.mapA + (long long)s * a.mstride;
    uint8_t *B = a.mapB ? a.mapB + (long long)s * a.mstride : nullptr;
    int baseA = 0, baseAB = 0;
    for (int p0 = 0; p0 < a.HW; p0 += 1024 * 32) {
        const int p = p0 + threadIdx.x * 32;
        u64 wa[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0};
        if (p < a.HW) { // mstride is a multiple of 32 and the pad bytes stay 0
            const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(A + p);
            const ulonglong2 a0 = pa[0], a1 = pa[1];
            wa[0] = a0.x; wa[1] = a0.y; wa[2] = a1.x; wa[3] = a1.y;
            if (wa[0] | wa[1] | wa[2] | wa[3]) { ulonglong2 z; z.x = 0; z.y = 0; reinterpret_cast<ulonglong2 *>(A + p)[0] = z; reinterpret_cast<ulonglong2 *>(A + p)[1] = z; }
            if (B) {
                const ulonglong2 *pb = reinterpret_cast<const ulonglong2 *>(B + p);
                const ulonglong2 b0 = pb[0], b1 = pb[1];
                wb[0] = b0.x; wb[1] = b0.y; wb[2] = b1.x; wb[3] = b1.y;
                if (wb[0] | wb[1] | wb[2] | wb[3]) { ulonglong2 z; z.x = 0; z.y = 0; reinterpret_cast<ulonglong2 *>(B + p)[0] = z; reinterpret_cast<ulonglong2 *>(B + p)[1] = z; }
            }
        }
        int tot;
        if (a.listA) {
            const int cnt = __popcll(wa[0]) + __popcll(wa[1]) + __popcll(wa[2]) + __popcll(wa[3]);
            const int pos = block_excl_scan_1024(cnt, lds, tot);
            if (cnt) {
                int *dst = a.listA + (long long)s * a.lstride + baseA + pos;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst = expand_bytes(wa[q], p + 8 * q, a.W, dst);
            }
            baseA += tot;
        }
        if (a.listAB) {
            const u64 w0 = wa[0] | wb[0], w1 = wa[1] | wb[1], w2 = wa[2] | wb[2], w3 = wa[3] | wb[3];
            const int cnt = __popcll(w0) + __popcll(w1) + __popcll(w2) + __popcll(w3);
            const int pos = block_excl_scan_1024(cnt, lds, tot);
            if (cnt) {
                int *dst = a.listAB + (long long)s * a.lstride + baseAB + pos;
                dst = expand_bytes(w0, p, a.W, dst); dst = expand_bytes(w1, p + 8, a.W, dst);
                dst = expand_bytes(w2, p + 16, a.W, dst); dst = expand_bytes(w3, p + 24, a.W, dst);
            }
            baseAB += tot;
        }
    }
    if (threadIdx.x == 0) {
        if (a.listA) a.nA[s] = baseA;
        if (a.listAB) a.nAB[s] = baseAB;
        const int st = a.listA ? baseA : baseAB;
        if (a.stat && st) atomicAdd(a.stat, (u64)st);
    }
}

// ------------------------------------------------------------------------------------------------
// Conv: event -> affected output sites (the index half of im2col_event, cutils.pyx:73-112, stride 1)
// Every site whose receptive field holds the event gets its byte set; the map IS the dedup
// (out_covered, cutils.pyx:108-109).
// ------------------------------------------------------------------------------------------------
struct ExpandArgs {
    const int *in_ev; const int *in_n; long long in_stride; // previous layer's event list
    int kh, kw, pt, pl, H, W, s0;                           // this layer's output geometry
    uint8_t *touch; long long mstride;                      // [S][mstride] byte map
};

__global__ __launch_bounds__(256) void k_conv_expand(ExpandArgs a)
{
    const int s = blockIdx.y + a.s0;
    const int kk = a.kh * a.kw;
    const int total = a.in_n[s] * kk;
    const int *ev = a.in_ev + (long long)s * a.in_stride;
    uint8_t *T = a.touch + (long long)s * a.mstride;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e / kk, r = e - i * kk;
        const int dy = r / a.kw, dx = r - dy * a.kw;
        const int pk = ev[i];
        const int oy = unpack_y(pk) + a.pt - dy, ox = unpack_x(pk) + a.pl - dx;
        if (oy >= 0 && oy < a.H && ox >= 0 && ox < a.W) T[oy * a.W + ox] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Conv: exact recompute of the touched sites (conv2d.py:118-123 = conv2d_event :144-181 twice:
// gather of cutils.pyx:116-127 + one GEMM each for the feature map (with bias) and the decay map).
// Arithmetic contract: out = fmaf-chain over k ascending from +0, then + bias (DESIGN.md "Numerics").
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    SrcView src;
    float *fm, *act;       // [S][sstride]
    long long sstride;
    const float *wt;       // [Krows][CoutP] transposed + zero padded weights
    const float *bias;     // [CoutP]
    int C, CoutP, H, W, HW, K, kh, kw, pt, pl;
    int KC;                // K-chunk staged in LDS per pass (mfma kernel)
    int TS;                // sites per tile (valu kernel)
    const int *sites; const int *nsites; long long lstride;
    int s0;
};

// Generic VALU kernel (any Cout / K; used for conv1 where K = 9 and for odd shapes).
// Block = 256 threads, tile = TS sites; the gathered [K][TS] feature / decay columns sit in LDS.
__global__ __launch_bounds__(256) void k_conv_valu(ConvArgs a)
{
    extern __shared__ float lds[];
    const int s = blockIdx.y + a.s0;
    const int n = a.nsites[s];
    const int TS = a.TS, K = a.K, KK = a.kh * a.kw;
    float *xf = lds, *xd = lds + K * TS;
    int *sp = reinterpret_cast<int *>(lds + 2 * K * TS); // [TS] packed site coords
    const int *sites = a.sites + (long long)s * a.lstride;
    float *fm = a.fm + (long long)s * a.sstride, *act = a.act + (long long)s * a.sstride;
    const int ih = a.src.H, iw = a.src.W;
    for (int tile = blockIdx.x; tile * TS < n; tile += gridDim.x) {
        __syncthreads();
        if ((int)threadIdx.x < TS) sp[threadIdx.x] = tile * TS + threadIdx.x < n ? sites[tile * TS + threadIdx.x] : -1;
        __syncthreads();
        for (int e = threadIdx.x; e < K * TS; e += 256) {
            const int k = e / TS, j = e - k * TS;
            float F = 0.0f, D = 0.0f;
            const int pk = sp[j];
            if (pk >= 0) {
                int c, r;
                chain_decode(k, a.src.C, KK, c, r); // k = position in the chain
                const int dy = r / a.kw, dx = r - dy * a.kw;
                const int iy = unpack_y(pk) + dy - a.pt, ix = unpack_x(pk) + dx - a.pl;
                if (iy >= 0 && iy < ih && ix >= 0 && ix < iw) { // zero padding, conv2d.py:68-72
                    float sv, L;
                    src_fetch(a.src, s, c, iy, ix, sv, L, D);
                    F = __fmul_rn(sv, L); // layer.py:80
                }
            }
            xf[e] = F;
            xd[e] = D;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < a.C * TS; o += 256) {
            const int j = o / a.C, co = o - j * a.C;
            const int pk = sp[j];
            if (pk < 0) continue;
            float af = 0.0f, ad = 0.0f;
            const float *w = a.wt + co;
            for (int k = 0; k < K; ++k) {
                const float wk = w[(long long)k * a.CoutP];
                af = __fmaf_rn(wk, xf[k * TS + j], af);
                ad = __fmaf_rn(wk, xd[k * TS + j], ad);
            }
            const long long q = ((long long)unpack_y(pk) * a.W + unpack_x(pk)) * a.C + co;
            fm[q] = __fadd_rn(af, a.bias[co]); // conv2d.py:176-178,120
            act[q] = ad;                        // conv2d.py:121-123 (no bias)
        }
    }
}

// First conv of a net: previous layer is the integration surface (Cin = 1), 3x3 kernel (K = 9) -- conv1 of the
// e-fcn.  Pure gather/scatter work (288 fma per site against 36 B gathered + 128 B scattered): one thread per
// site, the 9 inputs in registers, weights in LDS (broadcast reads), stores coalesced over neighbouring sites.
__global__ __launch_bounds__(256) void k_conv_first3x3(ConvArgs a)
{
    extern __shared__ float lds[];
    const int s = blockIdx.y + a.s0;
    const int n = a.nsites[s];
    if ((int)blockIdx.x * 256 >= n * ((a.C & 3) == 0 ? a.C >> 2 : 1)) return;
    float *w = lds, *b = lds + 9 * a.C; // w[k][co]
    for (int i = threadIdx.x; i < 9 * a.C; i += 256) w[i] = a.wt[(long long)(i / a.C) * a.CoutP + i % a.C];
    for (int i = threadIdx.x; i < a.C; i += 256) b[i] = a.bias[i];
    __syncthreads();
    const int *sites = a.sites + (long long)s * a.lstride;
    const float *S = a.src.s + (long long)s * a.src.bstride;
    float *fm = a.fm + (long long)s * a.sstride, *act = a.act + (long long)s * a.sstride;
    const int ih = a.src.H, iw = a.src.W;
    // C % 4 == 0: one thread per (site, four channels) -- the C/4 lanes of a site sit side by side, so a store instruction
    // writes each site's C*4 bytes as one contiguous run (with a thread per site, every 16-byte store of a wave went to 64
    // different lines: PMC counted three times the bytes the kernel has to write)
    const bool quad = (a.C & 3) == 0;
    const int G = quad ? a.C >> 2 : 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * G; e += gridDim.x * 256) {
        const int j = e / G, co0 = (e - j * G) * 4;
        const int pk = sites[j];
        const int oy = unpack_y(pk), ox = unpack_x(pk);
        float xf[9], xd[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int iy = oy + r / 3 - a.pt, ix = ox + r % 3 - a.pl;
            const bool in = iy >= 0 && iy < ih && ix >= 0 && ix < iw;
            const float v = S[min(max(iy, 0), ih - 1) * iw + min(max(ix, 0), iw - 1)]; // unconditional load, zero padding below
            const float L = v > 0.0f ? 1.0f : 0.0f;                                      // integration.py:30-43
            xf[r] = in ? __fmul_rn(v, L) : 0.0f;
            xd[r] = in ? L : 0.0f;
        }
        float *fq = fm + ((long long)oy * a.W + ox) * a.C, *aq = act + ((long long)oy * a.W + ox) * a.C;
        if (quad) { // a site's channels are contiguous: 16-byte stores
            f32x4 vf, vd;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float af = 0.0f, ad = 0.0f;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const float wk = w[r * a.C + co0 + u];
                    af = __fmaf_rn(wk, xf[r], af);
                    ad = __fmaf_rn(wk, xd[r], ad);
                }
                vf[u] = __fadd_rn(af, b[co0 + u]);
                vd[u] = ad;
            }
            *reinterpret_cast<f32x4 *>(fq + co0) = vf;
            *reinterpret_cast<f32x4 *>(aq + co0) = vd;
        } else {
            for (int co = 0; co < a.C; ++co) {
                float af = 0.0f, ad = 0.0f;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const float wk = w[r * a.C + co];
                    af = __fmaf_rn(wk, xf[r], af);
                    ad = __fmaf_rn(wk, xd[r], ad);
                }
                fq[co] = __fadd_rn(af, b[co]);
                aq[co] = ad;
            }
        }
    }
}

// Scatter one MFMA accumulator pair (feature, decay) of a 32-cout block for this lane's site.  C/D layout: register r
// holds cout (r&3) + 8*(r>>2) + 4*(lane>>5): four runs of 4 consecutive couts per lane, and the two half-waves
// interleave, so lanes l and l+32 together write the site's 128 contiguous bytes.
__device__ __forceinline__ void store_site_block(float *fq, float *aq, const f32x16 &af, const f32x16 &ad, const float *bias,
                                                 int co0, int C, int lane)
{
    const int cb = co0 + 4 * (lane >> 5);
    if ((C & 3) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = cb + 8 * g;
            if (co < C) {
                f32x4 vf, vd;
#pragma unroll
                for (int u = 0; u < 4; ++u) { vf[u] = __fadd_rn(af[4 * g + u], bias[co + u]); vd[u] = ad[4 * g + u]; }
                *reinterpret_cast<f32x4 *>(fq + co) = vf;
                *reinterpret_cast<f32x4 *>(aq + co) = vd;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb + (r & 3) + 8 * (r >> 2);
            if (co < C) { fq[co] = __fadd_rn(af[r], bias[co]); aq[co] = ad[r]; }
        }
    }
}

// Scatter for the kernels that run the MFMA with the SITES on the rows (A = inputs, B = weights): this lane owns
// cout `co`, register r holds site row (r&3) + 8*(r>>2) + 4*(lane>>5) of the 32-site block.  One store instruction
// then writes 32 consecutive couts of one site from lanes 0-31 and of another site from lanes 32-63: two whole 128-B
// segments of the channel-fastest state (instead of 32-B pieces).
template <typename RowToOffset>
__device__ __forceinline__ void store_rows(float *fm, float *act, const f32x16 &af, const f32x16 &ad, float bias, int co, int C,
                                           int lane, int nrows_valid, RowToOffset row_off)
{
    if (co >= C) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < nrows_valid) {
            const long long q = row_off(row);
            fm[q + co] = __fadd_rn(af[r], bias);
            act[q + co] = ad[r];
        }
    }
}

// f32 MFMA kernel (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered fmaf chain, so results are
// identical to k_conv_valu).  D[cout][site] = sum_k Wt[k][cout] * X[k][site]:
//   A operand = weights (lane l: A[i = l&31][k = l>>5]) straight from global/L2 (coalesced 2 x 128 B),
//   B operand = gathered columns from LDS (lane l: B[k = l>>5][j = l&31]),
//   C/D: col = lane&31 (site), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout).
// Block = 4 waves = NSW site sub-tiles (32 sites each) x NCW cout groups; each wave owns CBW cout
// blocks of 32 and both right-hand sides (feature, decay) -> 2*CBW accumulators of 16 registers.
template <int NSW, int NCW, int CBW>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs a)
{
    extern __shared__ float lds[];
    constexpr int TSB = 32 * NSW;
    const int s = blockIdx.y + a.s0;
    const int n = a.nsites[s];
    const int K = a.K, KK = a.kh * a.kw, KC = a.KC;
    float *xf = lds, *xd = lds + KC * TSB;
    int *sp = reinterpret_cast<int *>(lds + 2 * KC * TSB); // [TSB]
    int *ktab = sp + TSB;                                   // [KC] packed (c << 16 | dy << 8 | dx)
    const int *sites = a.sites + (long long)s * a.lstride;
    float *fm = a.fm + (long long)s * a.sstride, *act = a.act + (long long)s * a.sstride;
    const int ih = a.src.H, iw = a.src.W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sw = wv % NSW, cw = wv / NSW;
    const int cout_base = (blockIdx.z * NCW + cw) * CBW * 32;
    const int jl = threadIdx.x % TSB, kq = threadIdx.x / TSB; // gather role: one site column, every (256/TSB)-th k
    constexpr int KSTEP = 256 / TSB;

    for (int tile = blockIdx.x; tile * TSB < n; tile += gridDim.x) {
        __syncthreads();
        if ((int)threadIdx.x < TSB) sp[threadIdx.x] = tile * TSB + threadIdx.x < n ? sites[tile * TSB + threadIdx.x] : -1;
        f32x16 accf[CBW], accd[CBW];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accf[cb][r] = 0.0f; accd[cb][r] = 0.0f; }

        for (int k0 = 0; k0 < K; k0 += KC) {
            __syncthreads(); // previous chunk fully consumed; sp visible
            if ((int)threadIdx.x < KC) {
                const int k = k0 + threadIdx.x; // position in the chain
                int c = 0, r = 0;
                if (k < K) chain_decode(k, a.src.C, KK, c, r);
                const int dy = r / a.kw, dx = r - dy * a.kw;
                ktab[threadIdx.x] = k < K ? (c << 16) | (dy << 8) | dx : -1;
            }
            __syncthreads();
            {
                const int pk = sp[jl];
                const int sy = unpack_y(pk) - a.pt, sx = unpack_x(pk) - a.pl;
                for (int kk = kq; kk < KC; kk += KSTEP) {
                    float F = 0.0f, D = 0.0f;
                    const int kt = ktab[kk];
                    if (pk >= 0 && kt >= 0) {
                        const int iy = sy + ((kt >> 8) & 0xff), ix = sx + (kt & 0xff);
                        if (iy >= 0 && iy < ih && ix >= 0 && ix < iw) {
                            float sv, L;
                            src_fetch(a.src, s, kt >> 16, iy, ix, sv, L, D);
                            F = __fmul_rn(sv, L);
                        }
                    }
                    xf[kk * TSB + jl] = F;
                    xd[kk * TSB + jl] = D;
                }
            }
            __syncthreads();
            const float *wrow = a.wt + (long long)(k0 + (lane >> 5)) * a.CoutP + cout_base + (lane & 31);
            const float *bfp = xf + (lane >> 5) * TSB + sw * 32 + (lane & 31);
            const float *bdp = xd + (lane >> 5) * TSB + sw * 32 + (lane & 31);
#pragma unroll 4
            for (int kk = 0; kk < KC; kk += 2) {
                const float bf = bfp[kk * TSB], bd = bdp[kk * TSB];
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) {
                    const float aw = wrow[(long long)kk * a.CoutP + cb * 32];
                    accf[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bf, accf[cb], 0, 0, 0);
                    accd[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bd, accd[cb], 0, 0, 0);
                }
            }
        }
        const int pk = sp[sw * 32 + (lane & 31)];
        if (pk >= 0) {
            const long long q0 = ((long long)unpack_y(pk) * a.W + unpack_x(pk)) * a.C;
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
                store_site_block(fm + q0, act + q0, accf[cb], accd[cb], a.bias, cout_base + cb * 32, a.C, lane);
        }
    }
}

// Scatter of one 32-site x 32-cout accumulator pair with the sites' element offsets in LDS (off[32], 16-byte aligned):
// four 16-byte LDS reads give a lane its 16 row offsets; a full site block (the common case) stores without per-row tests.
// fm / act are block-uniform bases: every store is base + unsigned 32-bit offset.
__device__ __forceinline__ void store_site_offsets(float *fm, float *act, const f32x16 &af, const f32x16 &ad, float bias, int co, int C,
                                                   int lane, int nrows_valid, const int *off)
{
    if (co >= C) return;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int hb = 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const i32x4 o4 = *reinterpret_cast<const i32x4 *>(off + 8 * g + hb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            if (nrows_valid >= 32 || 8 * g + hb + j < nrows_valid) {
                const unsigned q = (unsigned)(o4[j] + co);
                st_u32(fm, q, __fadd_rn(af[r], bias));
                st_u32(act, q, ad[r]);
            }
        }
    }
}

// Active tiles of a tile-kernel layer (fallback of the listing in k_post: per-layer API, dense mode, state init): one block
// per stream scans the stream's touch map 8 bytes at a time and ORs every touched site into its tile's 256-bit site mask
// (bit = ly * TW + lx) in LDS; non-empty tiles are appended to the layer's list (stream-local index * tiles + tile) together
// with their masks.  The order of the list depends on the atomics; the results do not (tiles are independent).
// The list of a launch's active tiles is kept in AEV_TL_BUCKETS segments of cap entries, by the number of active sites of the
// tile (segment 7: the fullest tiles), and the tile kernel walks it from the fullest tiles down -- largest items first.  The
// kernel's blocks take a few items each (conv4 of the e-fcn at 256 streams: four items of ~80 us), and with the tiles in the
// order the streams' k_post blocks happen to append them the slowest block of a launch ran 13 % behind the mean (conv4; conv3
// 17 %): with the items sorted by size, strided rounds that alternate direction hand every block large and small ones, and the
// work queue's last draws are the smallest.
#ifndef AEV_TL_BUCKETS
#define AEV_TL_BUCKETS 32 // (at most 32: the tile kernel keeps its map of the segments in the lanes of one register)
#endif
#define AEV_TC_STRIDE 64 // ints per layer in the step's scratch counters: [0] tiles listed, [1] work queue, [4 .. 4 + AEV_TL_BUCKETS) tiles per segment
struct TileListArgs {
    const uint8_t *touch; long long mstride;
    int H, W, TH, TW, tilesX, tiles;
    int s0;
    int *list; u64 *mask; int *count; // count[0 .. AEV_TC_STRIDE) zeroed before the launch
    int cap;                          // entries per segment
    int nb;                           // segments in use (1: the list in the order of the appends; at most AEV_TL_BUCKETS)
};
__device__ __forceinline__ void tile_mask_set(u64 *tile_mask, int y, int x, int TH, int TW, int tilesX)
{
    const int ty = y / TH, tx = x / TW;
    const int q = (y - ty * TH) * TW + (x - tx * TW);
    atomicOr(reinterpret_cast<unsigned long long *>(&tile_mask[(ty * tilesX + tx) * 4 + (q >> 6)]), 1ull << (q & 63));
}
// threads t < tiles append their tile if its mask is not empty (block-wide; NT threads)
template <int NT>
__device__ __forceinline__ void tile_list_append(const u64 *tile_mask, int tiles, int stream_local, int *list, u64 *mask, int *count, int cap, int tile_sites, int nb)
{
    const int ln = threadIdx.x & 63;
    for (int t0 = 0; t0 < tiles; t0 += NT) {
        const int t = t0 + threadIdx.x;
        u64 m[4] = {0, 0, 0, 0};
        if (t < tiles) { m[0] = tile_mask[t * 4]; m[1] = tile_mask[t * 4 + 1]; m[2] = tile_mask[t * 4 + 2]; m[3] = tile_mask[t * 4 + 3]; }
        const int nact = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
        const bool on = nact != 0;
        const int bk = min(nb - 1, max(nact - 1, 0) * nb / tile_sites); // segment by fill
        const u64 bal = __ballot(on);
        if (!bal) continue; // (wave-uniform)
        // ONE atomic instruction per wave: lane b < nb reserves segment b's entries of this wave, lane 32 counts the launch's tiles
        u64 mine = bal; // the lanes of this lane's segment
        int cnt = ln == 32 || nb == 1 ? __popcll(bal) : 0; // lane b: entries for segment b; lane 32: all
        if (nb > 1) {
#pragma unroll 1
            for (int b = 0; b < nb; ++b) {
                const u64 bb = __ballot(on && bk == b);
                if (bk == b) mine = bb;
                if (ln == b) cnt = __popcll(bb);
            }
        }
        int base = 0;
        if ((ln < nb || ln == 32) && cnt) base = atomicAdd(count + (ln < nb ? 4 + ln : 0), cnt);
        base = __shfl(base, bk);
        if (on) {
            const int pos = bk * cap + base + __popcll(mine & ((1ull << ln) - 1));
            list[pos] = stream_local * tiles + t;
            mask[(long long)pos * 4] = m[0]; mask[(long long)pos * 4 + 1] = m[1]; mask[(long long)pos * 4 + 2] = m[2]; mask[(long long)pos * 4 + 3] = m[3];
        }
    }
}
__global__ __launch_bounds__(256) void k_tile_list(TileListArgs a)
{
    extern __shared__ u64 tl_mask[]; // [tiles][4]
    const int s = blockIdx.x;
    const uint8_t *T = a.touch + (long long)(s + a.s0) * a.mstride;
    for (int t = threadIdx.x; t < a.tiles * 4; t += 256) tl_mask[t] = 0;
    __syncthreads();
    const int HW = a.H * a.W;
    for (int p = threadIdx.x * 8; p < HW; p += 256 * 8) { // mstride is a multiple of 32
        u64 w = *reinterpret_cast<const u64 *>(T + p);
        while (w) {
            const int b = __ffsll((long long)w) - 1;
            w &= ~(0xffull << (b & ~7));
            const int q = p + (b >> 3), y = q / a.W, x = q - y * a.W;
            if (q < HW) tile_mask_set(tl_mask, y, x, a.TH, a.TW, a.tilesX);
        }
    }
    __syncthreads();
    tile_list_append<256>(tl_mask, a.tiles, s, a.list, a.mask, a.count, a.cap, a.TH * a.TW, a.nb);
}

// ------------------------------------------------------------------------------------------------
// First-touch ordered site lists (cutils.pyx:73-112).  Used by the pool layer (maxpool.py:123-131) and
// by the stand-alone im2col_event op.  Entry e = (event i, enumeration slot r): key = e.  The site's
// owner is its smallest key; listing the owners in key order IS first-touch order.
// ------------------------------------------------------------------------------------------------
struct TouchGeom {
    int mode;            // 0: stride 1 (k x k sites per event); 1: stride == k (one window per event)
    int kh, kw, oh, ow;  // kernel, output (site) map
    int ih, iw;          // input map (stride-1 clipping)
};

__device__ __forceinline__ int touch_site(const TouchGeom &g, int pk, int r)
{
    const int y = unpack_y(pk), x = unpack_x(pk);
    if (g.mode == 1) return (y / g.kh) * g.ow + x / g.kw; // cutils.pyx:83-87
    // stride 1: slot r = (ty - (y-kh+1)) * kw + (tx - (x-kw+1)) over top-left corners, cutils.pyx:78-82,92-101
    const int ry = r / g.kw, rx = r - ry * g.kw;
    const int ty = y - (g.kh - 1) + ry, tx = x - (g.kw - 1) + rx;
    if (ty < 0 || ty > g.ih - g.kh || tx < 0 || tx > g.iw - g.kw) return -1;
    return ty * g.ow + tx;
}

struct MarkArgs {
    TouchGeom g;
    const int *in_ev; const int *in_n; long long in_stride;
    int *key;            // [S][oh*ow], INT_MAX when idle
    uint8_t *pending;    // pool: [S][mstride] bytes cleared for incoming windows (maxpool.py:118-120); may be nullptr
    long long mstride;
    int s0;
};

__global__ __launch_bounds__(256) void k_touch_mark(MarkArgs a)
{
    const int s = blockIdx.y + a.s0;
    const int slots = a.g.mode == 1 ? 1 : a.g.kh * a.g.kw;
    const int total = a.in_n[s] * slots;
    const int *ev = a.in_ev + (long long)s * a.in_stride;
    int *key = a.key + (long long)s * a.g.oh * a.g.ow;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e / slots, r = e - i * slots;
        const int site = touch_site(a.g, ev[i], r);
        if (site < 0) continue;
        atomicMin(&key[site], e);
        if (a.pending) a.pending[(long long)s * a.mstride + site] = 0;
    }
}

struct ListArgs {
    TouchGeom g;
    const int *in_ev; const int *in_n; long long in_stride;
    int *key;
    const uint8_t *pending; // pool: windows still pending are appended in row-major order (maxpool.py:122-126)
    long long mstride;
    int s0;
    int *list; int *nlist; long long lstride;
    u64 *stat;
    const int *active; // nullptr: all streams active
};

__global__ __launch_bounds__(1024) void k_touch_list(ListArgs a)
{
    __shared__ int lds[20];
    const int s = blockIdx.x + a.s0;
    if (a.active && !a.active[s]) { if (threadIdx.x == 0) a.nlist[s] = 0; return; }
    const int slots = a.g.mode == 1 ? 1 : a.g.kh * a.g.kw;
    const int total = a.in_n[s] * slots;
    const int *ev = a.in_ev + (long long)s * a.in_stride;
    int *key = a.key + (long long)s * a.g.oh * a.g.ow;
    int *list = a.list + (long long)s * a.lstride;
    int base = 0;
    for (int e0 = 0; e0 < total; e0 += 1024) {
        const int e = e0 + threadIdx.x;
        int site = -1;
        if (e < total) {
            const int i = e / slots;
            site = touch_site(a.g, ev[i], e - i * slots);
            if (site >= 0 && key[site] != e) site = -1;
        }
        int tot;
        const int pos = block_excl_scan_1024(site >= 0 ? 1 : 0, lds, tot);
        if (site >= 0) {
            const int y = site / a.g.ow;
            list[base + pos] = pack_yx(y, site - y * a.g.ow);
            key[site] = INT_MAX; // leave the table idle for the next step
        }
        base += tot;
    }
    if (a.pending) {
        const uint8_t *P = a.pending + (long long)s * a.mstride;
        const int HW = a.g.oh * a.g.ow;
        for (int p0 = 0; p0 < HW; p0 += 1024 * 32) {
            const int p = p0 + threadIdx.x * 32;
            u64 w[4] = {0, 0, 0, 0};
            if (p < HW) {
                const ulonglong2 *pp = reinterpret_cast<const ulonglong2 *>(P + p);
                const ulonglong2 a0 = pp[0], a1 = pp[1];
                w[0] = a0.x; w[1] = a0.y; w[2] = a1.x; w[3] = a1.y;
            }
            int tot;
            const int cnt = __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
            const int pos = block_excl_scan_1024(cnt, lds, tot);
            if (cnt) {
                int *dst = list + base + pos;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst = expand_bytes(w[q], p + 8 * q, a.g.ow, dst);
            }
            base += tot;
        }
    }
    if (threadIdx.x == 0) {
        a.nlist[s] = base;
        if (a.stat && base) atomicAdd(a.stat, (u64)base);
    }
}

// (the pool layer's argument block and window arithmetic sit here because k_post may evaluate a small pool layer itself)
struct PoolArgs {
    SrcView src;           // previous layer (surface = max_arg, conv_actfn = min_arg)
    int C, H, W, kh, kw, stride, s0;
    uint8_t *idx; long long istride;  // [S][C*H*W]
    uint8_t *pending; long long mstride; // [S][mstride] byte map
    const int *list; const int *nlist; long long lstride;
    int plain;             // 1: first-occurrence argmax only (state init, maxpool.py:33), never flags pending
    // The pool layer's copy of its inputs at the argmax sites (same layout as idx, nullptr: none): surface and raw decay of
    // the winner.  The tile kernels read it as a plain conv source instead of chasing the argmax bytes (see aev_hip.hip).
    float *pfm, *pact;
    uint8_t *pflag;        // [S][mstride] "may decay" flags of the copy (SweepSeg::flag): one per window
    // [S] number of list entries that are windows of incoming events (the rest are pending windows, appended); nullptr:
    // unknown.  A pending-only window whose argmax does not change needs no rewrite of the copy: nothing but the leak
    // has touched its inputs since its last evaluation, and the copy leaked with them.
    const int *nevw;
};

// one window x channel: running argmax with the decay tie-break + slowest-decayer test
struct PoolPick {
    float bs, bd, slow, ba; int best; // ba: raw decay (_conv_actfn) of the winner, for the pool layer's copy of its inputs
    __device__ __forceinline__ void first(float sv, float D, float A = 0.0f) { bs = sv; bd = D; slow = D; ba = A; best = 0; }
    __device__ __forceinline__ void next(int r, float sv, float D, float A = 0.0f)
    {
        if (sv > bs) { best = r; bs = sv; bd = D; ba = A; }              // cutils.pyx:164-165
        else if (sv == bs && D < bd) { best = r; bd = D; ba = A; }       // cutils.pyx:166-168
        if (D < slow) slow = D;                                           // cutils.pyx:171-172
    }
};

// 2x2 / stride-2 windows over a conv layer with C % 4 == 0: one work item = one window x 4 consecutive channels, 16-byte loads of the
// four positions' channel runs, one 4-byte argmax store.  Items first, first + stride, ... of stream s's list of n windows, of which
// the first nin are windows of incoming events (nin < 0: unknown); list_l: an LDS copy of the first cap_l list entries (or cap_l = 0).
__device__ __forceinline__ void pool_eval_vec4_items(const PoolArgs &a, int s, int n, int nin, const int *list, const int *list_l, int cap_l,
                                                     int first, int stride)
{
    const int C4 = a.C >> 2;
    const int total = n * C4;
    const float al = a.src.alpha;
    const bool pl = a.plain != 0;
    const long long orow = (long long)a.src.bW * a.C;
    for (int e = first; e < total; e += stride) {
        const int j = e / C4, c = (e - j * C4) << 2;
        const int pk = j < cap_l ? list_l[j] : list[j];
        const int wy = unpack_y(pk), wx = unpack_x(pk);
        const long long o = (long long)s * a.src.bstride + ((long long)(2 * wy) * a.src.bW + 2 * wx) * a.C + c;
        f32x4 sv[4], av[4];
        sv[0] = *reinterpret_cast<const f32x4 *>(a.src.s + o);        av[0] = *reinterpret_cast<const f32x4 *>(a.src.a + o);
        sv[1] = *reinterpret_cast<const f32x4 *>(a.src.s + o + a.C);  av[1] = *reinterpret_cast<const f32x4 *>(a.src.a + o + a.C);
        sv[2] = *reinterpret_cast<const f32x4 *>(a.src.s + o + orow); av[2] = *reinterpret_cast<const f32x4 *>(a.src.a + o + orow);
        sv[3] = *reinterpret_cast<const f32x4 *>(a.src.s + o + orow + a.C);
        av[3] = *reinterpret_cast<const f32x4 *>(a.src.a + o + orow + a.C);
        unsigned packed = 0;
        bool flag = false;
        f32x4 wf, wa;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            PoolPick pp;
            pp.first(sv[0][u], pl ? 0.0f : __fmul_rn(av[0][u], sv[0][u] > 0.0f ? 1.0f : al), av[0][u]);
#pragma unroll
            for (int r = 1; r < 4; ++r) pp.next(r, sv[r][u], pl ? 0.0f : __fmul_rn(av[r][u], sv[r][u] > 0.0f ? 1.0f : al), av[r][u]);
            packed |= (unsigned)pp.best << (8 * u);
            flag |= pp.bd != pp.slow;
            wf[u] = pp.bs; wa[u] = pp.ba;
        }
        const int wsite = wy * a.W + wx;
        const long long po = (long long)s * a.istride + (long long)wsite * a.C + c;
        if (a.pfm && nin >= 0 && j >= nin && *reinterpret_cast<const unsigned *>(a.idx + po) == packed) { // pending-only, same winners
            if (!pl && flag) a.pending[(long long)s * a.mstride + wsite] = 1;
            continue;
        }
        *reinterpret_cast<unsigned *>(a.idx + po) = packed; // maxpool.py:145-151
        if (a.pfm) {
            *reinterpret_cast<f32x4 *>(a.pfm + po) = wf; *reinterpret_cast<f32x4 *>(a.pact + po) = wa; // (the copy has idx's layout)
            if (a.pflag && c == 0) a.pflag[(long long)s * a.mstride + wsite] = 1;
        }
        if (!pl && flag) a.pending[(long long)s * a.mstride + wy * a.W + wx] = 1;                                      // maxpool.py:140-142
    }
}

// ------------------------------------------------------------------------------------------------
// k_post: everything that is "one block per stream" between two heavy kernels, in ONE launch (23 launches per e-fcn
// step instead of 41):
//   A. this layer's byte maps -> row-major event list (conv2d.py:129-131 / integration.py:83), count of recomputed
//      sites, maps cleared for the next step;
//   B. if a pool follows: incoming windows un-flagged, first-touch ordered window list, pending windows appended
//      (maxpool.py:116-131, cutils.pyx:73-112);
//   C. if a conv consumes that list (or this layer's own events): its touch map is marked (cutils.pyx:73-112) and,
//      for the list-driven conv kernels, compacted into its site list.
// All stages run in one 1024-thread block per stream, ordered by block barriers.
// ------------------------------------------------------------------------------------------------
struct PostArgs {
    int s0;
    const int *active; // [S] streams without events in this step are left untouched (nullptr: all active)
    // A
    uint8_t *mapA, *mapB; long long mstride; int HW, W;
    int *ev, *nev; long long lstride; u64 *statA; int count_touch; // statA += (count_touch ? #touch : #events)
    uint8_t *dflag;     // conv: the layer's "may decay" flags (SweepSeg::flag): set for the recomputed sites = mapA; else nullptr
    // B
    int has_pool; TouchGeom pg; int *pkey; uint8_t *ppending; long long pmstride; int *plist, *pnlist, *pnevw; long long plstride;
    u64 *statP;
    // C
    int has_next; int n_kh, n_kw, n_pt, n_pl, n_H, n_W; uint8_t *n_touch; long long n_mstride;
    int *n_sites, *n_nsites; long long n_lstride; // non-null: also compact the next conv's touch map into its site list
    int *n_tlist, *n_tcount; u64 *n_tmask; int n_TH, n_TW, n_tilesX, n_tiles, n_cap, n_nb; // non-null: also list the next conv's active tiles (tile kernel)
    // end of a queued step (last layer's k_post only; else nullptr): the device step counter moves on (nothing after this
    // launch reads it; the next step's first kernel does)
    int *advance; u64 *advance_stat;
    // the input layer behind k_integ_step: prev_ts <- last_ts of the streams with events (integration.py:86); else nullptr
    long long *prev_ts; const long long *last_ts;
    // LDS (dynamic, after the tile masks; sizes in ints, 0 = not kept): the pool's first-touch key map (instead of the global one:
    // the block's atomicMin round trips were a third of the kernel), copies of the first l_ev events and l_pl windows (the
    // later stages re-read the lists the earlier ones wrote)
    int l_mask, l_key, l_pl, l_ev; // (l_mask: u64 words of the tile masks)
    // D (has_pool, a small pool layer): the block evaluates the windows it has just listed itself (k_pool_eval_vec4's work: a few
    // items per thread) -- one launch and one launch gap less on the chain's critical path
    int pe_inline; PoolArgs pe;
};

// ordered expansion of a 0/1 byte map into packed sites; returns the total (all threads)
__device__ __forceinline__ int dev_map_to_list(const uint8_t *M0, const uint8_t *M1, int HW, int W, int *list, int *lds, bool clear,
                                               uint8_t *C0, uint8_t *C1, int *count0, uint8_t *F0 = nullptr, int *mir = nullptr, int mir_cap = 0)
{
    int base = 0, c0 = 0;
    for (int p0 = 0; p0 < HW; p0 += 1024 * 32) {
        const int p = p0 + threadIdx.x * 32;
        u64 w[4] = {0, 0, 0, 0};
        if (p < HW) {
            // (all loads of the pass issued together: one round trip)
            const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(M0 + p);
            const ulonglong2 a0 = pa[0], a1 = pa[1];
            ulonglong2 *pf = reinterpret_cast<ulonglong2 *>(F0 + p);
            ulonglong2 f0, f1, b0, b1;
            f0.x = f0.y = f1.x = f1.y = b0.x = b0.y = b1.x = b1.y = 0;
            if (F0) { f0 = pf[0]; f1 = pf[1]; }
            if (M1) { const ulonglong2 *pb = reinterpret_cast<const ulonglong2 *>(M1 + p); b0 = pb[0]; b1 = pb[1]; }
            w[0] = a0.x; w[1] = a0.y; w[2] = a1.x; w[3] = a1.y;
            c0 += __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
            if (F0 && (w[0] | w[1] | w[2] | w[3])) { // the sites of M0 (0/1 bytes) get their "may decay" flag (SweepSeg::flag)
                f0.x |= w[0]; f0.y |= w[1]; f1.x |= w[2]; f1.y |= w[3];
                pf[0] = f0; pf[1] = f1;
            }
            if (clear && (w[0] | w[1] | w[2] | w[3])) { ulonglong2 z; z.x = 0; z.y = 0; reinterpret_cast<ulonglong2 *>(C0 + p)[0] = z; reinterpret_cast<ulonglong2 *>(C0 + p)[1] = z; }
            if (M1) {
                if (clear && (b0.x | b0.y | b1.x | b1.y)) { ulonglong2 z; z.x = 0; z.y = 0; reinterpret_cast<ulonglong2 *>(C1 + p)[0] = z; reinterpret_cast<ulonglong2 *>(C1 + p)[1] = z; }
                w[0] |= b0.x; w[1] |= b0.y; w[2] |= b1.x; w[3] |= b1.y;
            }
        }
        int tot;
        const int cnt = __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
        const int pos = block_excl_scan_1024(cnt, lds, tot);
        if (cnt) {
            int *dst = list + base + pos;
            if (mir_cap > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst = expand_bytes_mirror(w[q], p + 8 * q, W, dst, mir, (int)(dst - list), mir_cap);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst = expand_bytes(w[q], p + 8 * q, W, dst);
            }
        }
        base += tot;
    }
    if (count0) { // block-wide sum of the bits of M0
        int tot;
        block_excl_scan_1024(c0, lds, tot);
        *count0 = tot;
    }
    return base;
}

// (tile_mask: the block's dynamic LDS -- [n_tiles][4] site masks when the next conv's tiles are listed here; then the LDS tables)
__device__ __forceinline__ void post_body(const PostArgs &a, int *lds, u64 *tile_mask)
{
    int *key_l = reinterpret_cast<int *>(tile_mask + a.l_mask); // [l_key] first-touch keys of the pool's windows
    int *pl_l = key_l + a.l_key;                                // [l_pl]  copy of the window list
    int *ev_l = pl_l + a.l_pl;                                  // [l_ev]  copy of the event list
    const int s = blockIdx.x + a.s0;
    const int tid = threadIdx.x;
    if (a.advance && blockIdx.x == 0 && tid == 0) { *a.advance += 1; *a.advance_stat += 1; }
    if (a.n_tlist)
        for (int t = tid; t < a.n_tiles * 4; t += 1024) tile_mask[t] = 0; // (barriers follow before stage C)
    for (int t = tid; t < a.l_key; t += 1024) key_l[t] = INT_MAX;         // (barriers follow before stage B)
    if (a.active && !a.active[s]) { // no call is made for a stream without events (its maps are clear already)
        if (tid == 0) {
            a.nev[s] = 0;
            if (a.has_pool) { a.pnlist[s] = 0; if (a.pnevw) a.pnevw[s] = 0; }
            if (a.has_next && a.n_sites) a.n_nsites[s] = 0;
        }
        return;
    }
    if (a.prev_ts && tid == 0) a.prev_ts[s] = a.last_ts[s];
    // ---- A ----------------------------------------------------------------------------------------
    uint8_t *A = a.mapA + (long long)s * a.mstride;
    uint8_t *B = a.mapB ? a.mapB + (long long)s * a.mstride : nullptr;
    int *ev = a.ev + (long long)s * a.lstride;
    int ntouch = 0;
    const int nev = dev_map_to_list(A, B, a.HW, a.W, ev, lds, true, A, B, &ntouch, a.dflag ? a.dflag + (long long)s * a.mstride : nullptr, ev_l, a.l_ev);
    auto ev_at = [&](int e) __attribute__((always_inline)) { return e < a.l_ev ? ev_l[e] : ev[e]; };
    if (tid == 0) {
        a.nev[s] = nev;
        const int st = a.count_touch ? ntouch : nev;
        if (a.statA && st) atomicAdd(a.statA, (u64)st);
    }
    const int *src = ev, *src_l = ev_l;
    int nsrc = nev, src_cap = a.l_ev;
    __syncthreads(); // ev list visible to the whole block
    // ---- B ----------------------------------------------------------------------------------------
    if (a.has_pool) {
        int *key = a.pkey + (long long)s * a.pg.oh * a.pg.ow;
        uint8_t *pend = a.ppending + (long long)s * a.pmstride;
        int *plist = a.plist + (long long)s * a.plstride;
        // (the integer divisions of the index arithmetic are what this kernel spends its time on -- 16 waves share 4 SIMDs --
        //  so the 2x2 case shifts, and window coordinates are carried instead of being recovered from the site index)
        const bool two = a.pg.kh == 2 && a.pg.kw == 2;
        auto window = [&](int pk, int &wy, int &wx) __attribute__((always_inline)) { // cutils.pyx:83-87
            const int y = unpack_y(pk), x = unpack_x(pk);
            if (two) { wy = y >> 1; wx = x >> 1; } else { wy = y / a.pg.kh; wx = x / a.pg.kw; }
        };
        const bool lk = a.l_key > 0; // keys in LDS (fresh every launch) instead of the global map (INT_MAX between launches)
        for (int e = tid; e < nev; e += 1024) {
            int wy, wx;
            window(ev_at(e), wy, wx);
            const int site = wy * a.pg.ow + wx;
            if (lk) atomicMin(&key_l[site], e); else atomicMin(&key[site], e);
            pend[site] = 0;
        }
        __syncthreads();
        int base = 0;
        for (int e0 = 0; e0 < nev; e0 += 1024) {
            const int e = e0 + tid;
            int site = -1, wy = 0, wx = 0;
            if (e < nev) {
                window(ev_at(e), wy, wx);
                site = wy * a.pg.ow + wx;
                if ((lk ? key_l[site] : key[site]) != e) site = -1;
            }
            int tot;
            const int pos = block_excl_scan_1024(site >= 0 ? 1 : 0, lds, tot);
            if (site >= 0) {
                const int pk = pack_yx(wy, wx);
                plist[base + pos] = pk;
                if (base + pos < a.l_pl) pl_l[base + pos] = pk;
                if (!lk) key[site] = INT_MAX;
            }
            base += tot;
        }
        __syncthreads(); // pending bytes cleared above are visible
        const int nin_w = base; // windows of incoming events; pending windows follow
        if (tid == 0 && a.pnevw) a.pnevw[s] = base;
        base += dev_map_to_list(pend, nullptr, a.pg.oh * a.pg.ow, a.pg.ow, plist + base, lds, false, nullptr, nullptr, nullptr, nullptr,
                                pl_l + base, a.l_pl - base);
        if (tid == 0) {
            a.pnlist[s] = base;
            if (a.statP && base) atomicAdd(a.statP, (u64)base);
        }
        src = plist; src_l = pl_l; src_cap = a.l_pl;
        nsrc = base;
        __syncthreads();
        // ---- D ------------------------------------------------------------------------------------
        if (a.pe_inline) pool_eval_vec4_items(a.pe, s, base, nin_w, plist, pl_l, a.l_pl, tid, 1024);
    }
    // ---- C ----------------------------------------------------------------------------------------
    if (a.has_next) {
        uint8_t *T = a.n_touch + (long long)s * a.n_mstride;
        // one source per thread and iteration, its kh x kw targets in a loop; the tile coordinates of the targets follow
        // from ONE division per axis and source (the remainders are also the site's bit position in the tile's mask)
        for (int i = tid; i < nsrc; i += 1024) {
            const int pk = i < src_cap ? src_l[i] : src[i];
            const int y0 = unpack_y(pk) + a.n_pt, x0 = unpack_x(pk) + a.n_pl; // target of tap (0, 0); tap (dy, dx) -> (y0 - dy, x0 - dx)
            int ty0 = 0, ry0 = 0, tx0 = 0, rx0 = 0;
            if (a.n_tlist) { ty0 = y0 / a.n_TH; ry0 = y0 - ty0 * a.n_TH; tx0 = x0 / a.n_TW; rx0 = x0 - tx0 * a.n_TW; }
            int ty = ty0, ry = ry0;
            int pw = -1;     // mask word the pending bits belong to: neighbouring targets mostly share one (one LDS atomic
            u64 pbits = 0;   // per word instead of one per target)
            for (int dy = 0; dy < a.n_kh; ++dy) {
                const int oy = y0 - dy;
                if (oy >= 0 && oy < a.n_H) {
                    int tx = tx0, rx = rx0;
                    for (int dx = 0; dx < a.n_kw; ++dx) {
                        const int ox = x0 - dx;
                        if (ox >= 0 && ox < a.n_W) {
                            T[oy * a.n_W + ox] = 1;
                            if (a.n_tlist) {
                                const int q = ry * a.n_TW + rx;
                                const int wi = (ty * a.n_tilesX + tx) * 4 + (q >> 6);
                                if (wi != pw) {
                                    if (pw >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&tile_mask[pw]), pbits);
                                    pw = wi; pbits = 0;
                                }
                                pbits |= 1ull << (q & 63);
                            }
                        }
                        if (--rx < 0) { rx = a.n_TW - 1; --tx; }
                    }
                }
                if (--ry < 0) { ry = a.n_TH - 1; --ty; }
            }
            if (pw >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&tile_mask[pw]), pbits);
        }
        if (a.n_tlist) { // the next conv's active tiles of this stream (any order) with their site masks
            __syncthreads();
            tile_list_append<1024>(tile_mask, a.n_tiles, blockIdx.x, a.n_tlist, a.n_tmask, a.n_tcount, a.n_cap, a.n_TH * a.n_TW, a.n_nb);
        }
        if (a.n_sites) {
            __syncthreads();
            const int n = dev_map_to_list(T, nullptr, a.n_H * a.n_W, a.n_W, a.n_sites + (long long)s * a.n_lstride, lds, false,
                                          nullptr, nullptr, nullptr);
            if (tid == 0) a.n_nsites[s] = n;
        }
    }
}

__global__ __launch_bounds__(1024) void k_post(PostArgs a)
{
    __shared__ int lds[20];
    extern __shared__ u64 post_dyn[];
    post_body(a, lds, post_dyn);
}
// Two consecutive k_posts of a stream in one launch: the input layer's and the first conv layer's.  The second needs the first one's
// marks (the conv layer's touch map: written and read by the same block) and the leak sweep's -- not the values of the recompute
// between them, so the recompute follows both (one launch boundary less on every step's critical path).
__global__ __launch_bounds__(1024) void k_post_pair(PostArgs a, PostArgs b)
{
    __shared__ int lds[20];
    extern __shared__ u64 post_dyn[];
    post_body(a, lds, post_dyn);
    __threadfence_block();
    __syncthreads();
    post_body(b, lds, post_dyn);
}

#include "aev_book.hpp" // k_post2 / k_post2_pair: the same bookkeeping as bit-parallel map operations, 256 threads, a few KB of LDS (round 5)

// ------------------------------------------------------------------------------------------------
// Pool: re-evaluate the listed windows (maxpool.py:130-151 with min_argmax cutils.pyx:139-179)
// ------------------------------------------------------------------------------------------------
// FAST: 2x2 / stride 2 windows over a conv layer (every pool of the e-fcn): two aligned float2 loads per map.
template <bool FAST>
__global__ __launch_bounds__(256) void k_pool_eval(PoolArgs a)
{
    const int s = blockIdx.y + a.s0;
    const int n = a.nlist[s];
    const int total = n * a.C;
    const int *list = a.list + (long long)s * a.lstride;
    const int kk = a.kh * a.kw;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e / a.C, c = e - j * a.C; // channel fastest: a wave reads contiguous channel runs
        const int pk = list[j];
        const int wy = unpack_y(pk), wx = unpack_x(pk);
        PoolPick pp;
        if (FAST) {
            const long long o = (long long)s * a.src.bstride + ((long long)(2 * wy) * a.src.bW + 2 * wx) * a.C + c;
            const long long orow = (long long)a.src.bW * a.C;
            float2 s0, s1, a0, a1;
            s0.x = a.src.s[o]; s0.y = a.src.s[o + a.C]; s1.x = a.src.s[o + orow]; s1.y = a.src.s[o + orow + a.C];
            a0.x = a.src.a[o]; a0.y = a.src.a[o + a.C]; a1.x = a.src.a[o + orow]; a1.y = a.src.a[o + orow + a.C];
            const float al = a.src.alpha;
            const bool pl = a.plain != 0;
            pp.first(s0.x, pl ? 0.0f : __fmul_rn(a0.x, s0.x > 0.0f ? 1.0f : al), a0.x);
            pp.next(1, s0.y, pl ? 0.0f : __fmul_rn(a0.y, s0.y > 0.0f ? 1.0f : al), a0.y);
            pp.next(2, s1.x, pl ? 0.0f : __fmul_rn(a1.x, s1.x > 0.0f ? 1.0f : al), a1.x);
            pp.next(3, s1.y, pl ? 0.0f : __fmul_rn(a1.y, s1.y > 0.0f ? 1.0f : al), a1.y);
        } else {
            for (int r = 0; r < kk; ++r) {
                const int ry = r / a.kw, rx = r - ry * a.kw;
                float sv, L, D, A = 0.0f;
                src_fetch(a.src, s, c, wy * a.stride + ry, wx * a.stride + rx, sv, L, D);
                if (a.pfm) A = a.src.a[(long long)s * a.src.bstride + ((long long)(wy * a.stride + ry) * a.src.bW + wx * a.stride + rx) * a.C + c];
                if (a.plain) D = 0.0f;
                if (r == 0) pp.first(sv, D, A); else pp.next(r, sv, D, A);
            }
        }
        const long long po = (long long)s * a.istride + ((long long)wy * a.W + wx) * a.C + c;
        a.idx[po] = (uint8_t)pp.best;                                           // maxpool.py:145-151
        if (a.pfm) {
            a.pfm[po] = pp.bs; a.pact[po] = pp.ba;
            if (a.pflag && c == 0) a.pflag[(long long)s * a.mstride + wy * a.W + wx] = 1;
        }
        if (!a.plain && pp.bd != pp.slow)                                       // cutils.pyx:177, maxpool.py:140-142
            a.pending[(long long)s * a.mstride + wy * a.W + wx] = 1;            // same-value stores: no atomic needed
    }
}

// 2x2 / stride-2 windows over a conv layer with C % 4 == 0: one thread = one window x 4 consecutive channels, 16-byte
// loads of the four positions' channel runs, one 4-byte argmax store.
__global__ __launch_bounds__(256) void k_pool_eval_vec4(PoolArgs a)
{
    const int s = blockIdx.y + a.s0;
    pool_eval_vec4_items(a, s, a.nlist[s], a.nevw ? a.nevw[s] : -1, a.list + (long long)s * a.lstride, nullptr, 0,
                         blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// ------------------------------------------------------------------------------------------------
// Dense getters (layer.py:45-80) and the model output (event_numpy.py:79,101)
// ------------------------------------------------------------------------------------------------
// Rebuild a pool layer's copy of its inputs from its argmax bytes (all windows): after state init, reset, or any step
// made through the per-layer API (which does not maintain the copy).
struct PoolFillArgs {
    const float *s, *a; long long bstride; int bW;   // the conv below
    const uint8_t *idx; float *pfm, *pact; long long istride;
    int C, H, W, kh, kw, stride, s0;
};
__global__ __launch_bounds__(256) void k_pool_fill(PoolFillArgs a)
{
    const int s = blockIdx.y + a.s0;
    const int total = a.C * a.H * a.W;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int w = e / a.C, c = e - w * a.C;
        const int wy = w / a.W, wx = w - wy * a.W;
        const long long po = (long long)s * a.istride + e;
        const int j = a.idx[po];
        const int jy = j / a.kw, jx = j - jy * a.kw;
        const long long o = (long long)s * a.bstride + ((long long)(wy * a.stride + jy) * a.bW + wx * a.stride + jx) * a.C + c;
        a.pfm[po] = a.s[o]; // (the copy has idx's layout: [H][W][C])
        a.pact[po] = a.a[o];
    }
}

struct GetArgs {
    SrcView self;   // the layer itself seen as a source
    int which;      // AEV_SURFACE .. AEV_FEATUREMAP
    int hwc;        // 1: write [H][W][C] (transpose(1,2,0)), 0: [C][H][W]
    int s0;
    float *out;     // [nstreams][C*H*W]
};

__global__ __launch_bounds__(256) void k_getter(GetArgs a)
{
    const int s = blockIdx.y + a.s0;
    const int C = a.self.C, H = a.self.H, W = a.self.W;
    const int total = C * H * W;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int c = e / (H * W), p = e - c * H * W;
        const int y = p / W, x = p - y * W;
        float sv, L, D;
        src_fetch(a.self, s, c, y, x, sv, L, D);
        const float v = a.which == 0 ? sv : a.which == 1 ? L : a.which == 2 ? D : __fmul_rn(sv, L);
        const long long o = a.hwc ? (long long)p * C + c : e;
        a.out[(long long)blockIdx.y * total + o] = v;
    }
}

__global__ void k_fill_all_sites(int *list, int *nlist, long long lstride, int H, int W, int s0)
{
    const int s = blockIdx.y + s0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < H * W; p += gridDim.x * blockDim.x)
        list[(long long)s * lstride + p] = pack_yx(p / W, p % W);
    if (blockIdx.x == 0 && threadIdx.x == 0) nlist[s] = H * W;
}

// dense mode: every site of the first conv is active (the later layers follow: every recomputed site emits an event)
__global__ void k_fill_map(uint8_t *map, long long mstride, int HW, int s0, const int *active)
{
    const int s = blockIdx.y + s0;
    if (active && !active[s]) return;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) map[(long long)s * mstride + p] = 1;
}

// a queued run begins: the device step counter and the run's range {first step, end} (k_integ_step / k_leak_sweep_ahead read them)
__global__ void k_queue_begin(int *qstep, int *qrun, int t0, int t1)
{
    *qstep = t0;
    qrun[0] = t0; qrun[1] = t1;
}
__global__ void k_advance(int *qstep, u64 *stat_steps)
{
    *qstep += 1;
    *stat_steps += 1;
}

__global__ void k_fill_int(int *p, long long n, int v)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone ops (src/libs/cutils.pyx)
// ------------------------------------------------------------------------------------------------
// data half of im2col_event (cutils.pyx:116-127): thread per (site, row)
__global__ void k_op_gather(const float *img, int C, int H, int W, int kh, int kw, int stride, int chan_as_cols,
                            const int *list, const int *nlist, float *cols)
{
    const int n = *nlist;
    const int kk = kh * kw, rows = C * kk;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * rows;
         e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e / rows), k = (int)(e - (long long)j * rows);
        const int c = k / kk, r = k - c * kk;
        const int dy = r / kw, dx = r - dy * kw;
        const int pk = list[j];
        const int ty = unpack_y(pk) * stride, tx = unpack_x(pk) * stride;
        const float v = img[((long long)c * H + ty + dy) * W + tx + dx];
        if (chan_as_cols) cols[((long long)C * j + c) * kk + r] = v;
        else cols[(long long)j * rows + k] = v;
    }
}

// min_argmax (cutils.pyx:139-179): thread per column
__global__ void k_op_min_argmax(const float *mx, const float *mn, int R, int M, int *argmax, int *not_argmin)
{
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const float *a = mx + (long long)m * R, *b = mn + (long long)m * R;
        int best = 0, slow = 0;
        for (int r = 0; r < R; ++r) {
            if (a[r] > a[best]) best = r;
            else if (r > 0 && a[r] == a[best] && b[r] < b[best]) best = r;
            if (b[r] < b[slow]) slow = r;
        }
        argmax[m] = best;
        not_argmin[m] = b[best] != b[slow] ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone conv ops the reference's callers import by name: conv2d_event (src/layers/conv2d.py:144-181), the dense conv2d
// (:184-230) and the dense im2col (src/layers/functional.py:4-34).  One-shot host-pointer entry points for API parity (the
// stateful layers never go through them: they gather from the resident [H,W,C] state).  Plain [C,H,W] images.
// ------------------------------------------------------------------------------------------------
struct OpConvArgs {
    const float *img;          // [B][C][H][W]
    int B, C, H, W;
    int pt, pl;                // zero padding applied on the fly (a padded position contributes fmaf(w, +0, acc), as the padded array does)
    const float *wt;           // [chain position][CoutP]  (chain_pos order: the arithmetic contract)
    const float *bias;         // [Cout] or nullptr
    int Cout, CoutP, kh, kw, stride, oh, ow;
    const int *list;           // site list (y << 16 | x in output space), or nullptr: every site of every image
    int n;                     // list mode: number of listed sites
    float *out;                // list mode: [Cout][n]; dense: [B][Cout][oh][ow]
};

// thread per (cout, site), site fastest: the weight is wave-uniform, image reads and output writes are contiguous along x
__global__ __launch_bounds__(256) void k_op_conv(OpConvArgs a)
{
    const int kk = a.kh * a.kw;
    const long long per = a.list ? (long long)a.n : (long long)a.oh * a.ow;
    const long long total = (a.list ? 1 : a.B) * (long long)a.Cout * per;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(e / ((long long)a.Cout * per));
        const long long rem = e - (long long)b * a.Cout * per;
        const int co = (int)(rem / per), j = (int)(rem - (long long)co * per);
        int oy, ox;
        if (a.list) { const int pk = a.list[j]; oy = unpack_y(pk); ox = unpack_x(pk); }
        else { oy = j / a.ow; ox = j - oy * a.ow; }
        const int y0 = oy * a.stride - a.pt, x0 = ox * a.stride - a.pl;
        const float *img = a.img + (long long)b * a.C * a.H * a.W;
        float acc = 0.0f;
        int q = 0;
        for (int g = 0; g < a.C; g += 4) {                  // the chain: channel groups of four, taps ascending, the group's channels ascending
            const int gsz = min(4, a.C - g);
            for (int r = 0; r < kk; ++r) {
                const int dy = r / a.kw, dx = r - dy * a.kw;
                const int y = y0 + dy, x = x0 + dx;
                const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
                for (int c = g; c < g + gsz; ++c, ++q) {
                    const float v = in ? img[((long long)c * a.H + y) * a.W + x] : 0.0f;
                    acc = __fmaf_rn(a.wt[(long long)q * a.CoutP + co], v, acc);
                }
            }
        }
        if (a.bias) acc = __fadd_rn(acc, a.bias[co]);
        a.out[e] = acc;
    }
}

// dense im2col (functional.py:4-34): out [C*kh*kw][B*oh*ow] C-order, a pure copy of 4- or 8-byte elements
template <typename T>
__global__ __launch_bounds__(256) void k_op_im2col(const T *img, int B, int C, int H, int W, int kh, int kw, int stride, int oh, int ow, T *out)
{
    const int kk = kh * kw;
    const long long cols = (long long)B * oh * ow, total = (long long)C * kk * cols;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e / cols);
        const long long col = e - (long long)row * cols;
        const int c = row / kk, r = row - c * kk, dy = r / kw, dx = r - dy * kw;
        const int b = (int)(col / ((long long)oh * ow)), p = (int)(col - (long long)b * oh * ow), oy = p / ow, ox = p - oy * ow;
        out[e] = img[(((long long)b * C + c) * H + oy * stride + dy) * W + ox * stride + dx];
    }
}

// ------------------------------------------------------------------------------------------------
// NEXT ROW: YOLO decode + NMS, the step right after the path
// (src/libs/viz.py:27-46,125-164 up to the drawing; src/libs/utils.py:38-118).  One block per stream / batch entry.
// ------------------------------------------------------------------------------------------------
// Greedy NMS over n boxes held in LDS (centre x, y, w, h; float32 arithmetic in the reference's operation order).
// Visits by descending score, equal scores by descending index.  order[] (global, may be null) receives the pick order.
__device__ __forceinline__ int nms_block(const float *bx, const float *sc, int *alive, int *picked, int n, float thr, int *order,
                                         int *red /* [2*4 + 2] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int np_ = 0;
    for (;;) {
        float bs = 0.0f; int bi = -1;
        for (int j = tid; j < n; j += 256)
            if (alive[j] && (bi < 0 || sc[j] > bs || (sc[j] == bs && j > bi))) { bs = sc[j]; bi = j; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const float os = __shfl_xor(bs, d, 64); const int oi = __shfl_xor(bi, d, 64);
            if (oi >= 0 && (bi < 0 || os > bs || (os == bs && oi > bi))) { bs = os; bi = oi; }
        }
        if (lane == 0) { red[wv] = bi; reinterpret_cast<float *>(red)[4 + wv] = bs; }
        __syncthreads();
        if (tid == 0) {
            int i = -1; float v = 0.0f;
            for (int w = 0; w < 4; ++w) {
                const int oi = red[w]; const float os = reinterpret_cast<float *>(red)[4 + w];
                if (oi >= 0 && (i < 0 || os > v || (os == v && oi > i))) { i = oi; v = os; }
            }
            red[8] = i;
            if (i >= 0) { alive[i] = 0; picked[i] = 1; if (order) order[np_] = i; }
        }
        __syncthreads();
        const int i = red[8];
        if (i < 0) break;
        ++np_;
        const float xi = bx[4 * i], yi = bx[4 * i + 1], wi = bx[4 * i + 2], hi = bx[4 * i + 3];
        const float ai = __fmul_rn(wi, hi);
        for (int j = tid; j < n; j += 256) {
            if (!alive[j]) continue;
            const float xj = bx[4 * j], yj = bx[4 * j + 1], wj = bx[4 * j + 2], hj = bx[4 * j + 3];
            const float xx1 = fmaxf(xi - wi / 2, xj - wj / 2), yy1 = fmaxf(yi - hi / 2, yj - hj / 2);
            const float xx2 = fminf(xi + wi / 2, xj + wj / 2), yy2 = fminf(yi + hi / 2, yj + hj / 2);
            const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
            const float inter = __fmul_rn(ww, hh);
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(__fmul_rn(wj, hj), ai), inter)); // utils.py:105
            if (iou >= thr) alive[j] = 0;                                                          // utils.py:109-110
        }
        __syncthreads();
    }
    return np_;
}

struct DetectArgs {
    const float *pred; long long pstride; // [S][gh][gw][C+5B]
    int gh, gw, C, B, h_image, w_image;
    float conf_thr; int use_nms; float iou_thr;
    float *out;      // [S][nb][7] = x, y, w, h, conf, class, valid
    int *order;      // [S][nb] pick order (may be null)
    int *npicked;    // [S] (may be null)
};

__global__ __launch_bounds__(256) void k_yolo_detect(DetectArgs a)
{
    extern __shared__ float lds[];
    const int nb = a.gh * a.gw * a.B, D = a.C + 5 * a.B;
    float *bx = lds, *sc = lds + 4 * nb;
    int *alive = reinterpret_cast<int *>(sc + nb), *picked = alive + nb, *red = picked + nb;
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *pred = a.pred + (long long)s * a.pstride;
    float *out = a.out + (long long)s * nb * 7;
    int any = 0;
    for (int i = tid; i < nb; i += 256) {
        const int cell = i / a.B, b = i - cell * a.B, row = cell / a.gw, col = cell - row * a.gw;
        const float *p = pred + (long long)cell * D + a.C + b * 5;
        const float x = __fmul_rn(__fdiv_rn(__fadd_rn(p[0], (float)col), (float)a.gw), (float)a.w_image); // viz.py:40
        const float y = __fmul_rn(__fdiv_rn(__fadd_rn(p[1], (float)row), (float)a.gh), (float)a.h_image); // viz.py:41
        const float w = __fmul_rn(__fmul_rn(p[2], p[2]), (float)a.w_image);                               // viz.py:42
        const float h = __fmul_rn(__fmul_rn(p[3], p[3]), (float)a.h_image);                               // viz.py:43
        const float conf = p[4];
        int best = 0;
        float bv = __fmul_rn(pred[(long long)cell * D], conf);                                            // viz.py:144,170
        for (int c = 1; c < a.C; ++c) {
            const float v = __fmul_rn(pred[(long long)cell * D + c], conf);
            if (v > bv) { bv = v; best = c; }
        }
        const int valid = conf > a.conf_thr;                                                              // viz.py:150
        bx[4 * i] = x; bx[4 * i + 1] = y; bx[4 * i + 2] = w; bx[4 * i + 3] = h; sc[i] = conf;
        alive[i] = valid; picked[i] = 0;
        float *o = out + (long long)i * 7;
        o[0] = x; o[1] = y; o[2] = w; o[3] = h; o[4] = conf; o[5] = (float)best; o[6] = valid ? 1.0f : 0.0f;
        any |= valid;
    }
    any = __syncthreads_or(any);
    int np_ = 0;
    if (a.use_nms && any) {                                                                               // viz.py:152
        np_ = nms_block(bx, sc, alive, picked, nb, a.iou_thr, a.order ? a.order + (long long)s * nb : nullptr, red);
        for (int i = tid; i < nb; i += 256)
            if (!picked[i]) { float *o = out + (long long)i * 7; o[0] = o[1] = o[2] = o[3] = o[4] = 0.0f; o[6] = 0.0f; } // viz.py:154-164
    }
    if (tid == 0 && a.npicked) a.npicked[s] = np_;
}

// apply_nms on caller-provided boxes (utils.py:38-118): one block per batch entry
__global__ __launch_bounds__(256) void k_nms(const float *boxes, const float *scores, const uint8_t *valid, int n, float thr, int *order,
                                             int *npicked)
{
    extern __shared__ float lds[];
    float *bx = lds, *sc = lds + 4 * n;
    int *alive = reinterpret_cast<int *>(sc + n), *picked = alive + n, *red = picked + n;
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) {
        for (int q = 0; q < 4; ++q) bx[4 * i + q] = boxes[((long long)s * n + i) * 4 + q];
        sc[i] = scores[(long long)s * n + i];
        alive[i] = valid ? valid[(long long)s * n + i] != 0 : 1;
        picked[i] = 0;
    }
    __syncthreads();
    const int np_ = nms_block(bx, sc, alive, picked, n, thr, order + (long long)s * n, red);
    if (threadIdx.x == 0) npicked[s] = np_;
}

// ------------------------------------------------------------------------------------------------
// NEXT ROW: the event-batch feeder on the device -- Runner.data_transform (src/libs/runner.py:23-33: ts -= ts[0],
// utils.center_crop src/libs/utils.py:4-35 with its swapped-offset quirk, stack (y, x, ts)) and the batching rule of Runner.run
// (runner.py:65-72: np.array_split into ceil(n / batch_event_size) chunks, or time bins of batch_event_usec via np.digitize) for S
// whole samples at once, straight into the device-resident event queue.  Integer / index work, bound by HBM: one 1024-thread block
// per sample, three passes over its events (window, ordered compaction, chunk index), one scan of the (step, stream) counts, one
// scatter.
// ------------------------------------------------------------------------------------------------
struct FeedArgs {
    const int *x, *y, *ts; const long long *soff;   // raw events of all samples, sample s = [soff[s], soff[s+1])
    int crop, new_h, new_w;                         // crop != 0: utils.center_crop to (new_h, new_w)
    int size, usec;                                 // batching rule: size > 0 (runner.py:71-72) or usec > 0 (:66-69)
    int H, W, Tcap;                                 // the handle's surface (bounds check); capacity of cstart per stream
    int *cx, *cy, *cts, *chunk;                     // compacted events + their chunk index (same offsets as the raw arrays)
    int *n_in, *nchunks, *cstart;                   // [S], [S], [S][Tcap + 1]
    int *bad;                                       // set when an event lands outside the surface
};

__device__ __forceinline__ int floordiv2(int a) { return (a - (a < 0 ? 1 : 0)) / 2; } // Python's a // 2

__global__ __launch_bounds__(1024) void k_feed_prepare(FeedArgs a)
{
    __shared__ int lds[24];
    __shared__ int red[6];
    const int s = blockIdx.x, tid = threadIdx.x;
    const long long base = a.soff[s];
    const int n = (int)(a.soff[s + 1] - base);
    int *cstart = a.cstart + (long long)s * (a.Tcap + 1);
    if (n == 0) { if (tid == 0) { a.n_in[s] = 0; a.nchunks[s] = 0; } return; }
    const int *x = a.x + base, *y = a.y + base, *ts = a.ts + base;
    int *cx = a.cx + base, *cy = a.cy + base, *cts = a.cts + base, *chunk = a.chunk + base;
    const int ts0 = ts[0];                                                       // runner.py:25
    if (tid == 0) { red[0] = INT_MAX; red[1] = INT_MIN; red[2] = INT_MAX; red[3] = INT_MIN; red[4] = INT_MAX; red[5] = INT_MAX; }
    __syncthreads();
    int x_lo = 0, x_hi = INT_MAX, y_lo = 0, y_hi = INT_MAX;
    if (a.crop) {                                                                // utils.py:14-21
        int mnx = INT_MAX, mxx = INT_MIN, mny = INT_MAX, mxy = INT_MIN;
        for (int i = tid; i < n; i += 1024) { const int vx = x[i], vy = y[i]; mnx = min(mnx, vx); mxx = max(mxx, vx); mny = min(mny, vy); mxy = max(mxy, vy); }
        atomicMin(&red[0], mnx); atomicMax(&red[1], mxx); atomicMin(&red[2], mny); atomicMax(&red[3], mxy);
        __syncthreads();
        const int new_top = floordiv2(red[1] - red[0] - a.new_w), new_left = floordiv2(red[3] - red[2] - a.new_h);   // (the quirk: top from x, left from y)
        x_lo = new_left; x_hi = new_left + a.new_w; y_lo = new_top; y_hi = new_top + a.new_h;
    }
    int run = 0, mnx = INT_MAX, mny = INT_MAX;
    for (int i0 = 0; i0 < n; i0 += 1024) {                                        // ordered compaction of the events inside (utils.py:23-26)
        const int i = i0 + tid;
        int vx = 0, vy = 0, vt = 0, in = 0;
        if (i < n) { vx = x[i]; vy = y[i]; vt = ts[i] - ts0; in = !a.crop || (vx >= x_lo && vx < x_hi && vy >= y_lo && vy < y_hi); }
        int tot;
        const int pos = run + block_excl_scan_1024(in, lds, tot);
        if (in) { cx[pos] = vx; cy[pos] = vy; cts[pos] = vt; mnx = min(mnx, vx); mny = min(mny, vy); }
        run += tot;
    }
    atomicMin(&red[4], mnx); atomicMin(&red[5], mny);
    __syncthreads();
    const int m = run, sx = a.crop ? red[4] : 0, sy = a.crop ? red[5] : 0;       // utils.py:27-28: new_x -= new_x.min()
    int bad = 0;
    for (int i = tid; i < m; i += 1024) {
        const int vx = cx[i] - sx, vy = cy[i] - sy;
        cx[i] = vx; cy[i] = vy;
        bad |= vx < 0 || vx >= a.W || vy < 0 || vy >= a.H;
    }
    if (bad) *a.bad = 1;
    __syncthreads();
    if (m == 0) { if (tid == 0) { a.n_in[s] = 0; a.nchunks[s] = 0; } return; }
    if (a.usec <= 0) {                                                           // runner.py:71-72: np.array_split(events, ceil(m / size))
        const int nb = (m + a.size - 1) / a.size, q = m / nb, r = m % nb;
        for (int i = tid; i < m; i += 1024) chunk[i] = i < r * (q + 1) ? i / (q + 1) : r + (i - r * (q + 1)) / q;
        for (int t = tid; t <= nb; t += 1024) cstart[t] = t * q + min(t, r);
        if (tid == 0) { a.n_in[s] = m; a.nchunks[s] = nb; }
        return;
    }
    // runner.py:66-69: bins = arange(0, ts_last, usec); ids = digitize(ts, bins); split where the id changes
    const int last = cts[m - 1], nbins = last > 0 ? (last + a.usec - 1) / a.usec : 0;
    run = 0;
    for (int i0 = 0; i0 < m; i0 += 1024) {
        const int i = i0 + tid;
        int flag = 0;
        if (i < m && i > 0) flag = min(cts[i] / a.usec + 1, nbins) != min(cts[i - 1] / a.usec + 1, nbins);
        int tot;
        const int c = run + block_excl_scan_1024(flag, lds, tot) + flag;        // inclusive: chunk index of event i
        if (i < m) { chunk[i] = c; if (flag || i == 0) cstart[c] = i; }
        run += tot;
    }
    if (tid == 0) { cstart[run + 1] = m; a.n_in[s] = m; a.nchunks[s] = run + 1; }
}

// exclusive scan of the events per (step, stream) cell in queue order -> offsets [T*S + 1]
__global__ __launch_bounds__(1024) void k_feed_offsets(int S, int T, int Tcap, const int *cstart, const int *nchunks, long long *qoff)
{
    __shared__ int lds[24];
    long long base = 0;
    const long long cells = (long long)T * S;
    for (long long c0 = 0; c0 < cells; c0 += 1024) {
        const long long c = c0 + threadIdx.x;
        int v = 0;
        if (c < cells) {
            const int t = (int)(c / S), s = (int)(c - (long long)t * S);
            if (t < nchunks[s]) v = cstart[(long long)s * (Tcap + 1) + t + 1] - cstart[(long long)s * (Tcap + 1) + t];
        }
        int tot;
        const int e = block_excl_scan_1024(v, lds, tot);
        if (c < cells) qoff[c] = base + e;
        base += tot;
    }
    if (threadIdx.x == 0) qoff[cells] = base;
}

__global__ __launch_bounds__(256) void k_feed_scatter(FeedArgs a, int S, const long long *qoff, int *qyxt)
{
    const int s = blockIdx.y;
    const long long base = a.soff[s];
    const int m = a.n_in[s];
    const int *cstart = a.cstart + (long long)s * (a.Tcap + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int t = a.chunk[base + i];
        const long long row = qoff[(long long)t * S + s] + (i - cstart[t]);
        qyxt[3 * row] = a.cy[base + i]; qyxt[3 * row + 1] = a.cx[base + i]; qyxt[3 * row + 2] = a.cts[base + i];   // runner.py:32: stack([y, x, ts])
    }
}

// ------------------------------------------------------------------------------------------------
// NEXT ROW: N-MNIST / N-Caltech101 binary decoder (src/readers/file_reader.py:30-58).
// 5 bytes per record: x, y, (p << 7 | ts[22:16]), ts[15:8], ts[7:0]; y == 240 marks a timestamp overflow that adds 2^13 to
// itself and every later record (:44-47) and is dropped (:50).  Byte work bound by HBM: three passes over 1024-record
// blocks -- per-block (overflow count, event count), scan of the block pairs, decode + ordered scatter.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ndec_count(const uint8_t *raw, long long n, int *blk_ovf, int *blk_td)
{
    __shared__ int lds[20];
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    const int ovf = i < n && raw[5 * i + 1] == 240;
    const int td = i < n && !ovf;
    int t0, t1;
    block_excl_scan_1024(ovf, lds, t0);
    block_excl_scan_1024(td, lds, t1);
    if (threadIdx.x == 0) { blk_ovf[blockIdx.x] = t0; blk_td[blockIdx.x] = t1; }
}

// exclusive scan of the per-block pairs in place (one block; nblk <= 1 M records / 1024 ... looped)
__global__ __launch_bounds__(1024) void k_ndec_scan(int *blk_ovf, int *blk_td, int nblk, long long *total_td)
{
    __shared__ int lds[20];
    int base0 = 0, base1 = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int v0 = b < nblk ? blk_ovf[b] : 0, v1 = b < nblk ? blk_td[b] : 0;
        int t0, t1;
        const int e0 = block_excl_scan_1024(v0, lds, t0);
        const int e1 = block_excl_scan_1024(v1, lds, t1);
        if (b < nblk) { blk_ovf[b] = base0 + e0; blk_td[b] = base1 + e1; }
        base0 += t0; base1 += t1;
    }
    if (threadIdx.x == 0) *total_td = base1;
}

__global__ __launch_bounds__(1024) void k_ndec_emit(const uint8_t *raw, long long n, const int *blk_ovf, const int *blk_td, int *x,
                                                    int *y, int *ts, int *p)
{
    __shared__ int lds[20];
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
    if (i < n) { const uint8_t *r = raw + 5 * i; b0 = r[0]; b1 = r[1]; b2 = r[2]; b3 = r[3]; b4 = r[4]; }
    const int ovf = i < n && b1 == 240;
    const int td = i < n && !ovf;
    int t;
    const int eo = block_excl_scan_1024(ovf, lds, t);
    const int et = block_excl_scan_1024(td, lds, t);
    if (td) {
        const unsigned carry = (unsigned)(blk_ovf[blockIdx.x] + eo) << 13;      // markers strictly before this record ...
        const long long o = (long long)blk_td[blockIdx.x] + et;
        x[o] = (int)b0; y[o] = (int)b1;                                         // :38-39
        ts[o] = (int)((((b2 & 127u) << 16) | (b3 << 8) | b4) + carry);          // :41,46  (... a marker is never an event)
        p[o] = (int)((b2 & 128u) >> 7);                                         // :40
    }
}


#include "aev_conv_patch.hpp"
#include "aev_conv_x6.hpp"
#include "aev_conv_1x1.hpp"


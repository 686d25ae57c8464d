// aev_device.hpp -- device-side data views and helpers shared by the gfx950 kernels.
//
// Data layout in HBM (see DESIGN.md "Layout"):
//   * every stateful map is stream-major and CHANNEL-FASTEST: [S][H][W][C] float32 (padded to 64 floats per
//     stream).  The reference keeps [C,H,W] (conv2d.py:58-63); here every access pattern wants a site's channels
//     side by side: a recomputed site is one contiguous C*4-byte segment (full-line writes instead of C scattered
//     4-byte read-modify-writes -- measured 170 us vs the same kernel's 9 gathers for conv1), a pooling window reads
//     4 contiguous channel vectors, a receptive-field patch element is a contiguous channel run.  The dense getters
//     transpose back to [C,H,W] for the caller;
//   * event sets are byte maps (one 0/1 byte per row-major site): thousands of threads mark sites with plain
//     same-value stores (no atomics), and a row-major ordered event list (np.where order, conv2d.py:131)
//     falls out of popcount + prefix sum over 32-byte groups;
//   * event lists are int32 (y << 16 | x), per stream, capacity H*W, count in a device int;
//   * the max-pool argmax (maxpool.py:33-35 keeps int32) is one byte per pooled element.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

#define AEV_KIND_INTEG 0
#define AEV_KIND_CONV 1
#define AEV_KIND_POOL 2

__host__ __device__ __forceinline__ int pack_yx(int y, int x) { return (y << 16) | x; }
__host__ __device__ __forceinline__ int unpack_y(int p) { return p >> 16; }
__host__ __device__ __forceinline__ int unpack_x(int p) { return p & 0xffff; }

// Order of a conv output's fmaf chain (the arithmetic contract, DESIGN.md "Numerics"): input channels in groups of four (the
// last group may be smaller), per group the kernel taps ascending, per tap the group's channels ascending.  Position q of the
// chain <-> (channel c, tap r); the transposed weights are stored in chain order.  For Cin = 1 and for 1x1 kernels this is the
// plain ascending order; for the MFMA kernels the four (two) k of one instruction are the channels of a group at one tap.
__host__ __device__ __forceinline__ int chain_pos(int c, int r, int cin, int kk)
{
    const int g = c >> 2, gsz = min(4, cin - 4 * g);
    return g * 4 * kk + r * gsz + (c & 3);
}
__host__ __device__ __forceinline__ void chain_decode(int q, int cin, int kk, int &c, int &r)
{
    const int g = q / (4 * kk), rem = q - g * 4 * kk, gsz = min(4, cin - 4 * g);
    r = rem / gsz;
    c = 4 * g + rem - r * gsz;
}

// How a consumer layer reads the OUTPUT maps of its previous layer without materialising them
// (the reference materialises three dense maps per pool layer per step: maxpool.py:42-79).
struct SrcView {
    int kind;            // kind of the previous layer
    int C, H, W;         // its output shape
    int bkind;           // kind of the layer that owns the float state (itself, or the pool's input)
    int bH, bW;          // that layer's H, W
    const float *s;      // integration surface / conv _featuremap            [S][bH][bW][C]
    const float *a;      // conv _conv_actfn (nullptr for integration)        [S][bH][bW][C]
    long long bstride;
    float alpha;         // leaky-ReLU slope of the owning conv
    const uint8_t *idx;  // pool argmax [S][H][W][C] (nullptr unless kind == POOL)
    long long istride;
    int pkh, pkw, pstride;
};

// surface(), layer_actfn(), conv_actfn() of the previous layer at (c,y,x) of ITS output map:
//   integration: s = S, L = (S>0), D = (S>0)                        integration.py:30-43
//   conv       : s = fm, L = fm>0 ? 1 : alpha, D = act * L          conv2d.py:80-94
//   pool       : the owning map's triple at the stored argmax       maxpool.py:42-79
__device__ __forceinline__ void src_fetch(const SrcView &v, int stream, int c, int y, int x, float &s, float &L,
                                          float &D)
{
    int by = y, bx = x;
    if (v.kind == AEV_KIND_POOL) {
        const int j = v.idx[(long long)stream * v.istride + ((long long)y * v.W + x) * v.C + c];
        int jy, jx;
        if (v.pkw == 2) { jy = j >> 1; jx = j & 1; } else { jy = j / v.pkw; jx = j - jy * v.pkw; }
        by = y * v.pstride + jy;
        bx = x * v.pstride + jx;
    }
    const long long o = (long long)stream * v.bstride + ((long long)by * v.bW + bx) * v.C + c;
    s = v.s[o];
    if (v.bkind == AEV_KIND_INTEG) {
        L = s > 0.0f ? 1.0f : 0.0f;
        D = L;
    } else {
        L = s > 0.0f ? 1.0f : v.alpha;
        D = __fmul_rn(v.a[o], L);
    }
}

// Exclusive prefix sum over a 1024-thread block (16 wavefronts of 64).  lds: >= 17 ints.
// Returns the exclusive prefix of v for this thread, total in `total`.  All threads must call.
__device__ __forceinline__ int block_excl_scan_1024(int v, int *lds, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        const int w = lane < 16 ? lds[lane] : 0;
        int winc = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int t = __shfl_up(winc, d, 64);
            if (lane >= d) winc += t;
        }
        if (lane < 16) lds[lane] = winc - w;
        if (lane == 15) lds[16] = winc;
    }
    __syncthreads();
    const int res = inc - v + lds[wv];
    total = lds[16];
    __syncthreads();
    return res;
}

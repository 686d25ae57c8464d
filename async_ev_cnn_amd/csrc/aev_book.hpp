// aev_book.hpp -- k_post2: the bookkeeping between two heavy kernels as BIT-PARALLEL map operations (round 5).
// Included by aev_kernels.hpp behind k_post (PostArgs, PoolArgs, PoolPick, tile lists are defined there).
//
// k_post (round 1-4) walks LISTS: one thread per event finds its pooling window with an atomicMin on a first-touch key, re-reads the
// key, marks the 3x3 targets of every listed window with nine byte stores and an LDS atomic per tile-mask word -- 1 024 threads, up to
// 128 KB of LDS, ~1 700 VALU instructions per wave and a chain of dependent round trips: 30 us per launch at 256 streams, 25 us
// for ONE stream, and a footprint nothing can run beside (0.055 of the HBM peak, eight times on the critical path).
//
// Everything k_post computes is a function of a few small 0/1 MAPS, and every ordered list the reference returns is an order OF MAP
// POSITIONS:
//   A  events out = np.where(touched | sign changed) in row-major order (conv2d.py:129-131, integration.py:83): the set bits of
//      E = A | B in word order;
//   B  the pool layer's windows in first-touch order of ITS input events (cutils.pyx:73-112 over maxpool.py:123-126).  The input
//      events are E's bits in row-major order, so within window row wy the windows that have an event in map row 2 wy come first (by
//      x), then those whose first event lies in row 2 wy + 1: two bit rows, Top = pairs-OR of E[2 wy], Bot = pairs-OR of
//      E[2 wy + 1] & ~Top -- no keys, no atomics; the pending windows behind them are np.where(recompute & ~incoming), row-major
//      (maxpool.py:118-126);
//   C  the consumer's touched sites (im2col_event's site enumeration, cutils.pyx:73-112, as a set) = the DILATION of the source
//      bits by the kernel window: OR of shifted rows; the tile kernel's 256-bit site masks are bit fields of those rows.
// One 256-thread block per stream keeps the maps as bit rows in a few KB of LDS (e-fcn: < 20 KB), converts bytes <-> bits at the
// HBM boundary (the byte maps stay: thousands of threads of other kernels mark them with plain same-value stores) and emits the
// lists with popcounts and a block scan.  Bit-identical outputs to k_post (which stays as the fallback for geometries this kernel
// does not take: pool windows other than 2x2, maps beyond its LDS budget); AEV_POST_V1=1 forces the old kernel.
//
// What the kernel's time is made of (AEV_POST_STAMPS, first version of this file: same speed as k_post): a dependent trip to memory
// costs 1.5-2 us under load and a __syncthreads() is one -- hipcc puts s_waitcnt vmcnt(0) in front of every s_barrier, and gfx9
// counts STORES in vmcnt: every phase that ends with stores and a barrier waits for the stores' acknowledgements.  So:
//   * ONE trip for everything the block reads (all map loads of a thread are in flight together; the stream's activity flag is a
//     MASK on the loaded bits, not a branch in front of the loads; the "may decay" flags are set with byte stores instead of a
//     read-modify-write of the flag map);
//   * everything after that works on LDS between LDS-ONLY barriers (s_waitcnt lgkmcnt(0); s_barrier): stores are fire-and-forget;
//   * the tile list's one atomic round trip (the position of the stream's tiles in the launch's list) is requested as soon as the
//     tile masks exist and used behind the other emissions;
//   * list emission is balanced: a first pass (thread-contiguous words) turns popcounts into a per-word offset table, a second one
//     hands HALF words round robin to the threads -- the events of an N-Caltech101-like stream are edges, a few dense words;
//   * a small pool layer evaluated in place has two windows' loads (16 x 16 B per thread) in flight: at 256 streams that evaluation is
//     bandwidth, not latency (pool3 of the e-fcn: 231 MB per launch).
#pragma once

// Threads per block: the algorithm needs few, but a lone wave per SIMD hides nothing -- every phase is a few hundred dependent
// instructions, LDS round trips and one trip to memory (AEV_POST_STAMPS).  Measured (round 5, ms per step; 256 / 512 / 1024 threads;
// k_post: the old kernel): 256 streams 1.91 / 1.88 / 1.90 (k_post 1.90); one stream 0.158 / 0.147 / 0.143 (k_post 0.157).
#define AEV_POST2_NT_MANY 512
#define AEV_POST2_NT_FEW 1024
#define AEV_POST2_FEW_STREAMS 16 // up to so many streams per launch: the 1024-thread instance
#define AEV_POST2_LDS_MAX (48 * 1024)

struct Post2Lay {
    int ok;                                   // 0: this launch takes k_post
    int WW, PWW, SWW, TWW;                    // u64 words per bit row: own map, pool map, stage C's source map, the consumer's map
    int sH, sW;                               // stage C's source map (the pool layer's or the own one)
    int o_E, o_top, o_bot, o_pend, o_src, o_T, o_tmask; // word offsets of the bit rows in the dynamic LDS
    int words;                                // u64 words of LDS to clear at the start of a body
    int o_pre;                                // int offset of the per-word offset table of the list emissions
    int o_plm, plm_cap;                       // int offset / capacity of the LDS mirror of the window list (inline pool evaluation)
    unsigned long long *dbg;                  // diagnostics (AEV_POST_STAMPS): [16] per-phase sums of block 0's 100-MHz ticks, [15] = launches; else nullptr
};

// barrier for data exchanged through LDS only: no wait for the wave's outstanding global loads / stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive block scan with ONE LDS-only barrier: consecutive scans alternate between two scratch areas (sc: 2 * NT / 64 ints; k: scan counter)
template <int NT>
__device__ __forceinline__ int block_excl_scan_lds(int v, int *sc, int &k, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *my = sc + (k & 1) * (NT / 64);
    ++k;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) my[wv] = inc;
    lds_barrier();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const int t = my[w]; base += w < wv ? t : 0; tot += t; }
    total = tot;
    return inc - v + base;
}

// eight 0/1 bytes -> eight bits (byte i -> bit i): the partial products of the multiply land on distinct bits, no carries
__device__ __forceinline__ unsigned pack8(u64 w) { return (unsigned)((w * 0x0102040810204080ull) >> 56); }
__device__ __forceinline__ unsigned pack16(const ulonglong2 &v) { return pack8(v.x) | (pack8(v.y) << 8); }
// eight bits -> eight 0/1 bytes
__device__ __forceinline__ u64 spread8(unsigned b)
{
    const u64 t = ((u64)(b & 0xffu) * 0x0101010101010101ull) & 0x8040201008040201ull; // byte i: bit i of b, in place
    return ((t + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
}
// OR of neighbouring bit pairs, compressed: bit i of the result = bit 2i | bit 2i + 1 (32 results)
__device__ __forceinline__ u64 pair_or_compress(u64 w)
{
    u64 t = (w | (w >> 1)) & 0x5555555555555555ull;
    t = (t | (t >> 1)) & 0x3333333333333333ull;
    t = (t | (t >> 2)) & 0x0f0f0f0f0f0f0f0full;
    t = (t | (t >> 4)) & 0x00ff00ff00ff00ffull;
    t = (t | (t >> 8)) & 0x0000ffff0000ffffull;
    t = (t | (t >> 16)) & 0x00000000ffffffffull;
    return t;
}

// `bits` = 16 consecutive sites of a row-major map of width W starting at linear site p0 -> OR into its bit rows (WW words per row)
__device__ __forceinline__ void rows_or16(u64 *rows, int W, int WW, int p0, unsigned bits)
{
    if (!bits) return;
    int y = p0 / W, x = p0 - y * W, left = 16;
    while (bits) {
        const int run = min(left, W - x);
        const unsigned piece = bits & ((1u << run) - 1u);
        if (piece) {
            const int wi = y * WW + (x >> 6), sh = x & 63;
            atomicOr(reinterpret_cast<unsigned long long *>(rows + wi), (unsigned long long)piece << sh);
            if (sh + run > 64) atomicOr(reinterpret_cast<unsigned long long *>(rows + wi + 1), (unsigned long long)piece >> (64 - sh));
        }
        bits >>= run; left -= run; x = 0; ++y;
    }
}
// ... and back: the 16 sites from linear site p0 on, out of the bit rows of an H x W map
__device__ __forceinline__ unsigned rows_get16(const u64 *rows, int H, int W, int WW, int p0)
{
    int y = p0 / W, x = p0 - y * W, done = 0;
    unsigned out = 0;
    while (done < 16 && y < H) {
        const int run = min(16 - done, W - x);
        const int wi = y * WW + (x >> 6), sh = x & 63;
        u64 w = rows[wi] >> sh;
        if (sh + run > 64) w |= rows[wi + 1] << (64 - sh);
        out |= ((unsigned)w & ((1u << run) - 1u)) << done;
        done += run; x = 0; ++y;
    }
    return out;
}
// word j of (row << k): bit b of the result = bit 64 j + b - k of the row (nw words; out-of-range bits are 0)
__device__ __forceinline__ u64 row_shifted_word(const u64 *row, int nw, int j, int k)
{
    const int o = 64 * j - k;              // source bit of the result's bit 0
    const int wi = o >> 6, sh = o & 63;    // (arithmetic shift: floor)
    const u64 lo = (wi >= 0 && wi < nw) ? row[wi] : 0ull;
    const u64 hi = (wi + 1 >= 0 && wi + 1 < nw) ? row[wi + 1] : 0ull;
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}
// the same out of three consecutive words in registers (w[0] = word j - 1, w[1] = word j, w[2] = word j + 1), |k| < 64
__device__ __forceinline__ u64 shifted3(const u64 (&w)[3], int k)
{
    if (k == 0) return w[1];
    if (k > 0) return (w[1] << k) | (w[0] >> (64 - k));
    return (w[1] >> -k) | (w[2] << (64 + k));
}

struct EmitCtx { int *sc; int *pre; int k; }; // scan scratch, per-word offset table, scan counter

// Ordered emission of the set bits of a sequence of nseg words: get(seg, bits, pk0) yields the word and the packed (y, x) of its bit
// 0; entry = pk0 + bit.  Pass 1: a thread takes a contiguous run of words, one block scan, the words' offsets into the table; pass 2:
// half words round robin.  Entries below index mir_cap (counted from the list's start; this call's first entry has index mir_base) are
// mirrored into LDS.  Returns the number of entries (all threads).
template <int NT, class Get>
__device__ __forceinline__ int emit_ordered(int nseg, Get get, int *out, int *mir, int mir_cap, int mir_base, EmitCtx &cx)
{
    const int tid = threadIdx.x;
    if (nseg <= 64) { // a small map (every deep layer of the e-fcn): ONE wave, a word per lane, a wave scan -- no block scan, no offset table
        int *my = cx.sc + (cx.k & 1) * (NT / 64);
        ++cx.k;
        if (tid < 64) {
            u64 b = 0; int pk0 = 0;
            if (tid < nseg) get(tid, b, pk0);
            const int cnt = __popcll(b);
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(inc, d, 64);
                if (tid >= d) inc += t;
            }
            int p = inc - cnt;
            while (b) {
                const int pk = pk0 + __builtin_ctzll(b);
                out[p] = pk;
                if (mir_base + p < mir_cap) mir[mir_base + p] = pk;
                ++p;
                b &= b - 1;
            }
            if (tid == 63) my[0] = inc;
        }
        lds_barrier();
        return my[0];
    }
    const int per = (nseg + NT - 1) / NT;
    const int g0 = min(tid * per, nseg), g1 = min(g0 + per, nseg);
    int cnt = 0;
    for (int g = g0; g < g1; ++g) { u64 b; int pk0; get(g, b, pk0); cnt += __popcll(b); }
    int tot;
    int pos = block_excl_scan_lds<NT>(cnt, cx.sc, cx.k, tot);
    for (int g = g0; g < g1; ++g) { u64 b; int pk0; get(g, b, pk0); cx.pre[g] = pos; pos += __popcll(b); }
    lds_barrier();
    if (tot)
        for (int h = tid; h < 2 * nseg; h += NT) {
            u64 b; int pk0;
            get(h >> 1, b, pk0);
            unsigned w = (h & 1) ? (unsigned)(b >> 32) : (unsigned)b;
            if (!w) continue;
            int p = cx.pre[h >> 1] + ((h & 1) ? __popc((unsigned)b) : 0);
            pk0 += (h & 1) * 32;
            while (w) {
                const int pk = pk0 + __builtin_ctz(w);
                out[p] = pk;
                if (mir_base + p < mir_cap) mir[mir_base + p] = pk;
                ++p;
                w &= w - 1;
            }
        }
    lds_barrier(); // (the table is reused by the next emission)
    return tot;
}

// pool_eval_vec4_items (aev_kernels.hpp) with NB work items per thread in flight: NB x 8 16-byte loads + NB argmax words before the
// first use.  Same arithmetic, same stores.
template <int NB>
__device__ __forceinline__ void pool_eval_vec4_batched(const PoolArgs &a, int s, int n, int nin, const int *list, const int *list_l, int cap_l,
                                                       int first, int stride)
{
    const int C4 = a.C >> 2;
    const int total = n * C4;
    if (total <= 0) return;
    const float al = a.src.alpha;
    const bool pl = a.plain != 0;
    const long long orow = (long long)a.src.bW * a.C;
    const float *sb = a.src.s + (long long)s * a.src.bstride, *ab = a.src.a + (long long)s * a.src.bstride;
    for (int e0 = first; e0 < total; e0 += stride * NB) {
        f32x4 sv[NB][4], av[NB][4];
        unsigned old[NB];
        int wsite[NB], cc[NB], jj[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { // (clamped, unguarded)
            const int e = min(e0 + b * stride, total - 1);
            const int j = e / C4, c = (e - j * C4) << 2;
            const int pk = j < cap_l ? list_l[j] : list[j];
            const int wy = unpack_y(pk), wx = unpack_x(pk);
            const long long o = ((long long)(2 * wy) * a.src.bW + 2 * wx) * a.C + c;
            sv[b][0] = *reinterpret_cast<const f32x4 *>(sb + o);              av[b][0] = *reinterpret_cast<const f32x4 *>(ab + o);
            sv[b][1] = *reinterpret_cast<const f32x4 *>(sb + o + a.C);        av[b][1] = *reinterpret_cast<const f32x4 *>(ab + o + a.C);
            sv[b][2] = *reinterpret_cast<const f32x4 *>(sb + o + orow);       av[b][2] = *reinterpret_cast<const f32x4 *>(ab + o + orow);
            sv[b][3] = *reinterpret_cast<const f32x4 *>(sb + o + orow + a.C); av[b][3] = *reinterpret_cast<const f32x4 *>(ab + o + orow + a.C);
            wsite[b] = wy * a.W + wx; cc[b] = c; jj[b] = j;
            old[b] = *reinterpret_cast<const unsigned *>(a.idx + (long long)s * a.istride + (long long)wsite[b] * a.C + c);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (e0 + b * stride >= total) continue;
            unsigned packed = 0;
            bool flag = false;
            f32x4 wf, wa;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                PoolPick pp;
                pp.first(sv[b][0][u], pl ? 0.0f : __fmul_rn(av[b][0][u], sv[b][0][u] > 0.0f ? 1.0f : al), av[b][0][u]);
#pragma unroll
                for (int r = 1; r < 4; ++r) pp.next(r, sv[b][r][u], pl ? 0.0f : __fmul_rn(av[b][r][u], sv[b][r][u] > 0.0f ? 1.0f : al), av[b][r][u]);
                packed |= (unsigned)pp.best << (8 * u);
                flag |= pp.bd != pp.slow;
                wf[u] = pp.bs; wa[u] = pp.ba;
            }
            const long long po = (long long)s * a.istride + (long long)wsite[b] * a.C + cc[b];
            if (!pl && flag) a.pending[(long long)s * a.mstride + wsite[b]] = 1;                                 // maxpool.py:140-142
            if (a.pfm && nin >= 0 && jj[b] >= nin && old[b] == packed) continue;                                 // pending-only, same winners: nothing to rewrite
            *reinterpret_cast<unsigned *>(a.idx + po) = packed;                                                  // maxpool.py:145-151
            if (a.pfm) {
                *reinterpret_cast<f32x4 *>(a.pfm + po) = wf; *reinterpret_cast<f32x4 *>(a.pact + po) = wa;     // (the copy has idx's layout)
                if (a.pflag && cc[b] == 0) a.pflag[(long long)s * a.mstride + wsite[b]] = 1;
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void post2_body(const PostArgs &a, const Post2Lay &ly, u64 *dyn, int *sc)
{
    const int s = blockIdx.x + a.s0;
    const int tid = threadIdx.x;
    unsigned long long t_last = ly.dbg ? __builtin_amdgcn_s_memrealtime() : 0, tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int ph) __attribute__((always_inline)) {
        if (ly.dbg) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); tph[ph] += t - t_last; t_last = t; }
    };
    if (a.advance && blockIdx.x == 0 && tid == 0) { *a.advance += 1; *a.advance_stat += 1; }
    u64 *E = dyn + ly.o_E, *Top = dyn + ly.o_top, *Bot = dyn + ly.o_bot, *Pend = dyn + ly.o_pend, *Src = dyn + ly.o_src, *T = dyn + ly.o_T;
    u64 *tile_mask = dyn + ly.o_tmask;
    int *plm = reinterpret_cast<int *>(dyn) + ly.o_plm;
    EmitCtx cx{sc, reinterpret_cast<int *>(dyn) + ly.o_pre, 0};
    const int H = a.HW / a.W, W = a.W, WW = ly.WW;
    // ---- ONE trip to memory: the stream's activity, this layer's touch map A and sign-change map B, the pool layer's pending map ----
    // A stream without events in the step gets no call (its maps are clear already, its pending windows stay pending): its bits are
    // masked to zero instead of branching in front of the loads -- the same empty lists come out.
    const int act = a.active ? a.active[s] : 1;
    uint8_t *A = a.mapA + (long long)s * a.mstride;
    uint8_t *B = a.mapB ? a.mapB + (long long)s * a.mstride : A;
    const bool hasB = a.mapB != nullptr;
    const int nchunks = (a.HW + 15) >> 4; // (mstride is a multiple of 64 and the pad bytes stay 0)
    constexpr int UB = (2304 + NT - 1) / NT, UP = (768 + NT - 1) / NT; // chunks of 16 sites per thread and pass: the e-fcn's largest map (160 x 224 = 2240 chunks) is one pass
    ulonglong2 va[UB], vb[UB], vp[UP];
#pragma unroll
    for (int k = 0; k < UB; ++k) { // (clamped, unguarded: all loads are in flight together)
        const int c = min(tid + k * NT, nchunks - 1);
        va[k] = *reinterpret_cast<const ulonglong2 *>(A + 16 * c);
        vb[k] = *reinterpret_cast<const ulonglong2 *>(B + 16 * c);
    }
    const int pn = a.has_pool ? (a.pg.oh * a.pg.ow + 15) >> 4 : 0;
    const uint8_t *P = a.has_pool ? a.ppending + (long long)s * a.pmstride : A;
#pragma unroll
    for (int k = 0; k < UP; ++k) vp[k] = *reinterpret_cast<const ulonglong2 *>(P + 16 * min(tid + k * NT, max(pn, 1) - 1));
    for (int t = tid; t < ly.words; t += NT) dyn[t] = 0;
    lds_barrier();
    stamp(0);
    const unsigned am = act ? 0xffffu : 0u;
    uint8_t *Fl = a.dflag ? a.dflag + (long long)s * a.mstride : nullptr;
    int cntA = 0;
    auto chunk = [&](int c, const ulonglong2 &xa, const ulonglong2 &xb) __attribute__((always_inline)) {
        const unsigned ab = pack16(xa) & am, bb = hasB ? pack16(xb) & am : 0u;
        cntA += __popc(ab);
        const ulonglong2 z = {0ull, 0ull};
        if (ab) {
            *reinterpret_cast<ulonglong2 *>(A + 16 * c) = z;
            if (Fl) { // the recomputed sites may decay (SweepSeg::flag): byte stores, no read-modify-write of the flag map
                unsigned w = ab;
                while (w) { Fl[16 * c + __builtin_ctz(w)] = 1; w &= w - 1; }
            }
        }
        if (bb) *reinterpret_cast<ulonglong2 *>(B + 16 * c) = z;
        rows_or16(E, W, WW, 16 * c, ab | bb);
    };
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int c = tid + k * NT;
        if (c < nchunks) chunk(c, va[k], vb[k]);
    }
    for (int c = tid + UB * NT; c < nchunks; c += NT) // (maps beyond UB * NT * 16 sites: the rest, a chunk at a time)
        chunk(c, *reinterpret_cast<const ulonglong2 *>(A + 16 * c), *reinterpret_cast<const ulonglong2 *>(B + 16 * c));
    if (a.has_pool) {
#pragma unroll
        for (int k = 0; k < UP; ++k) {
            const int c = tid + k * NT;
            if (c < pn) rows_or16(Pend, a.pg.ow, ly.PWW, 16 * c, pack16(vp[k]) & am);
        }
        for (int c = tid + UP * NT; c < pn; c += NT) rows_or16(Pend, a.pg.ow, ly.PWW, 16 * c, pack16(*reinterpret_cast<const ulonglong2 *>(P + 16 * c)) & am);
    }
    stamp(1); // maps loaded, cleared, bits in LDS (this thread's share)
    if (a.count_touch || a.statA) { // block-wide popcount of A
        int tot;
        block_excl_scan_lds<NT>(cntA, cx.sc, cx.k, tot);
        cntA = tot;
    }
    lds_barrier();
    stamp(2);
    // ---- A: the event list ----------------------------------------------------------------------
    {
        int *ev = a.ev + (long long)s * a.lstride;
        const int nev = emit_ordered<NT>(H * WW, [&](int g, u64 &b, int &pk0) __attribute__((always_inline)) {
            const int y = g / WW, j = g - y * WW;
            b = E[g]; pk0 = pack_yx(y, 64 * j);
        }, ev, nullptr, 0, 0, cx);
        if (tid == 0) {
            a.nev[s] = nev;
            const int st = a.count_touch ? cntA : nev;
            if (a.statA && st) atomicAdd(a.statA, (u64)st);
        }
    }
    stamp(3);
    // ---- B: the pool layer's windows (2x2 / stride 2) ---------------------------------------------------
    int npl = 0, nin_w = 0;
    if (a.has_pool) {
        const int oh = a.pg.oh, ow = a.pg.ow, PWW = ly.PWW;
        uint8_t *pend = a.ppending + (long long)s * a.pmstride;
        for (int q = tid; q < oh * PWW; q += NT) {
            const int wy = q / PWW, j = q - wy * PWW;
            const u64 *r0 = E + (2 * wy) * WW, *r1 = r0 + WW;
            const u64 t0 = pair_or_compress(r0[2 * j]) | (2 * j + 1 < WW ? pair_or_compress(r0[2 * j + 1]) << 32 : 0ull);
            const u64 t1 = pair_or_compress(r1[2 * j]) | (2 * j + 1 < WW ? pair_or_compress(r1[2 * j + 1]) << 32 : 0ull);
            const u64 in = t0 | t1, pb = Pend[q];
            u64 clr = pb & in; // incoming windows are removed from the recompute list (maxpool.py:118-120)
            while (clr) {
                pend[wy * ow + 64 * j + __builtin_ctzll(clr)] = 0;
                clr &= clr - 1;
            }
            Top[q] = t0; Bot[q] = t1 & ~t0; Pend[q] = pb & ~in;
            Src[q] = in | pb;
        }
        lds_barrier();
        stamp(4);
        int *plist = a.plist + (long long)s * a.plstride;
        nin_w = emit_ordered<NT>(2 * oh * PWW, [&](int g, u64 &b, int &pk0) __attribute__((always_inline)) {
            const int wy = g / (2 * PWW), r = g - wy * 2 * PWW, cls = r >= PWW ? 1 : 0, j = r - cls * PWW;
            b = (cls ? Bot : Top)[wy * PWW + j]; pk0 = pack_yx(wy, 64 * j);
        }, plist, plm, ly.plm_cap, 0, cx);
        const int npend = emit_ordered<NT>(oh * PWW, [&](int g, u64 &b, int &pk0) __attribute__((always_inline)) {
            const int wy = g / PWW, j = g - wy * PWW;
            b = Pend[g]; pk0 = pack_yx(wy, 64 * j);
        }, plist + nin_w, plm, ly.plm_cap, nin_w, cx);
        npl = nin_w + npend;
        if (tid == 0) {
            if (a.pnevw) a.pnevw[s] = nin_w;
            a.pnlist[s] = npl;
            if (a.statP && npl) atomicAdd(a.statP, (u64)npl);
        }
    }
    stamp(5);
    // ---- C: the consumer's touch map, tile list, site list ----------------------------------------------
    if (a.has_next) {
        const u64 *S = a.has_pool ? Src : E;
        const int sH = ly.sH, SWW = ly.SWW, TWW = ly.TWW, nH = a.n_H, nW = a.n_W;
        for (int q = tid; q < nH * TWW; q += NT) { // separable: OR over the window's rows, then over its columns (shifts)
            const int oy = q / TWW, j = q - oy * TWW;
            u64 v[3] = {0ull, 0ull, 0ull};
            for (int dy = 0; dy < a.n_kh; ++dy) {
                const int r = oy - a.n_pt + dy; // source row whose tap dy lands on target row oy (oy = r + pt - dy)
                if (r < 0 || r >= sH) continue;
                const u64 *row = S + r * SWW;
                if (j - 1 >= 0 && j - 1 < SWW) v[0] |= row[j - 1];
                if (j < SWW) v[1] |= row[j];
                if (j + 1 < SWW) v[2] |= row[j + 1];
            }
            u64 t = 0;
            for (int dx = 0; dx < a.n_kw; ++dx) t |= shifted3(v, a.n_pl - dx); // target x = source x + pl - dx
            const int valid = nW - 64 * j;
            if (valid < 64) t &= (1ull << valid) - 1ull;
            T[q] = t;
        }
        lds_barrier();
        stamp(6);
        // the consumer's active tiles with their 256-bit site masks (bit = ly * TW + lx): masks first, so that the list position can be
        // requested (one atomic round trip) while the touch bytes and the site list are written
        const bool tiles = a.n_tlist != nullptr, tiles_one = tiles && a.n_tiles <= NT;
        u64 m[4] = {0, 0, 0, 0};
        int t_base = 0, t_bk = 0;
        u64 t_mine = 0;
        bool t_on = false;
        const int ln = tid & 63;
        if (tiles) {
            const int items = a.n_tiles * a.n_TH;
            for (int it = tid; it < items; it += NT) {
                const int t = it / a.n_TH, lyy = it - t * a.n_TH;
                const int ty = t / a.n_tilesX, tx = t - ty * a.n_tilesX;
                const int y = ty * a.n_TH + lyy, x0 = tx * a.n_TW;
                if (y >= nH) continue;
                const int wdt = min(a.n_TW, nW - x0);
                const u64 bits = row_shifted_word(T + y * TWW, TWW, 0, -x0) & ((1ull << wdt) - 1ull); // (TW <= 14)
                if (!bits) continue;
                const int q0 = lyy * a.n_TW, wi = t * 4 + (q0 >> 6), sh = q0 & 63;
                atomicOr(reinterpret_cast<unsigned long long *>(tile_mask + wi), (unsigned long long)bits << sh);
                if (sh + wdt > 64) atomicOr(reinterpret_cast<unsigned long long *>(tile_mask + wi + 1), (unsigned long long)bits >> (64 - sh));
            }
            lds_barrier();
            stamp(8);
            if (tiles_one) { // tile_list_append's reservation (aev_kernels.hpp), the result used below
                const int t = tid;
                if (t < a.n_tiles) { m[0] = tile_mask[t * 4]; m[1] = tile_mask[t * 4 + 1]; m[2] = tile_mask[t * 4 + 2]; m[3] = tile_mask[t * 4 + 3]; }
                const int nact = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
                t_on = nact != 0;
                const int nb = a.n_nb;
                t_bk = min(nb - 1, max(nact - 1, 0) * nb / (a.n_TH * a.n_TW)); // segment by fill
                const u64 bal = __ballot(t_on);
                t_mine = bal;
                int cnt = ln == 32 || nb == 1 ? __popcll(bal) : 0; // lane b: entries for segment b; lane 32: all
                if (nb > 1 && bal) {
#pragma unroll 1
                    for (int b = 0; b < nb; ++b) {
                        const u64 bb = __ballot(t_on && t_bk == b);
                        if (t_bk == b) t_mine = bb;
                        if (ln == b) cnt = __popcll(bb);
                    }
                }
                if ((ln < nb || ln == 32) && cnt) t_base = atomicAdd(a.n_tcount + (ln < nb ? 4 + ln : 0), cnt);
            }
        }
        uint8_t *Tb = a.n_touch + (long long)s * a.n_mstride;
        const int tn = (nH * nW + 15) >> 4;
        for (int c = tid; c < tn; c += NT) {
            const unsigned b = rows_get16(T, nH, nW, TWW, 16 * c);
            if (b) { ulonglong2 v; v.x = spread8(b); v.y = spread8(b >> 8); *reinterpret_cast<ulonglong2 *>(Tb + 16 * c) = v; }
        }
        stamp(7);
        if (a.n_sites) { // list-driven consumer: its touched sites, row-major
            const int n = emit_ordered<NT>(nH * TWW, [&](int g, u64 &b, int &pk0) __attribute__((always_inline)) {
                const int y = g / TWW, j = g - y * TWW;
                b = T[g]; pk0 = pack_yx(y, 64 * j);
            }, a.n_sites + (long long)s * a.n_lstride, nullptr, 0, 0, cx);
            if (tid == 0) a.n_nsites[s] = n;
            stamp(10);
        }
        if (tiles_one) {
            const int base = __shfl(t_base, t_bk);
            if (t_on) {
                const long long pos = (long long)t_bk * a.n_cap + base + __popcll(t_mine & ((1ull << ln) - 1));
                a.n_tlist[pos] = blockIdx.x * a.n_tiles + tid;
                a.n_tmask[pos * 4] = m[0]; a.n_tmask[pos * 4 + 1] = m[1]; a.n_tmask[pos * 4 + 2] = m[2]; a.n_tmask[pos * 4 + 3] = m[3];
            }
        } else if (tiles)
            tile_list_append<NT>(tile_mask, a.n_tiles, blockIdx.x, a.n_tlist, a.n_tmask, a.n_tcount, a.n_cap, a.n_TH * a.n_TW, a.n_nb);
        if (tiles) stamp(9);
    }
    if (a.prev_ts && tid == 0 && act) a.prev_ts[s] = a.last_ts[s];
    // ---- D: a small pool layer is evaluated in place (the windows listed above) --------------------------
    if (a.has_pool && a.pe_inline) {
        // A FULL barrier: the evaluation re-flags windows whose pending bytes stage B has just cleared with stores of other waves (two
        // stores to one byte must not overtake each other), and reads the list's tail beyond the mirror back from global memory
        __syncthreads();
        pool_eval_vec4_batched<2>(a.pe, s, npl, nin_w, a.plist + (long long)s * a.plstride, plm, ly.plm_cap, tid, NT);
        stamp(11);
    }
    if (ly.dbg && blockIdx.x == 0 && tid == 0) {
        for (int q = 0; q < 12; ++q) ly.dbg[q] += tph[q];
        ly.dbg[15] += 1;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_post2(PostArgs a, Post2Lay ly)
{
    __shared__ int sc[2 * NT / 64];
    extern __shared__ u64 post2_dyn[];
    post2_body<NT>(a, ly, post2_dyn, sc);
}
// the input layer's and the first conv layer's bookkeeping in one launch (as k_post_pair): the second body reads the touch bytes the
// first one has stored (same block: a full barrier between the two)
template <int NT>
__global__ __launch_bounds__(NT) void k_post2_pair(PostArgs a, Post2Lay la, PostArgs b, Post2Lay lb)
{
    __shared__ int sc[2 * NT / 64];
    extern __shared__ u64 post2_dyn[];
    post2_body<NT>(a, la, post2_dyn, sc);
    __threadfence_block();
    __syncthreads();
    post2_body<NT>(b, lb, post2_dyn, sc);
}

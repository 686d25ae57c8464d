// aev_conv_x6.hpp -- the spatial-tile kernel on bf16 MFMAs with three-way split f32 operands (round 5; OPT-IN: aev_net_set_conv_math).
// Included at the end of aev_kernels.hpp, behind aev_conv_patch.hpp (PatchArgs, the tile lists).
//
// The f32 MFMA runs at 256 flop per cycle and CU, the bf16 MFMA of gfx950 (v_mfma_f32_16x16x32_bf16) at 4 096.  An f32 value is EXACTLY
// the sum of three bf16 values (its 24-bit significand cut into 8 + 8 + 8 bits: hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000,
// lo = x - hi - mid), and the product of two such sums, without the three terms below 2^-24 of it, is six bf16 products:
//      a b  ~  ah bh + (ah bm + am bh) + (ah bl + al bh + am bm)
// accumulated in f32 inside the MFMA.  tools/probes/mfma_bf16x6.hip: 2.4x the f32 MFMA rate, error against float64 over a K = 1 152
// contraction rms 5.2e-7 / max 4.6e-6 -- the float32 fmaf chain of the arithmetic contract has 6.1e-7 / 5.3e-6.  What it is NOT is
// bit-identical to that chain (4 % of the outputs are): with this kernel the float maps agree with the fmaf-chain reference of the tests to north_star's 1e-5 (sampled streams; a million outputs: max 1.6e-5)
// instead of to the bit, and a sign decision on a value within ~1e-6 of zero may fall the other way.  Operands that are small
// integers / dyadic (the exact fixture family) split exactly and accumulate exactly: index bookkeeping stays bit-exact there.
//
// Shape of the computation:
//   * k order of a chunk of 16 input channels: five k-blocks of 32 = (two taps) x (16 channels); the tap pairs are (0,1) (3,4) (6,7) --
//     neighbours in x --, (2,5) -- neighbours in y --, (8, none).  Lane group g = lane / 16 of the MFMA holds the eight channels 8 (g & 1) ..
//     of tap g >> 1 of the pair: ONE 16-byte read per operand part.  (9 taps in 10 slots: 10 % of the MFMA cycles multiply zeros.)
//   * LDS image of a chunk (two buffers of 16 (TH + 2) slots; ONE where the layer has a single chunk): per patch slot (py * 16 + px) six runs of 16 bf16 -- feature hi / mid / lo, decay hi / mid /
//     lo --, slot stride 208 B (13 x 16: the 16 lanes of a read pass hit 16 different bank groups).  The split happens ONCE per staged value,
//     when the chunk is written to LDS (inside the k loop it would be VALU-bound).
//   * weights: pre-split on the host, [part][chunk][k-block][cout][lane group][8 channels] bf16; a lane reads its A operands straight from
//     global memory (L2: <= 2 MB per layer, shared by all blocks), one k-block ahead, and uses each for all its site blocks.
//   * a wave owns 64 couts (four 16-cout halves; NH = 2: 32) x up to NAMAX site blocks of 16 sites x 2 maps: every value operand read from LDS feeds four
//     MFMAs per product (the 6x operand traffic of the split would otherwise make the kernel LDS-bound).  Block = 4 waves = NSG site groups x
//     NCG cout groups.
//   * persistent blocks over the layer's tile list (strided), site table from the tile's 256-bit mask, scatter of the active sites: as
//     k_conv_patch (aev_conv_patch.hpp), without its item pipeline (measured with it: no gain -- docs/KERNEL_NOTES.md, round 5).
//   * instances (aev_hip.hip, patch_kernel): <2,2,5> 128 couts per block, one block per CU; <4,1,2,2,2> 32 couts per block, two blocks per CU
//     (layers of 32 and of 64 couts).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define AEV_X6_SLOT 208          // bytes per patch slot in LDS: 6 parts x 16 channels x 2 B, padded to 13 x 16
#define AEV_X6_BUF (256 * AEV_X6_SLOT) // the largest buffer: a patch of 16 rows (a launch's buffers hold the rows its tiles have: X6Args::slots)
#define AEV_X6_KB 5              // k-blocks (tap pairs) per chunk of 16 channels
// tap (dy * 3 + dx) of pair p, position q: (0,1) (3,4) (6,7) (2,5) (8,-)
__host__ __device__ constexpr int x6_tap(int p, int q) { return p < 3 ? 3 * p + q : p == 3 ? 2 + 3 * q : (q ? -1 : 8); }

struct X6Args {
    PatchArgs p;                 // geometry, source, state, tile list (wt unused)
    const unsigned short *w3;    // split weights [3][cin / 16][AEV_X6_KB][CoutP][4][8] bf16
    int nch;                     // cin / 16
    int slots;                   // 16 (TH + 2): patch slots per LDS buffer (two buffers; ONE where the layer has a single chunk)
};

// 8 consecutive bf16 (16 B) as an MFMA operand
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 &v) { union { u32x4 u; bf16x8 b; } c; c.u = v; return c.b; }

template <int NSG, int NCG, int NAMAX, int NH = 4, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void k_conv_x6(X6Args xa)
{
    static_assert(NSG * NCG == 4, "four waves per block");
    static_assert(NH == 2 || NH == 4, "a wave owns two or four 16-cout halves");
    extern __shared__ float lds[];
    const PatchArgs &a = xa.p;
    const SrcView &v = a.src;
    constexpr int ACAP = 16 * NAMAX * NSG;          // active sites of a tile
    constexpr int CW = 16 * NH * NCG;               // couts of a block
    const int bufb = xa.slots * AEV_X6_SLOT;                                     // bytes per buffer
    unsigned char *img = reinterpret_cast<unsigned char *>(lds);                 // two buffers of `bufb` bytes (one: single-chunk layers)
    int *lsite = reinterpret_cast<int *>(img + (xa.nch > 1 ? 2 : 1) * bufb);     // [ACAP] (ly << 8) | lx of the active sites, row-major
    int *loff = lsite + ACAP;                                                    // [ACAP] element offset of the site in the stream's map (padding rows: the sink)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = wv % NSG, cg = wv / NSG;
    const int g = lane >> 4, tapsel = g >> 1, oct = g & 1;
    const int items = __builtin_amdgcn_readfirstlane(a.tcount[0]) * a.gz;
    const int tw_inv = 65536 / a.TW + 1;
    // list position of the t-th tile (fullest segment first), as k_conv_patch
    int tmap = 0;
    if (a.nb > 1) {
        const int cv = lane < AEV_TL_BUCKETS ? a.tcount[4 + lane] : 0;
        int first = 0;
#pragma unroll 1
        for (int b = a.nb - 1; b > 0; --b)
            if ((lane & 31) < b) first += __builtin_amdgcn_readlane(cv, b);
        tmap = lane < 32 ? first : (lane & 31) * a.cap - first;
    }
    auto lpos = [&](int t) __attribute__((always_inline)) {
        if (a.nb <= 1) return t;
        const int b = __ffsll((long long)__ballot(lane < AEV_TL_BUCKETS && tmap <= t)) - 1;
        return t + __builtin_amdgcn_readlane(tmap, 32 + b);
    };
    const size_t wpart = (size_t)xa.nch * AEV_X6_KB * a.CoutP * 32; // bf16 elements per weight part
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int pos = lpos(item / a.gz), z = item % a.gz;
        const int ent = __builtin_amdgcn_readfirstlane(a.tlist[pos]);
        const int s = ent / a.tiles + a.s0, tile = ent % a.tiles;
        const int ty0 = (tile / a.tilesX) * a.TH, tx0 = (tile % a.tilesX) * a.TW;
        // ---- the tile's site table from its 256-bit mask (every wave builds all of it: the same values to the same addresses) ----
        const unsigned *tm = reinterpret_cast<const unsigned *>(a.tmask) + (long long)pos * 8;
        const unsigned mw = tm[lane & 7];
        __syncthreads(); // (the scatter of the item before has read the table)
        int nact = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 m = (u64)(unsigned)__builtin_amdgcn_readlane((int)mw, 2 * j) | ((u64)(unsigned)__builtin_amdgcn_readlane((int)mw, 2 * j + 1) << 32);
            if ((m >> lane) & 1) {
                const int r = nact + __popcll(m & ((1ull << lane) - 1));
                const int q = lane + 64 * j, ly = (q * tw_inv) >> 16, lx = q - ly * a.TW;
                lsite[r] = (ly << 8) | lx;
                loff[r] = ((ty0 + ly) * a.W + tx0 + lx) * a.C;
            }
            nact += __popcll(m);
        }
        if (lane < ((16 - (nact & 15)) & 15)) loff[nact + lane] = a.HW * a.C; // the rows that pad the last site block: the sink behind the map
        // ---- per-lane patch geometry of the staging: element e = tid + 256 q -> slot e / 4, channels 4 (e % 4) .. + 3 of the chunk ----
        int voff[4];
        unsigned outside = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, slot = e >> 2, py = slot >> 4, px = slot & 15;
            const int iy = ty0 - a.pt + py, ix = tx0 - a.pl + px;
            const bool in = py < a.TH + 2 && px < a.TW + 2 && iy >= 0 && iy < v.H && ix >= 0 && ix < v.W; // zero padding, conv2d.py:68-72
            if (!in) outside |= 1u << q;
            voff[q] = (min(max(iy, 0), v.H - 1) * v.bW + min(max(ix, 0), v.W - 1)) * v.C + 4 * (e & 3);
        }
        const float *s_ptr = v.s + (long long)s * v.bstride, *a_ptr = v.a + (long long)s * v.bstride;
        // (the four staging elements of a thread in two halves: two elements' values -- 16 registers -- are live at a time, requested and
        //  committed between the k-blocks of the chunk before)
        f32x4 sv[2], av[2];
        auto stage_load = [&](int c, int half) __attribute__((always_inline)) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                if (tid + 256 * (2 * half + q2) < 4 * xa.slots) { sv[q2] = ld_u32x4(s_ptr + 16 * c, (unsigned)voff[2 * half + q2]); av[q2] = ld_u32x4(a_ptr + 16 * c, (unsigned)voff[2 * half + q2]); }
        };
        auto stage_store = [&](int buf, int half) __attribute__((always_inline)) { // activation, three-way split, LDS
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int q = 2 * half + q2;
                const int e = tid + 256 * q;
                if (e >= 4 * xa.slots) continue; // (behind the patch rows of this launch)
                unsigned char *dst = img + buf * bufb + (e >> 2) * AEV_X6_SLOT + (e & 3) * 8;
                const unsigned keep = ((outside >> q) & 1) ? 0u : 0xffffffffu; // zero padding: the values of an out-of-map slot are masked to +0
                unsigned fh[4], fm[4], fl[4], dh[4], dm[4], dl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float x = sv[q2][u], y = av[q2][u];
                    const float L = x > 0.0f ? 1.0f : v.alpha;                                           // conv2d.py:85-88
                    const float F = __uint_as_float(__float_as_uint(__fmul_rn(x, L)) & keep);            // layer.py:80
                    const float D = __uint_as_float(__float_as_uint(__fmul_rn(y, L)) & keep);            // conv2d.py:93
                    fh[u] = __float_as_uint(F) & 0xffff0000u;
                    const float fr = __fsub_rn(F, __uint_as_float(fh[u]));
                    fm[u] = __float_as_uint(fr) & 0xffff0000u;
                    fl[u] = __float_as_uint(__fsub_rn(fr, __uint_as_float(fm[u])));                     // (at most 8 significant bits: a bf16 already)
                    dh[u] = __float_as_uint(D) & 0xffff0000u;
                    const float dr = __fsub_rn(D, __uint_as_float(dh[u]));
                    dm[u] = __float_as_uint(dr) & 0xffff0000u;
                    dl[u] = __float_as_uint(__fsub_rn(dr, __uint_as_float(dm[u])));
                }
                // two bf16 (the upper halves of two f32 patterns) in one dword: bytes 2, 3 of `lo` and of `hi` -- one v_perm_b32
                auto pk = [](unsigned lo, unsigned hi) __attribute__((always_inline)) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };
                *reinterpret_cast<u32x2 *>(dst + 0) = (u32x2){pk(fh[0], fh[1]), pk(fh[2], fh[3])};
                *reinterpret_cast<u32x2 *>(dst + 32) = (u32x2){pk(fm[0], fm[1]), pk(fm[2], fm[3])};
                *reinterpret_cast<u32x2 *>(dst + 64) = (u32x2){pk(fl[0], fl[1]), pk(fl[2], fl[3])};
                *reinterpret_cast<u32x2 *>(dst + 96) = (u32x2){pk(dh[0], dh[1]), pk(dh[2], dh[3])};
                *reinterpret_cast<u32x2 *>(dst + 128) = (u32x2){pk(dm[0], dm[1]), pk(dm[2], dm[3])};
                *reinterpret_cast<u32x2 *>(dst + 160) = (u32x2){pk(dl[0], dl[1]), pk(dl[2], dl[3])};
            }
        };
        stage_load(0, 0);
        __syncthreads(); // table complete
        // this lane's site of each of the wave's site blocks: byte offsets of its operand in a buffer for the x pairs, the y pair, the last tap
        const int nsb = (nact + 15) >> 4;
        const int mine = nsb > sg ? (nsb - sg + NSG - 1) / NSG : 0; // site blocks sg, sg + NSG, ... (wave-uniform)
        int bx1[NAMAX]; // taps (3p, 3p + 1): + 16 p slots; the y pair (2, 5): + off_y; tap 8 (lane groups 2, 3: zero weights): + off_8
        const int off_y = (2 + 15 * tapsel) * AEV_X6_SLOT, off_8 = (34 - tapsel) * AEV_X6_SLOT;
#pragma unroll
        for (int i = 0; i < NAMAX; ++i) {
            const int j = (sg + i * NSG) * 16 + (lane & 15);
            const int ls = (i < mine && j < nact) ? lsite[j] : 0;
            const int sb = (ls >> 8) * 16 + (ls & 0xff);
            bx1[i] = (sb + tapsel) * AEV_X6_SLOT + oct * 16;
        }
        const int cout0 = z * CW + cg * 16 * NH; // the wave's first cout
        f32x4 accf[NAMAX][NH], accd[NAMAX][NH];
#pragma unroll
        for (int i = 0; i < NAMAX; ++i)
#pragma unroll
            for (int h = 0; h < NH; ++h) { accf[i][h] = (f32x4){0.f, 0.f, 0.f, 0.f}; accd[i][h] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        // weights of k-block kb for this lane: [part][kb][cout][g][8]
        const unsigned short *wl = xa.w3 + ((size_t)(cout0 + (lane & 15)) * 4 + g) * 8;
        // two register sets for the weights, used alternately without copies: k-block kb lives in set kb & 1 (five k-blocks per chunk: the
        // chunk's parity decides which set its first k-block finds its weights in -- a compile-time constant inside `chunk`)
        u32x4 w0[3][NH], w1[3][NH];
        auto wload = [&](int kb, u32x4 (&w)[3][NH]) __attribute__((always_inline)) {
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    w[part][h] = *reinterpret_cast<const u32x4 *>(wl + part * wpart + ((size_t)kb * a.CoutP + 16 * h) * 32);
        };
        const int nkb = xa.nch * AEV_X6_KB;
        auto kblock = [&](const unsigned char *B, int kb, int p, u32x4 (&w)[3][NH], u32x4 (&wn)[3][NH]) __attribute__((always_inline)) {
            wload(min(kb + 1, nkb - 1), wn);
            const int poff = p < 3 ? 16 * p * AEV_X6_SLOT : p == 3 ? off_y : off_8;
#pragma unroll
            for (int i = 0; i < NAMAX; ++i) {
                if (i < mine) {
                    const u32x4 *bp = reinterpret_cast<const u32x4 *>(B + bx1[i] + poff);
                    const bf16x8 fh = as_bf16x8(bp[0]), fm = as_bf16x8(bp[2]), fl = as_bf16x8(bp[4]);
                    const bf16x8 dh = as_bf16x8(bp[6]), dm = as_bf16x8(bp[8]), dl = as_bf16x8(bp[10]);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const bf16x8 wh = as_bf16x8(w[0][h]), wm = as_bf16x8(w[1][h]), wlo = as_bf16x8(w[2][h]);
                        // small terms first: (wm fm + wl fh + wh fl) + (wm fh + wh fm) + wh fh
                        accf[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, fm, accf[i][h], 0, 0, 0);
                        accd[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, dm, accd[i][h], 0, 0, 0);
                        accf[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, fh, accf[i][h], 0, 0, 0);
                        accd[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, dh, accd[i][h], 0, 0, 0);
                        accf[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, fl, accf[i][h], 0, 0, 0);
                        accd[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dl, accd[i][h], 0, 0, 0);
                        accf[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, fh, accf[i][h], 0, 0, 0);
                        accd[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, dh, accd[i][h], 0, 0, 0);
                        accf[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, fm, accf[i][h], 0, 0, 0);
                        accd[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dm, accd[i][h], 0, 0, 0);
                        accf[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, fh, accf[i][h], 0, 0, 0);
                        accd[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dh, accd[i][h], 0, 0, 0);
                    }
                }
            }
        };
        auto chunk = [&](int c, auto par_t) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_t)::value;
            const bool more = c + 1 < xa.nch;
            if (more) stage_load(c + 1, 0);
            const unsigned char *B = img + (c & 1) * bufb;
#pragma unroll
            for (int p = 0; p < AEV_X6_KB; ++p) {
                if (mine > 0) {
                    if (((PAR + p) & 1) == 0) kblock(B, c * AEV_X6_KB + p, p, w0, w1);
                    else kblock(B, c * AEV_X6_KB + p, p, w1, w0);
                }
                if (p == 1 && more) { stage_store((c & 1) ^ 1, 0); stage_load(c + 1, 1); } // (the first half has landed; the second is requested)
            }
            if (more) stage_store((c & 1) ^ 1, 1);
            __syncthreads(); // the other buffer is complete, and every wave is done reading this one
        };
        wload(0, w0);
        stage_store(0, 0);
        stage_load(0, 1);
        stage_store(0, 1); // (a block's every item starts exposed: two trips)
        __syncthreads();
        for (int c = 0; c < xa.nch; c += 2) {
            chunk(c, std::integral_constant<int, 0>{});
            if (c + 1 < xa.nch) chunk(c + 1, std::integral_constant<int, 1>{});
        }
        // ---- scatter the active sites (C / D layout of the 16x16 MFMA with the couts on the rows: register r of lane l = cout 4 (l / 16) + r
        //      of the 16-cout half, site l % 16 of the block: four consecutive couts of one site, one 16-byte store per map) ----
        if (mine > 0) {
            float *fm_ = a.fm + (long long)s * a.sstride, *act_ = a.act + (long long)s * a.sstride;
#pragma unroll
            for (int i = 0; i < NAMAX; ++i) {
                if (i < mine) {
                    const unsigned ro = (unsigned)loff[(sg + i * NSG) * 16 + (lane & 15)];
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const int co = cout0 + 16 * h + 4 * (lane >> 4);
                        if (co < a.C) { // (C % 4 == 0)
                            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + co);
                            f32x4 vf;
#pragma unroll
                            for (int r = 0; r < 4; ++r) vf[r] = __fadd_rn(accf[i][h][r], b4[r]);               // conv2d.py:176-178,120
                            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(fm_) + (size_t)((ro + (unsigned)co) * 4u)) = vf;
                            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(act_) + (size_t)((ro + (unsigned)co) * 4u)) = accd[i][h]; // conv2d.py:121-123
                        }
                    }
                }
            }
        }
    }
}

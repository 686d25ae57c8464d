// aev_conv_1x1.hpp -- the 1x1 conv kernel (the e-fcn head: conv6, conv7).  Included at the end of aev_kernels.hpp.
#pragma once

// ------------------------------------------------------------------------------------------------
// 1x1 conv (the e-fcn head: conv6 256->512, conv7 512->110), f32 MFMA.  No spatial structure to exploit, so a
// tile is a run of COLUMNS over (stream, site): G consecutive streams x H*W sites are flattened and cut into
// tiles of up to ACAP columns -- a 5x7 map (35 sites) alone would fill 35 of 64 MFMA columns, four streams fill
// 140 of 160.  The block gathers the columns' channel vectors (CC channels per chunk, in flight under the previous chunk's
// MFMAs) into LDS planes [CC][PPS1]; the GEMM reads B = plane[c][column], A = weight slice [CC][CW] from LDS.  k = c
// ascending: the arithmetic contract's fmaf chain.  The source is a plain [H][W][C] map: a conv layer's state or the copy a
// pool layer keeps of its inputs (aev_hip.hip: pool_copy_fill).
// ------------------------------------------------------------------------------------------------
#define AEV_COLS1 160
struct OneArgs {
    SrcView src;
    float *fm, *act;
    long long sstride;
    const float *wt;    // [Krows][CoutP]
    const float *bias;
    int C, CoutP, HW, W, cin;
    int G, S;           // streams per group, total streams
    const uint8_t *touch; long long mstride;
    int s0, ns;
};

// Two shapes of a block.  RS = 0: 160 columns x 128 couts, wave w = cout block w over all (up to five) site blocks of the
// tile: every gathered value feeds four cout blocks -- the throughput shape.  RS = 1 ("row split"): 128 columns x 32 couts,
// wave w = site block w of the one cout block: four times the blocks and a quarter of the MFMAs per wave -- the shape for few
// streams, where a 5x7 map x 128 couts is ONE block of the other shape (conv7 of the e-fcn alone: 42 -> 25 us).  With many
// streams it loses: a quarter of the MFMAs per gathered byte makes the kernel gather-bound (+25 us per step at 256 streams).
template <int CC, int RS> struct OneLds {
    static constexpr int COLS = RS ? 128 : AEV_COLS1;        // columns (sites over the streams of a group) of a tile
    static constexpr int PP = COLS + 1;                      // plane stride (+1 against bank conflicts of the transposed writes)
    static constexpr int CW = RS ? 32 : 128;                 // couts of a block
    static constexpr int BUF = 2 * CC * PP + CC * CW;        // floats per buffer: P2 planes + Wl
    static constexpr int TABLES = 2 * BUF;                   // cio[PP], cvo[PP], lsite[COLS], wcnt[8], loff[COLS] follow
};
// Same construction as conv_patch_body (see there): two LDS buffers, the next chunk's gather issued as single loads /
// stores between the MFMAs of the current one with its arithmetic in two batches, no VALU instruction in the k loop
// (k-step t reads plane 2t in lanes 0-31 and plane 2t+1 in lanes 32-63: one base per site block + an immediate).
template <int CC, int NACT, int RS>
__device__ __forceinline__ void conv_1x1_body(const OneArgs &a, float *lds, const int *lsite, const int *loff, int nact, int sbase, int col0)
{
    using LD = OneLds<CC, RS>;
    constexpr int CW = LD::CW, PP = LD::PP, COLS = LD::COLS;
    constexpr int WF4 = CC * CW / 4;
    constexpr int NWL = WF4 / 256;            // 16-byte passes over the weight slice (exact for CC = 16, 32; row split: CC = 32)
    constexpr int NA = NACT > 0 ? NACT : 1;
    static_assert(!RS || NACT <= 1, "row split: one site block per wave");
    constexpr int BUF = LD::BUF;
    static_assert(WF4 % 256 == 0, "the weight slice is a whole number of passes");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SrcView &v = a.src;
    const int cout_base = blockIdx.z * CW + (RS ? 0 : wv * 32);
    const int blk0 = RS ? wv : 0; // first (row split: only) site block of this wave
    const int h = lane >> 5;

    // Gather mapping (state is channel-fastest): consecutive lanes take consecutive channels of one column and the write
    // into the [c][column] LDS planes transposes.  Four channels per element, 16-byte loads (pass i of thread t: channels
    // 4*(t % (CC/4)) .. +3, column i*CSTEP + t/(CC/4)).
    constexpr int G4 = CC / 4;
    constexpr int CSTEP = 256 / G4;                        // columns covered by one pass
    constexpr int NE = (COLS + CSTEP - 1) / CSTEP;         // elements per thread and chunk
    const int cl = (tid % G4) * 4, q0 = tid / G4;
    int *cio = reinterpret_cast<int *>(lds + LD::TABLES); // [PP] bit 31 = no such column
    int *cvo = cio + PP;                                          // [PP] value offset of the column
    if (tid < COLS) {
        const int colg = col0 + tid;
        const int sl = colg / a.HW, site = colg - sl * a.HW;
        const bool valid = sl < a.G && sbase + sl < a.s0 + a.ns;
        const int sv = valid ? sl : 0, y = valid ? site / a.W : 0, x = valid ? site - (site / a.W) * a.W : 0;
        cio[tid] = valid ? 0 : (int)0x80000000;
        cvo[tid] = (int)(sv * v.bstride + ((long long)y * v.bW + x) * v.C);
    }
    __syncthreads(); // (also: lsite / loff complete)
    int voff[NE];
    unsigned invalid = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int col = min(i * CSTEP + q0, COLS - 1);
        const int d = cio[col];
        if (d < 0 || i * CSTEP + q0 >= COLS) invalid |= 1u << i;
        voff[i] = cvo[col] + cl;
    }
    const float *s_ptr = v.s + (long long)sbase * v.bstride, *a_ptr = v.a + (long long)sbase * v.bstride;
    const unsigned woff0 = (unsigned)(tid / (CW / 4)) * (unsigned)a.CoutP + blockIdx.z * CW + (tid % (CW / 4)) * 4;
    constexpr int WSTEP = 256 / (CW / 4); // 8 weight rows per pass

    f32x4 sval4[NE], aval4[NE];
    f32x4 wreg[NWL];
    auto T2 = [&](int q, int c) __attribute__((always_inline)) {
        sval4[q] = ld_u32x4(s_ptr + c, (unsigned)voff[q]);
        aval4[q] = ld_u32x4(a_ptr + c, (unsigned)voff[q]);
    };
    auto T3 = [&](float2 *P2n_lane) __attribute__((always_inline)) { // P2n_lane = next buffer's P2 + cl*PP + q0
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const bool bad = (invalid >> q) & 1;
            // (columns past the tile are not written)
            if ((q + 1) * CSTEP <= COLS || q * CSTEP + q0 < COLS) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float sv = sval4[q][u], av = aval4[q][u];
                    const float L = sv > 0.0f ? 1.0f : v.alpha;
                    P2n_lane[u * PP + q * CSTEP] = make_float2(bad ? 0.0f : __fmul_rn(sv, L), bad ? 0.0f : __fmul_rn(av, L));
                }
            }
        }
    };
    auto W1 = [&](int i, int c) __attribute__((always_inline)) {
        wreg[i] = ld_u32x4(a.wt + (long long)(c + i * WSTEP) * a.CoutP, woff0);
    };
    auto W2 = [&](int i, float *Wn_lane) __attribute__((always_inline)) { *reinterpret_cast<f32x4 *>(Wn_lane + i * 1024) = wreg[i]; };

    f32x16 accf[NA], accd[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accf[i][r] = 0.0f; accd[i][r] = 0.0f; }

    // ---- chunk 0 into buffer 0 (exposed) ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NWL; ++i) W1(i, 0);
#pragma unroll
    for (int q = 0; q < NE; ++q) T2(q, 0);
    int bj8[NA]; // byte offset of this lane's column (+ the odd plane for lanes 32-63) in a P2 buffer
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = (blk0 + i) * 32 + (lane & 31);
        bj8[i] = ((j < nact ? lsite[j] : 0) + h * PP) * 8;
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) W2(i, lds + 2 * CC * PP + tid * 4);
    T3(reinterpret_cast<float2 *>(lds) + cl * PP + q0);
    __syncthreads();

    auto chunk = [&](int cnext, int buf, auto more_t) __attribute__((always_inline)) {
        constexpr bool more = decltype(more_t)::value;
        constexpr int NS = CC / 2;   // k-steps = task slots of a chunk
        constexpr int QT = NS / 4;
        float *cur_b = lds + buf * BUF, *nxt_b = lds + (buf ^ 1) * BUF;
        const int p2c = buf * BUF * 4;
        const float *wr = cur_b + 2 * CC * PP + (RS ? 0 : wv * 32) + (lane & 31) + h * CW;
        float2 *P2n_lane = reinterpret_cast<float2 *>(nxt_b) + cl * PP + q0;
        float *Wn_lane = nxt_b + 2 * CC * PP + tid * 4;
        int base[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) base[i] = p2c + bj8[i];
        auto lds_pair = [&](int byte_off) __attribute__((always_inline)) {
            return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(lds) + byte_off);
        };
        if (NACT > 0) {
            float aw[2];
            float2 bv[2][NA];
            aw[0] = wr[0];
#pragma unroll
            for (int i = 0; i < NA; ++i) bv[0][i] = lds_pair(base[i]); // (exposed)
#pragma unroll
            for (int t = 0; t < NS; ++t) { // k-step t: channels 2t (lanes 0-31) and 2t+1 (lanes 32-63)
                const int cur = t & 1, nxt = cur ^ 1;
                if (t < NS - 1) {
                    aw[nxt] = wr[(2 * (t + 1)) * CW];
#pragma unroll
                    for (int i = 0; i < NA; ++i) bv[nxt][i] = lds_pair(base[i] + 2 * (t + 1) * PP * 8);
                }
                if (more) {
#pragma unroll
                    for (int q = 0; q < NE; ++q)
                        if ((q * QT) / NE == t) T2(q, cnext);
#pragma unroll
                    for (int i2 = 0; i2 < NWL; ++i2) {
                        const int S1 = (i2 * 2 * QT) / NWL, S2 = 2 * QT + S1;
                        if (S1 == t) W1(i2, cnext);
                        if (S2 == t) W2(i2, Wn_lane);
                    }
                    if (t == 3 * QT) T3(P2n_lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    accf[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][i].x, aw[cur], accf[i], 0, 0, 0); // sites on the rows
                    accd[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][i].y, aw[cur], accd[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (more) {
#pragma unroll
            for (int i = 0; i < NWL; ++i) W1(i, cnext);
#pragma unroll
            for (int q = 0; q < NE; ++q) T2(q, cnext);
#pragma unroll
            for (int i = 0; i < NWL; ++i) W2(i, Wn_lane);
            T3(P2n_lane);
        }
        if (more) __syncthreads(); // buffer b^1 is complete, and every wave is done reading buffer b
    };
    const float bias_v = a.bias[min(cout_base + (lane & 31), a.C - 1)]; // fetched ahead of the GEMM: no exposed round trip in the scatter
    int buf = 0, c0 = 0;
#pragma unroll 1
    for (; c0 + CC < a.cin; c0 += CC, buf ^= 1) chunk(c0 + CC, buf, std::true_type{});
    chunk(0, buf, std::false_type{});

    if (NACT > 0) {
        const int co = cout_base + (lane & 31);
        float *fm = a.fm + (long long)sbase * a.sstride, *act = a.act + (long long)sbase * a.sstride;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            store_site_offsets(fm, act, accf[i], accd[i], bias_v, co, a.C, lane, 32, loff + (blk0 + i) * 32); // (padding rows: sink)
        }
    }
}

// grid: x = tiles per stream group, y = stream groups, z = cout blocks of 128 (row split: of 32)
template <int CC, int RS>
__global__ __launch_bounds__(256, RS ? 2 : 1) void k_conv_1x1(OneArgs a)
{
    extern __shared__ float lds[];
    using LD = OneLds<CC, RS>;
    constexpr int ACAP = LD::COLS;
    const int sbase = a.s0 + blockIdx.y * a.G;
    const int col0 = blockIdx.x * ACAP;
    int *lsite = reinterpret_cast<int *>(lds + LD::TABLES) + 2 * LD::PP + 2; // [ACAP] (after the two column tables; 16-byte aligned)
    int *wcnt = lsite + ACAP;                                                          // [4] (+4 pad)
    int *loff = wcnt + 8;                                                              // [ACAP] element offset of every active site, from stream sbase
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const int colg = col0 + tid;
        const int sl = colg / a.HW, site = colg - sl * a.HW;
        const bool act = tid < ACAP && sl < a.G && sbase + sl < a.s0 + a.ns &&
                         a.touch[(long long)(sbase + sl) * a.mstride + site] != 0;
        const u64 bal = __ballot(act);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; ++w) base += wcnt[w];
        if (act) {
            const int j = base + __popcll(bal & ((1ull << lane) - 1));
            lsite[j] = tid;
            loff[j] = (int)(sl * a.sstride + (long long)site * a.C);
        }
    }
    const int nact = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (nact == 0) return;
    if (tid < ((32 - (nact & 31)) & 31)) loff[nact + tid] = a.HW * a.C; // padding rows of the last site block -> the sink behind stream sbase's map
    if constexpr (RS != 0) { // (waves without a site block only gather; the same barriers on both sides)
        if (wv * 32 < nact) conv_1x1_body<CC, 1, RS>(a, lds, lsite, loff, nact, sbase, col0);
        else conv_1x1_body<CC, 0, RS>(a, lds, lsite, loff, nact, sbase, col0);
    } else {
        switch ((nact + 31) >> 5) {
        case 1: conv_1x1_body<CC, 1, RS>(a, lds, lsite, loff, nact, sbase, col0); break;
        case 2: conv_1x1_body<CC, 2, RS>(a, lds, lsite, loff, nact, sbase, col0); break;
        case 3: conv_1x1_body<CC, 3, RS>(a, lds, lsite, loff, nact, sbase, col0); break;
        case 4: conv_1x1_body<CC, 4, RS>(a, lds, lsite, loff, nact, sbase, col0); break;
        default: conv_1x1_body<CC, 5, RS>(a, lds, lsite, loff, nact, sbase, col0); break;
        }
    }
}


// aev_conv_patch.hpp -- the spatial-tile kernel: every 3x3 async conv layer with Cin % 4 == 0 and Cout >= 32 (conv2 ... conv5 of the e-fcn).
// Included at the end of aev_kernels.hpp (SrcView, the load / store helpers and the tile lists are defined there).
#pragma once

// ------------------------------------------------------------------------------------------------
// 3x3 conv, spatial-tile kernel: the production path for every 3x3 layer with an even Cin.
//
// One work item = one TH x TW tile of OUTPUT sites of one stream (x one block of couts), taken from the layer's list of
// active tiles; a (persistent) block
//   1. expands the tile's 256-bit site mask (built next to the list by k_post / k_tile_list) into the row-major table
//      of active sites and their element offsets -- every wave for itself, no exchange;
//   2. stages the (TH+2) x (TW+2) receptive-field patch of the previous layer's feature map and decay
//      map ONCE per CC input channels in LDS (the 9-fold overlap of neighbouring receptive fields is
//      served from LDS instead of 9 global gathers: 204*Cin fetches per 128-site tile, not 1152*Cin).
//      The source is always a plain [H][W][C] map: a conv layer's state, or the copy a pool layer keeps of
//      its inputs at the argmax sites (aev_hip.hip: pool_copy_fill) -- no argmax byte is chased here;
//   3. stages the matching [CC*9][CW] slice of the transposed weights in LDS with 16-byte loads;
//   4. runs the implicit GEMM on v_mfma_f32_16x16x4_f32 (16 sites x 16 couts x the four channels of a group at one tap):
//      B operand = patch[c][ly+dy][lx+dx] read with a per-lane base (the site) + an immediate (the k index), A operand
//      = weight slice; two right-hand sides (feature, decay) share every weight fragment;
//   5. scatters only the active sites (fm with bias, act without).
// k order inside a tile is the contract's chain order (four-channel group, tap, channel), chunk after chunk, so the result is bit for bit
// the fmaf chain of the arithmetic contract (identical to k_conv_valu / k_conv_mfma).
// Waves: NSW site groups x NCW cout groups; a wave owns up to 2 * NSBW site blocks (16 sites each, strided by NSW)
// x CBW cout blocks (32 each, as two 16-cout halves) x 2 maps accumulators.
// ------------------------------------------------------------------------------------------------
struct PatchArgs {
    SrcView src;
    float *fm, *act;
    long long sstride;
    const float *wt;    // [Krows][CoutP]
    const float *bias;
    int C, CoutP, H, W, HW, cin, pt, pl;
    int TH, TW, tilesX;
    int tiles, gz;              // tiles per stream, cout blocks
    int force_queue;            // tests: > 0 use the work queue whatever the number of items, < 0 never
    int static_eighths;         // eighths of the items handed out by striding before the work queue takes over (k_conv_patch)
    const int *tlist, *tcount;  // active tiles (stream-local index * tiles + tile) in AEV_TL_BUCKETS segments of cap entries; tcount[0] = their number (work items = tcount[0] * gz), tcount[1] = work queue, tcount[4 + b] = entries of segment b
    int cap, nb;                // entries per segment, segments in use
    const u64 *tmask;           // [entry][4] the tiles' active sites, bit = ly * TW + lx
    int s0;
    unsigned long long *stamps; // diagnostics (AEV_STAMPS): per-item phase cycle sums of wave 0 [stamp_items][8], then per-block start / end [grid][2]; else nullptr
    int stamp_items;
};

// The body for a wave that owns NACT (compile-time) active site blocks: no data-dependent branch sits inside the
// MFMA loop, so the accumulators stay in AGPRs without copies.  Every wave of the block runs the same number of
// barriers whatever its NACT.
//
// LDS image of a chunk (two buffers): P2[CC][257] (feature, decay) pairs -- one 256-slot plane per input channel,
// slot = py*16 + px of the patch (row stride fixed at AEV_PW = 16: TW <= 14, TH <= 14) -- and Wl[CC*9][CW], the
// matching rows of the transposed weights.
//
// What a wave can and cannot hide under its own MFMAs (tools/probes/mfma_shadow.hip: one wave per SIMD,
// v_mfma_f32_32x32x2_f32 = 64 cycles): LDS reads / writes, global loads and SALU instructions placed between two
// MFMAs cost nothing; every VALU instruction costs its 4 cycles, and every interruption of the MFMA stream by VALU
// work another 12 (one instruction) to 45 cycles (a batch).  Hence
//   * the k loop contains NO VALU instruction: with the compile-time row stride every operand address is a per-lane
//     base register + an immediate.  Lanes 32-63 take the odd k of a k-step -- the next channel of the group at the same tap,
//     i.e. the next plane: one base per site block;
//   * while the MFMAs run on buffer b the SAME waves build chunk c+1 in buffer b^1, as single load / store
//     instructions spread over the chunk's k-steps, with their arithmetic gathered in two batches per chunk (measured
//     before, with the gather issued in one burst ahead of the GEMM and committed after it: 6 k of 32 k cycles per
//     chunk with the MFMA pipe idle, the four waves queueing on the CU's one address unit):
//         slot 0                  : A2  addresses of the next chunk's values from the argmax bytes      (VALU batch)
//         quarter 0 of the slots  : T2  surface / decay values of chunk c+1                          (global -> VGPR)
//                                   T1  argmax bytes of the pool indirection of chunk c+2            (global -> VGPR)
//         quarters 0-1 / 2-3      : W1 / W2  weight rows, 16 B per lane                     (global -> VGPR -> LDS)
//         first slot of quarter 3 : A3 + T3  slope, products (VALU batch) -> P2 of buffer b^1          (VGPR -> LDS)
//     then ONE barrier.  Every load has half a chunk (>= 5 us at five site blocks per wave) to land.
#define AEV_PPS 257 // patch plane stride (256 slots per channel + 1: the transposed gather writes spread over the LDS banks)
#define AEV_PW 16   // patch row stride
template <int CC, int CW, int NT = 256> struct PatchLds {
    static constexpr int NWL = (CC * 9 * CW / 4 + NT - 1) / NT; // passes of the block's lanes x 16 B over the weight slice
    static constexpr int BUF = 2 * CC * AEV_PPS + CC * 9 * CW; // floats per buffer: P2 planes + Wl
    static constexpr int TABLES = 2 * BUF;                     // lsite[ACAP], wcnt[4], loff[ACAP] follow
};
// slot offset of position k of a four-channel group's chain: channel k % 4 of the group at tap k / 4 (aev_device.hpp: chain_pos)
__host__ __device__ constexpr int patch_koff(int k) { return (k % 4) * AEV_PPS + ((k / 4) / 3) * AEV_PW + (k / 4) % 3; }
// what a block carries from one work item to the next, and what it knows of the next item
template <int NEL> struct PatchCarry {
    // per-lane source offsets and block-uniform source pointers of the item about to start: the hand-over into the next item set them, so a
    // staged item starts without recomputing them (a third of the instructions between two items' GEMMs)
    int voff[NEL];
    unsigned outside;
    const float *s_ptr, *a_ptr, *wt_ptr;
    int buf;      // buffer that holds chunk 0 of the item about to start
    bool staged;  // the previous item's last chunk built this item's chunk 0
    bool wres;    // the layer's WHOLE weight matrix is in LDS (see WRES)
    unsigned long long t_top, t_disp; // (diagnostics: shader clock at the item's top and right in front of the dispatch)
};
struct PatchNext { bool has; int s, ty0, tx0, z; };
// elements (four channels of one patch slot, one 16-byte load per map) per thread and chunk of the gather (see conv_patch_body)
template <int CC, int NT = 256> struct PatchGather { static constexpr int NEL = 256 * (CC / 4) / NT; }; // (256 patch slots x CC / 4 channel quads over the block's threads)
// Site blocks are 16 sites (v_mfma_f32_16x16x4_f32: 16 sites x 16 couts x the four channels of a group at one tap), not the 32 of
// v_mfma_f32_32x32x2_f32 (rounds 1-2): the same chain of fmaf per output (tools/probes/mfma_order.hip), the same MFMA cycles per
// site, but a tile's last site block is padded to 16 rows instead of 32 -- the padding rows were 12-25 % of the MFMAs.
// pq, pt, h1, h2: the kernel's item pipeline (see k_conv_patch) -- pq in front of the barrier of every chunk that has one, pt behind
// the item's last chunk, in front of the scatter; items of four chunks or more (la1): h1 in front of the second chunk, h2 in front
// of the second-to-last one -- the next item is not known before h2.
// HALF: the block's couts are ONE block of 32 and the NCW = 2 wave groups split it in halves of 16 -- wave (site group, half):
// the unit of work is (16-site block, 16 couts), so a tile's site blocks spread over two site groups instead of four (conv2 of
// the e-fcn: 5.4 site blocks per tile over four waves = 2 : 1 : 1 : 1, and every chunk barrier waits for the first).
// WRES: a layer of exactly two chunks and one cout block (conv2 of the e-fcn: 16 input channels, 32 couts) has its whole weight matrix
// in the two weight slices of the LDS image -- chunk 0's slice lives in buffer 0 and chunk 1's in buffer 1 for the life of the block
// (an item's chunk c always lands in buffer c: staged items keep the parity, exposed ones start in buffer 0) -- so only a block's
// first item stages weights; every later item (WRES) skips the weight tasks: a quarter of its vector-memory instructions and half of
// the bytes it pulls through L1 (18 KB of weights against 20 KB of patch values per item).  Round 4: conv2 -1.7 %, step -0.4 %.
template <int CC, int NSW, int NCW, int CBW, int NACT, bool HALF, bool WRES, class HookQ, class HookT, class Hook1, class Hook2>
__device__ __forceinline__ void conv_patch_body(const PatchArgs &a, float *lds, const int *lsite, const int *loff, int nact, int s,
                                                int ty0, int tx0, int z, int item, const PatchNext &nx, PatchCarry<PatchGather<CC, 64 * NSW * NCW>::NEL> &carry, int sw,
                                                HookQ &pq, HookT &pt, bool la1, Hook1 &h1, Hook2 &h2)
{
    static_assert(NSW * NCW == 4 || NSW * NCW == 8, "four or eight waves per block");
    static_assert(!HALF || (NCW == 2 && CBW == 1), "half mode: two wave groups, one cout block");
    constexpr int CW = HALF ? 32 : NCW * CBW * 32;
    constexpr int HB = HALF ? 1 : 2;            // 16-cout halves of a cout block a wave computes
    constexpr int NT = 64 * NSW * NCW;          // threads of the block: a wave per (site group, cout group)
    constexpr int NWL = PatchLds<CC, CW, NT>::NWL;
    constexpr int NA = NACT > 0 ? NACT : 1;
    constexpr int PP = AEV_PPS, PW = AEV_PW;
    constexpr int BUF = PatchLds<CC, CW>::BUF;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SrcView &v = a.src;
    const int cw = wv / NSW; // (sw: the wave's site group of this item, chosen by the caller)
    const int cwoff = HALF ? cw * 16 : cw * CBW * 32; // the wave's first cout within the block's
    const int cout_base = z * CW + cwoff;
    // No runtime condition guards a load: hipcc turns a guarded load (even under a wave-uniform condition) into branch +
    // s_waitcnt vmcnt(0) per element and serialises the gather.  Out-of-map positions load from a clamped address and are
    // zeroed when the LDS image is written.
    //
    // Gather mapping: state is channel-fastest, so consecutive lanes take consecutive CHANNELS of one patch position
    // (a wave reads a few contiguous runs instead of 64 separate lines) and the write into the [c][slot] LDS planes
    // is the transpose.  FOUR channels per element, one 16-byte load per map; pass i of thread t: channels
    // 4*(t % (CC/4)) .. +3, slot i * (256 / (CC/4)) + t / (CC/4).
    constexpr int G4 = CC / 4;                    // channel groups of a chunk
    constexpr int NEL = PatchGather<CC, NT>::NEL; // elements per thread and chunk
    constexpr int ESTEP = NT / G4;                // slots covered by one pass
    const int cl = (tid % G4) * 4, p0 = tid / G4;
    int (&voff)[NEL] = carry.voff;
    unsigned &outside = carry.outside;
    // per-lane source offsets of a tile's patch: voff (values), bit mask of the zero-padded slots
    auto tile_offsets = [&](int tY0, int tX0) __attribute__((always_inline)) {
        unsigned out = 0;
#pragma unroll
        for (int i = 0; i < NEL; ++i) {
            const int p = i * ESTEP + p0, py = p / PW, px = p % PW;
            const int iy = tY0 - a.pt + py, ix = tX0 - a.pl + px;
            const bool inside = py < a.TH + 2 && px < a.TW + 2 && iy >= 0 && iy < v.H && ix >= 0 && ix < v.W; // zero padding, conv2d.py:68-72
            const int cy = min(max(iy, 0), v.H - 1), cx = min(max(ix, 0), v.W - 1);
            if (!inside) out |= 1u << i;
            voff[i] = (cy * v.bW + cx) * v.C + cl;
        }
        outside = out;
    };
    // Every address below is a block-uniform base (SGPR pair, advanced with SALU instructions) + an unsigned 32-bit
    // per-lane offset that does not change from chunk to chunk: the tasks are (almost) pure load / store instructions.
    // (The bases are variables: the last chunks of an item point them at the next item -- a staged item finds them, and its
    //  offsets, in the carry.)
    const float *&s_ptr = carry.s_ptr, *&a_ptr = carry.a_ptr, *&wt_ptr = carry.wt_ptr;
    if (!carry.staged) {
        tile_offsets(ty0, tx0);
        s_ptr = v.s + (long long)s * v.bstride; a_ptr = v.a + (long long)s * v.bstride;
        wt_ptr = a.wt + z * CW;
    }
    constexpr int WSTEP = NT / (CW / 4); // weight rows covered by one pass of the block (16 B per lane)
    const unsigned woff0 = (unsigned)(tid / (CW / 4)) * (unsigned)a.CoutP + (tid % (CW / 4)) * 4;

    f32x4 sval4[NEL], aval4[NEL];
    f32x4 wreg[NWL];
    auto T2 = [&](int q, int c) __attribute__((always_inline)) { // values of channels c + cl .. +3 at the element's patch slot
        sval4[q] = ld_u32x4(s_ptr + c, (unsigned)voff[q]);
        aval4[q] = ld_u32x4(a_ptr + c, (unsigned)voff[q]);
    };
    auto T3 = [&](float2 *P2n_lane) __attribute__((always_inline)) { // P2n_lane = next buffer's P2 + cl*PP + p0
#pragma unroll
        for (int q = 0; q < NEL; ++q) {
            const bool out = (outside >> q) & 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sv = sval4[q][u], av = aval4[q][u];
                const float L = sv > 0.0f ? 1.0f : v.alpha;                                   // conv2d.py:85-88
                const float F = __fmul_rn(sv, L), D = __fmul_rn(av, L);                       // layer.py:80, conv2d.py:93
                P2n_lane[u * PP + q * ESTEP] = make_float2(out ? 0.0f : F, out ? 0.0f : D);
            }
        }
    };
    // weight rows: pass i of the block moves rows [i*WSTEP, (i+1)*WSTEP) of the chunk's slice, 16 B per lane.  The last pass
    // may run past the slice: it reads the following rows (the host pads the weight array) and does not store them.
    auto W1 = [&](int i, int c) __attribute__((always_inline)) { wreg[i] = ld_u32x4(wt_ptr + (long long)(c * 9 + i * WSTEP) * a.CoutP, woff0); };
    auto W2 = [&](int i, float *Wn_lane) __attribute__((always_inline)) { // Wn_lane = destination buffer's Wl + tid*4
        constexpr int WF4 = CC * 9 * CW / 4;
        if ((i + 1) * NT <= WF4 || tid < WF4 - i * NT) *reinterpret_cast<f32x4 *>(Wn_lane + i * NT * 4) = wreg[i];
    };

    // diagnostic stamps (guide: 'In-kernel stamps'): only when a.stamps is set
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool stamping = a.stamps != nullptr;
    auto stamp = [&](int ph) __attribute__((always_inline)) {
        if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tph[ph] += t - tlast; tlast = t; }
    };
    if (stamping) tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long t_entry = tlast;

    constexpr int SB = 16;                            // sites per site block
    // The scatter's bias values, fetched HERE: loaded where they are used (as until round 3) every (site block, half) of the
    // epilogue was  global_load -> s_waitcnt vmcnt(0) -> stores, and that wait also drains the stores of the pair before it
    // (gfx9 counts loads and stores in one counter; with both kinds in flight hipcc waits for zero) -- up to four exposed
    // round trips per item, the 3.5 k cycles the stamps showed in conv2's epilogue.  Clamped address, no guard (C % 4 == 0).
    f32x4 bias4[CBW][2];
    if (NACT > 0 && (a.C & 3) == 0) {
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                bias4[cb][hb] = *reinterpret_cast<const f32x4 *>(a.bias + min(cout_base + cb * 32 + hb * 16 + 4 * (lane >> 4), a.C - 4));
    }
    f32x4 acf[NA][CBW][2], acd[NA][CBW][2]; // [site block][cout block][half of the cout block]
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acf[i][cb][hb][r] = 0.0f; acd[i][cb][hb][r] = 0.0f; }

    int buf = carry.buf;
    if (!carry.staged) { // ---- chunk 0 into buffer 0, exposed (the block's first item) -------------------
        buf = 0;
#pragma unroll
        for (int i = 0; i < (WRES ? 0 : NWL); ++i) W1(i, 0);
#pragma unroll
        for (int q = 0; q < NEL; ++q) T2(q, 0);
#pragma unroll
        for (int i = 0; i < (WRES ? 0 : NWL); ++i) W2(i, lds + 2 * CC * PP + tid * 4);
        T3(reinterpret_cast<float2 *>(lds) + cl * PP + p0);
        __syncthreads();
    }
    // this lane's site of each of the wave's site blocks as a byte offset into a P2 plane (the item's site table was built during
    // the item before, in front of a chunk barrier -- or by the kernel's prologue, in front of the barrier just above)
    int bj8[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = (sw + i * NSW) * SB + (lane & (SB - 1));
        const int ls = j < nact ? lsite[j] : 0;
        bj8[i] = ((ls >> 8) * PW + (ls & 0xff)) * 8;
    }
    stamp(0); // prologue

    // One chunk: the GEMM on buffer buf, and (MORE) the hand-over into the other buffer: values and weight rows of channel chunk
    // cnext, from wherever the bases point.  The chunk without hand-over is peeled off the loop below, so that the loop body is one
    // straight code path (with an if / else around two copies of the GEMM hipcc permutes the accumulator registers at the join,
    // through VGPRs, once per chunk).
    auto chunk = [&](int cnext, int buf, auto more_t) __attribute__((always_inline)) {
        constexpr bool more = decltype(more_t)::value;
        float *cur_b = lds + buf * BUF, *nxt_b = lds + (buf ^ 1) * BUF;
        const int p2c = buf * BUF * 4; // byte offset of the current buffer's P2 planes
        float2 *P2n_lane = reinterpret_cast<float2 *>(nxt_b) + cl * PP + p0;
        float *Wn_lane = nxt_b + 2 * CC * PP + tid * 4;
        int base[NA]; // LDS byte address of site block i's operand in a k-step = base[i] + an immediate (lane group g: channel g's plane)
#pragma unroll
        for (int i = 0; i < NA; ++i) base[i] = p2c + bj8[i] + (lane >> 4) * 8 * PP;
        auto lds_pair = [&](int byte_off) __attribute__((always_inline)) {
            return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(lds) + byte_off);
        };
        // ---- implicit GEMM on the LDS image of the chunk -------------------------------------------
        // One iteration = one group of four channels = 9 k-steps, k-step = tap: lane group lane / 16 takes channel lane / 16 of the
        // group (B: its plane at the site's tap position; A: its weight row), two MFMAs (the two 16-cout halves) per site block,
        // cout block and map.  The operands of k-step t+1 are fetched into the other half of a register double buffer before the
        // MFMAs of k-step t are issued (hipcc on its own serialises ds_read -> s_waitcnt -> v_mfma).  Task slots of the hand-over:
        // one per k-step.
        auto gemm_quad16 = [&](int cq, auto slotted) __attribute__((always_inline)) {
            constexpr bool SL = decltype(slotted)::value;
            constexpr int NS = 9;
            constexpr int QT = NS * (CC / 4) / 4 > 0 ? NS * (CC / 4) / 4 : 1; // a quarter of the chunk's slots
            const float *wr = cur_b + 2 * CC * PP + cwoff + (lane & 15) + (lane >> 4) * CW + (cq * 36) * CW;
            const int qb = cq * 4 * PP * 8;         // byte offset of the iteration's first plane
            float aw[2][CBW][2];
            float2 bv[2][NA];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) { aw[0][cb][0] = wr[cb * 32]; if (HB > 1) aw[0][cb][1] = wr[cb * 32 + 16]; }
#pragma unroll
            for (int i = 0; i < NA; ++i) bv[0][i] = lds_pair(base[i] + qb); // (exposed)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int cur = t & 1, nxt = cur ^ 1;
                if (t < 8) {
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb) { aw[nxt][cb][0] = wr[(4 * (t + 1)) * CW + cb * 32]; if (HB > 1) aw[nxt][cb][1] = wr[(4 * (t + 1)) * CW + cb * 32 + 16]; }
#pragma unroll
                    for (int i = 0; i < NA; ++i) bv[nxt][i] = lds_pair(base[i] + qb + 8 * patch_koff(4 * (t + 1)));
                }
                if (SL) {
                    const int sl = cq * NS + t; // slot within the chunk (cq is an unrolled loop counter here)
#pragma unroll
                    for (int q = 0; q < NEL; ++q)
                        if ((q * QT) / NEL == sl) T2(q, cnext);
#pragma unroll
                    for (int i2 = 0; i2 < (WRES ? 0 : NWL); ++i2) {
                        const int S1 = (i2 * 2 * QT) / NWL, S2 = 2 * QT + S1;
                        if (S1 == sl) W1(i2, cnext);
                        if (S2 == sl) W2(i2, Wn_lane);
                    }
                    if (sl == 3 * QT) T3(P2n_lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb) {
                            // couts on the ROWS (A = weights), sites on the columns (B = patch): a lane ends up with FOUR CONSECUTIVE
                            // couts of one site -- one 16-byte store per lane and map in the scatter
                            acf[i][cb][hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cur][cb][hb], bv[cur][i].x, acf[i][cb][hb], 0, 0, 0);
                            acd[i][cb][hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cur][cb][hb], bv[cur][i].y, acd[i][cb][hb], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (NACT > 0) {
#pragma unroll
            for (int cq = 0; cq < CC / 4; ++cq) gemm_quad16(cq, more_t);
        } else if (more) { // a wave without site blocks still gathers its share
#pragma unroll
            for (int i = 0; i < (WRES ? 0 : NWL); ++i) W1(i, cnext);
#pragma unroll
            for (int q = 0; q < NEL; ++q) T2(q, cnext);
#pragma unroll
            for (int i = 0; i < (WRES ? 0 : NWL); ++i) W2(i, Wn_lane);
            T3(P2n_lane);
        }
        stamp(1); // gemm (with the hand-over inside)
        if (more) { pq(); __syncthreads(); } // buffer b^1 is complete, and every wave is done reading buffer b
        stamp(2);
    };
    // Across items: with a next item and at least two chunks per item, the last chunk of this item builds chunk 0 of the NEXT
    // item, so that the next item starts its GEMM without waiting for memory.  The first item of a block, and items of one chunk,
    // start exposed.
    const int nch = a.cin / CC;
    bool xp = !la1 && nx.has && nch >= 2;
    int nsl = xp ? nch : nch - 1; // chunks that carry a hand-over
#pragma unroll 1
    for (int ci = 0; ci < nsl; ++ci, buf ^= 1) {
        if (la1 && ci == 1) h1();
        if (la1 && ci == nch - 2) { // only now is the next item known
            h2();
            xp = nx.has;
            nsl = xp ? nch : nch - 1;
        }
        int cnext = (ci + 1) * CC;
        if (xp && ci == nch - 1) { // the values and the weights fetched from here on belong to the next item
            tile_offsets(nx.ty0, nx.tx0);
            s_ptr = v.s + (long long)nx.s * v.bstride; a_ptr = v.a + (long long)nx.s * v.bstride;
            wt_ptr = a.wt + nx.z * CW;
            cnext = 0;
        }
        chunk(cnext, buf, std::true_type{});
    }
    if (!xp) chunk(0, buf, std::false_type{});
    pt(); stamp(4);
    carry.staged = xp;
    carry.buf = buf;
    // ---- scatter the active sites --------------------------------------------------------------------
    if (NACT > 0) {
        float *fm = a.fm + (long long)s * a.sstride, *act = a.act + (long long)s * a.sstride;
        // C/D layout of the 16x16x4 MFMA with the couts on the rows: register r of lane l = cout 4 * (l / 16) + r of the 16-cout half,
        // site l % 16 of the block: a lane stores its site's four consecutive couts with ONE 16-byte instruction per map
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned ro = (unsigned)loff[(sw + i * NSW) * 16 + (lane & 15)]; // (padding rows: sink)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    const int co = cout_base + cb * 32 + hb * 16 + 4 * (lane >> 4);
                    if ((a.C & 3) == 0) {
                        if (co < a.C) {
                            const f32x4 b4 = bias4[cb][hb];
                            f32x4 vf;
#pragma unroll
                            for (int r = 0; r < 4; ++r) vf[r] = __fadd_rn(acf[i][cb][hb][r], b4[r]);
                            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(fm) + (size_t)((ro + (unsigned)co) * 4u)) = vf;
                            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(act) + (size_t)((ro + (unsigned)co) * 4u)) = acd[i][cb][hb];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < a.C) {
                                st_u32(fm, ro + (unsigned)(co + r), __fadd_rn(acf[i][cb][hb][r], a.bias[co + r]));
                                st_u32(act, ro + (unsigned)(co + r), acd[i][cb][hb][r]);
                            }
                    }
                }
        }
    }
    if (stamping && tid == 0) {
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        unsigned long long *o = a.stamps + (long long)item * 8;
        for (int q = 0; q < 3; ++q) o[q] = tph[q];
        o[3] = tend - tlast; // epilogue
        o[4] = tph[4];       // the item pipeline's hook behind the last chunk (the next item's site table)
        // site blocks of the item | cycles from the item's top to the dispatch (requests) << 16 | from there to here (dispatch, offsets, pointers) << 40
        o[5] = (unsigned long long)((nact + SB - 1) / SB) | (min(carry.t_disp - carry.t_top, 0xffffffull) << 16) | (min(t_entry - carry.t_disp, 0xffffffull) << 40);

        o[7] = NACT;
    }
}

// Persistent blocks: the grid is what fits on the chip at once, and a block walks over the work items (active tile x cout
// block z, z fastest) with a stride of the grid.  A freshly launched block costs several microseconds before its first
// MFMA (dispatch, then site mask -> value, two dependent trips to memory); measured with one block per
// item, a third of the wave slots of conv2 stood empty.  The site mask of the NEXT item is fetched a whole item early.
// wave-uniform dispatch on the number of site blocks the wave owns in this item (a compile-time constant inside the body)
template <int CC, int NSW, int NCW, int CBW, int N, bool HALF, bool WRES> struct PatchDispatch {
    template <typename... Rest> static __device__ __forceinline__ void run(int mine, Rest &&...rest)
    {
        if (mine >= N) conv_patch_body<CC, NSW, NCW, CBW, N, HALF, WRES>(rest...);
        else PatchDispatch<CC, NSW, NCW, CBW, N - 1, HALF, WRES>::run(mine, rest...);
    }
};
template <int CC, int NSW, int NCW, int CBW, bool HALF, bool WRES> struct PatchDispatch<CC, NSW, NCW, CBW, 0, HALF, WRES> {
    template <typename... Rest> static __device__ __forceinline__ void run(int, Rest &&...rest) { conv_patch_body<CC, NSW, NCW, CBW, 0, HALF, WRES>(rest...); }
};

template <int CC, int NSW, int NCW, int CBW, int NSBW, bool HALF = false>
__global__ __launch_bounds__(64 * NSW * NCW, (NSW * NCW == 8 ? 1 : HALF ? 3 : NSBW == 1 ? 3 : NSBW == 2 || NSBW == 4 ? 2 : 1)) void k_conv_patch(PatchArgs a)
{
    extern __shared__ float lds[];
    constexpr int CW = HALF ? 32 : NCW * CBW * 32; // couts per block
    constexpr int ACAP = 32 * NSW * NSBW;       // max active sites per tile
    static_assert(NSBW <= 5, "at most 5 site blocks per wave");
    int *lsite_t = reinterpret_cast<int *>(lds + PatchLds<CC, CW>::TABLES); // [2][ACAP] active sites of the item (even / odd items), (ly << 8) | lx
    int *loff_t = lsite_t + 2 * ACAP;                                           // [2][ACAP] element offset of every active site in the stream's map
    int *qslot = loff_t + 2 * ACAP;                                             // [4] work-queue hand-over between the waves
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long t_blk = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0; // diagnostic: block start (100 MHz)
    const unsigned long long c_blk = a.stamps ? __builtin_amdgcn_s_memtime() : 0;     //             ... in shader clocks
    const int items = __builtin_amdgcn_readfirstlane(a.tcount[0]) * a.gz;
    const int G = (int)gridDim.x;
    // Work queue: a block's first item is its index, every further one comes from an atomic counter (tcount[1], zeroed
    // before the launch) -- blocks that become resident late (another stream's kernels on the CU) or draw heavy tiles
    // simply take fewer.  With few items per block the queue's look-ahead (two items drawn ahead) would unbalance the tail:
    // plain striding then (round r gives block b item r * G + (b + r) % G -- rotated, because the list repeats with the number
    // of tiles per stream and a plain stride would hand a block the same kind of tile, large or small, every round).
    // (strided rounds alternate direction: the list is sorted by size, see AEV_TL_BUCKETS)
    auto strided = [&](int r) { return r * G + ((r & 1) ? G - 1 - (int)blockIdx.x : (int)blockIdx.x); };
    // List position of the t-th tile in the kernel's order (fullest segment first).  One register holds the map: lanes 0-31 the
    // first t of segment b, lanes 32-63 what to add to t within it.
    int tmap = 0; // (one segment: position = t)
    if (a.nb > 1) {
        static_assert(AEV_TL_BUCKETS <= 32, "the segment map lives in the lanes of one register");
        const int cv = lane < AEV_TL_BUCKETS ? a.tcount[4 + lane] : 0;
        int first = 0; // tiles in the segments above this lane's
#pragma unroll 1
        for (int b = a.nb - 1; b > 0; --b)
            if ((lane & 31) < b) first += __builtin_amdgcn_readlane(cv, b);
        tmap = lane < 32 ? first : (lane & 31) * a.cap - first;
    }
    auto lpos = [&](int t) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln)); // (opaque: see build_table)
        if (a.nb <= 1) return t;
        const int b = __ffsll((long long)__ballot(ln < AEV_TL_BUCKETS && tmap <= t)) - 1; // the lowest segment whose first t is <= t (the top segment's is 0)
        return t + __builtin_amdgcn_readlane(tmap, 32 + b);
    };
    // The queue takes over for the LAST rounds only: the first R0 rounds are strided (no atomic), the counter hands out the items from
    // R0 * G on.  All blocks of a launch drawing every item from one counter is one same-address atomic per item: ~15 ns each at
    // the L2 -- a floor of 170 us for conv2's 11 k items at 256 streams (measured with everything else knocked out: 167 us with
    // the queue, 86 us strided), bursts of several hundred of them while the blocks still run in step (their returns come back in
    // order with the gather's loads, which then wait), and 6 % of the blocks' time against striding.
    // Items of four chunks or more (la1) are drawn ONE item ahead instead of three (below), and then the queue pays for launches of few
    // items per block as well, on a list sorted by size: conv4 of the e-fcn at 256 streams is four items of 80 us per block (queue
    // from more than three items per block on: step -0.6 %; conv5's two items per block stay strided).
    const bool la1 = a.cin / CC >= 4;
    const bool dyn = (items > (la1 ? 3 : 6) * G && a.force_queue >= 0) || a.force_queue > 0;
    const int R0 = !dyn ? 0x3fffffff : a.force_queue > 0 ? 1 : max(1, (int)((long long)items * a.static_eighths / 8) / G), R0G = dyn ? R0 * G : 0;
    const unsigned *tmask32 = reinterpret_cast<const unsigned *>(a.tmask);
    // What the pipeline fetches per item, in ONE register: lanes 0-7 the eight words of a tile's site mask (list position im), the
    // other lanes the list entry at position ie.  (Clamped positions instead of guards: no branch, no wait.)
    auto fetch = [&](int im, int ie) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln)); // (opaque: see build_table)
        const unsigned *pm = tmask32 + (long long)lpos(min(im, items - 1) / a.gz) * 8 + (ln & 7);
        const unsigned *pe = reinterpret_cast<const unsigned *>(a.tlist) + lpos(min(ie, items - 1) / a.gz);
        return *(ln < 8 ? pm : pe);
    };
    // The site table of a tile from its 256-bit mask (fv: see fetch), into table par: active sites row-major, (ly << 8) | lx and
    // the element offset in the stream's map.  Every wave builds the whole table (the same values to the same addresses) and reads
    // only what it wrote itself.  Returns the number of active sites.
    // (The lane index is made opaque: left alone, hipcc hoists everything that depends on the lane only -- site coordinates, lane
    //  masks, address parts: some 35 registers -- out of the item loop and keeps it alive through the GEMM.)
    const int tw_inv = 65536 / a.TW + 1; // q / TW for q < 256, TW <= 14: (q * tw_inv) >> 16
    auto build_table = [&](unsigned fv, int par, int tY0, int tX0) __attribute__((always_inline)) {
        int *ls = lsite_t + par * ACAP, *lo = loff_t + par * ACAP;
        int n = 0, ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 m = (u64)(unsigned)__builtin_amdgcn_readlane((int)fv, 2 * j) | ((u64)(unsigned)__builtin_amdgcn_readlane((int)fv, 2 * j + 1) << 32);
            if ((m >> ln) & 1) {
                const int r = n + __popcll(m & ((1ull << ln) - 1));
                const int q = ln + 64 * j, ly = (q * tw_inv) >> 16, lx = q - ly * a.TW;
                ls[r] = (ly << 8) | lx;
                lo[r] = ((tY0 + ly) * a.W + tX0 + lx) * a.C;
            }
            n += __popcll(m);
        }
        // the rows that pad the last site block store into the sink behind the stream's map (the site at index H*W; aev_hip.hip:
        // sstride): the scatter then has no per-row tests -- with them a partial block took 1.7x as long as a full one
        if (ln < ((32 - (n & 31)) & 31)) lo[n + ln] = a.HW * a.C;
        return n;
    };
    // ---- the item pipeline ---------------------------------------------------------------------------------------------
    // The top of an item has no barrier and waits for nothing but one LDS read: at the start of item k the block holds, block-
    // uniform in SGPRs, the queue indices and list entries of items k and k+1 and the number of active sites of item k, and in
    // LDS the site table of item k (table k & 1) and the index of item k+2.
    //   top of item k       every wave reads the index of item k+2 (static striding: computes it) and requests the list entry of item
    //                       k+2 and the mask of item k+1 (one load, see fetch); thread 0 requests the queue index of item k+3 (atomic);
    //   pq, chunk barriers  thread 0 puts that index into LDS in front of a chunk barrier of the item (an item of ONE chunk has
    //                       none: the next item's top does it, with a barrier of its own);
    //   pt, behind the last chunk, in front of the scatter: the entry of item k+2 into an SGPR, the site table of item k+1 from its
    //                       mask (requested a whole item of chunks ago; the scatter's stores are not issued yet, so the wait
    //                       for the loads does not wait for stores) into table (k + 1) & 1 -- every wave is past the scatter of
    //                       item k-1, which read it: the last chunk's barrier (an exposed item: its staging barrier) lies between.
    // Until round 3 all of this sat at the top of the item, behind a barrier of its own and three waits for vector memory that
    // also drained the scatter's stores (mask and queue index went through LDS): 6 k of conv2's 32 k cycles per item (AEV_STAMPS)
    // without an MFMA in the block.
    const int nch1 = a.cin / CC < 2;
    int i0 = blockIdx.x, i1 = 0x3fffffff, i2 = 0;
    if (tid == 0 && !la1) { // (exposed, once per block)
        qslot[2] = 1 < R0 ? strided(1) : atomicAdd(const_cast<int *>(a.tcount) + 1, 1) + R0G;
        qslot[0] = 2 < R0 ? strided(2) : atomicAdd(const_cast<int *>(a.tcount) + 1, 1) + R0G;
    }
    __syncthreads();
    if (!la1) i1 = __builtin_amdgcn_readfirstlane(qslot[2]);
    int ent = i0 < items ? __builtin_amdgcn_readfirstlane(a.tlist[lpos(i0 / a.gz)]) : 0; // (exposed, once per block)
    int ent1 = i1 < items ? __builtin_amdgcn_readfirstlane(a.tlist[lpos(i1 / a.gz)]) : 0, ent2 = 0;
    int nact = 0, nact_next = 0;
    unsigned fv = 0;
    if (i0 < items) {
        const int tile = ent % a.tiles;
        fv = fetch(i0, i0);
        nact = build_table(fv, 0, (tile / a.tilesX) * a.TH, (tile % a.tilesX) * a.TW); // (read behind the first item's staging barrier)
    }
    int i3_raw = 0;
    PatchCarry<PatchGather<CC, 64 * NSW * NCW>::NEL> carry;
    carry.staged = false; carry.buf = 0; carry.outside = 0; carry.wres = false;
    const bool can_wres = HALF && a.cin == 2 * CC && a.gz == 1; // (the instance of the 32-cout layers: see WRES)
    unsigned long long t_prev = 0; int item_prev = 0; // (diagnostics)
    // (an entry is decoded once, as the NEXT item's: the item's own stream and tile origin are what the item before called nx)
    int z = i0 % a.gz, s = ent / a.tiles + a.s0, ty0 = ((ent % a.tiles) / a.tilesX) * a.TH, tx0 = ((ent % a.tiles) % a.tilesX) * a.TW;
    for (int k = 0; i0 < items; ++k) {
    const int item = i0;
    if (a.stamps) {
        const unsigned long long t_item = __builtin_amdgcn_s_memtime();
        if (tid == 0 && k > 0) a.stamps[(long long)item_prev * 8 + 6] = t_item - t_prev; // the previous item, top to top
        t_prev = t_item; item_prev = item;
        carry.t_top = t_item;
    }
    PatchNext nx;
    auto decode_next = [&]() __attribute__((always_inline)) {
        const int tn = ent1 % a.tiles;
        nx.has = i1 < items;
        nx.s = ent1 / a.tiles + a.s0; nx.ty0 = (tn / a.tilesX) * a.TH; nx.tx0 = (tn % a.tilesX) * a.TW; nx.z = i1 % a.gz;
    };
    // thread 0 requests a queue index (the opaque zero makes the address look lane-dependent: with a uniform address hipcc turns
    // the atomic into its wave-aggregated form, which reads the result back at once -- s_waitcnt vmcnt(0), ~6 k cycles, per work
    // item); + R0 * G where the result is used: no wait here
    auto draw = [&]() __attribute__((always_inline)) {
        int zoff = 0;
        asm volatile("" : "+v"(zoff));
        i3_raw = atomicAdd(const_cast<int *>(a.tcount) + 1 + zoff, 1);
    };
    if (la1) {
        // one item ahead: the index of item k+1 is requested at the top of item k, in LDS behind the first chunk (pq), read and followed
        // by the fetch of entry and mask in front of the second chunk (h1), decoded in front of the second-to-last chunk (h2) -- where
        // the hand-over into the next item begins
        nx.has = false; nx.s = s; nx.ty0 = ty0; nx.tx0 = tx0; nx.z = z;
        if (tid == 0 && k + 1 >= R0) draw();
    } else {
        decode_next();
        if (nch1 && k > 0) { // an item of one chunk has no chunk barrier for the queue hand-over
            if (tid == 0) qslot[k & 1] = i3_raw + R0G;
            __syncthreads();
        }
        i2 = k + 2 < R0 ? strided(k + 2) : __builtin_amdgcn_readfirstlane(qslot[k & 1]);
        fv = fetch(i1, i2); // mask of item k+1, entry of item k+2
        if (tid == 0 && k + 3 >= R0) draw();
    }
    auto pq = [&]() __attribute__((always_inline)) { if (tid == 0) qslot[(la1 ? k : k + 1) & 1] = i3_raw + R0G; };
    auto h1 = [&]() __attribute__((always_inline)) {
        i1 = k + 1 < R0 ? strided(k + 1) : __builtin_amdgcn_readfirstlane(qslot[k & 1]);
        fv = fetch(i1, i1); // mask and entry of item k+1
    };
    auto h2 = [&]() __attribute__((always_inline)) {
        ent1 = __builtin_amdgcn_readlane((int)fv, 8);
        decode_next();
    };
    auto pt = [&]() __attribute__((always_inline)) {
        ent2 = __builtin_amdgcn_readlane((int)fv, 8);
        nact_next = nx.has ? build_table(fv, (k + 1) & 1, nx.ty0, nx.tx0) : 0;
    };
    constexpr int SBW = 16;
    const int nsb = (nact + SBW - 1) / SBW;
    // Which site group a wave takes ROTATES from item to item.  A tile rarely fills all its site blocks (conv2 of the e-fcn:
    // 2.6 of 4 on average), the blocks are handed out from group 0 up, and a block's wave w always runs on the same SIMD: with
    // a fixed assignment the SIMD of wave 0 does MFMAs for every item of every block on the CU while the SIMD of the last
    // wave mostly gathers.  (Measured: no difference on the e-fcn; kept -- it costs nothing.)
    const int rot = k + (int)blockIdx.x + (int)(blockIdx.x >> 8);
    const int sw = (wv + rot) % NSW;
    if (a.stamps) carry.t_disp = __builtin_amdgcn_s_memtime();
    const int mine = nsb > sw ? (nsb - sw + NSW - 1) / NSW : 0; // site blocks sw, sw+NSW, ... below nsb (wave-uniform)
    if (HALF && carry.wres)
        PatchDispatch<CC, NSW, NCW, CBW, 2 * NSBW, HALF, HALF>::run(mine, a, lds, lsite_t + (k & 1) * ACAP, loff_t + (k & 1) * ACAP, nact, s, ty0, tx0, z, item, nx, carry, sw,
                                                                    pq, pt, la1, h1, h2);
    else
        PatchDispatch<CC, NSW, NCW, CBW, 2 * NSBW, HALF, false>::run(mine, a, lds, lsite_t + (k & 1) * ACAP, loff_t + (k & 1) * ACAP, nact, s, ty0, tx0, z, item, nx, carry, sw,
                                                                     pq, pt, la1, h1, h2);
    carry.wres = can_wres; // (both slices are in place once the block's first item is through)
    i0 = i1; i1 = la1 ? 0x3fffffff : i2; ent1 = ent2; nact = nact_next;
    z = nx.z; s = nx.s; ty0 = nx.ty0; tx0 = nx.tx0;
    } // items
    if (a.stamps && tid == 0 && blockIdx.x < items) { // diagnostic: the block's start and end (100 MHz), behind the item records
        unsigned long long *o = a.stamps + (long long)a.stamp_items * 8 + (long long)blockIdx.x * 4;
        o[0] = t_blk;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = __builtin_amdgcn_s_memtime() - c_blk; // shader clocks of the block: o[2] / (o[1] - o[0]) * 100 MHz = the clock it ran at
        o[3] = (unsigned long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); // HW_ID, XCC_ID: where
    }
}


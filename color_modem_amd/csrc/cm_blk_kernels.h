// cm_blk_kernels.h - time-blocked PAL delay-line decoder with the half-band FIRs on the matrix pipe (gfx950).
//
// Same arithmetic, same lane = scan line layout and the same per-sample schedule (stream indices, FilterFunction edges) as
// the streaming decoder of cm_kernels.h / cm_stages.h, but a lane advances its line 16 samples at a time and holds the
// block in registers:
//   * the five 20-tap half-band FIR chains of the path (resample_poly up / down by 2) run on v_mfma_f32_16x16x32_f16 with
//     their data operands staged through LDS (cm_blk_fir.h): per chain and 16 samples 12 MFMAs, 4 LDS stores, 8 LDS loads
//     and about 60 vector instructions instead of 320 v_fma_f32;
//   * the recursive filters walk the block sample by sample with their state in registers (the scalar / packed section
//     code of cm_stages.h / cm_stages_pk.h);
//   * one wavefront per 64 calls, one wave per SIMD, no hand-over and no barriers;
//   * the output tile is aligned to 16 samples (two outputs are carried to the next block), so a lane holds the 16 outputs
//     of its row for each plane and stores them as 4 x 16 bytes into a one-plane LDS tile; the row-wise read-back adds the
//     luma source (the same pixels of the input row, brought into LDS by global_load_lds at the start of the block) and
//     stores 64-byte row segments.
// LDS per wave: 6 operand slots (24 KiB) + input tile, luma tile, output tile (4 KiB each) = 36 KiB -> 4 waves per CU.
// Scales: every FIR runs on taps times kBlkScale; the centre taps carry the same factor, so stage outputs are kBlkScale^n
// times the streaming decoder's and the factor of the base pair (kBlkScale^4) is divided out of the per-line combination
// coefficients when a lane loads them.
#ifndef CM_BLK_KERNELS_H
#define CM_BLK_KERNELS_H

#include "cm_kernels.h"
#include "cm_blk_fir.h"

namespace cm {

// Development builds (-DCM_BLK_DIAG): cycles per stage of the interior block body, printed by one wave.
#ifdef CM_BLK_DIAG
#define CM_BLK_STAMP(i) do { if (!EDGE) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); diag_t[i] += now_ - diag_last; diag_last = now_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CM_BLK_STAMP(i) do { } while (0)
#endif

// Rows of a workgroup as 16-byte-unit offsets from a wave-uniform base: the lane that moves chunk (lane & 3) of row
// lane / 4 + 16 q keeps that row's offset for q = 0 .. 3 (fetched once from the lane that owns the row).
typedef __attribute__((address_space(1))) f4 blk_global_f4;
struct BlkRows {
    unsigned off[4];
    __device__ __forceinline__ void init(const float *base, const float *mine, int lane) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *r = ptr_from((lane / 4 + 16 * q) * 4, mine);
            off[q] = r ? (unsigned)((r - base) >> 2) : 0xffffffffu;
        }
    }
};

// Tile c (samples 16 c .. 16 c + 15 of the 64 rows) straight into LDS, [row][16] floats.
__device__ __forceinline__ void blk_fill(const Geom &g, lds_float *tile, const float *base, const BlkRows &rows, int c, int lane) {
    int col = kBlk * c + 4 * (lane & 3);
    if (col > g.Wp - 4) col = g.Wp - 4;  // never read past the (pitched) row; such samples are masked by the consumer
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const blk_global_f4 *src = (const blk_global_f4 *)base + rows.off[q] + (col >> 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(tile + q * 256), 16, 0, 0);
    }
}

// One plane of the 16-sample output tile: own row in (4 x ds_write_b128, chunks XOR-swizzled like flush_tile's), rows out
// as 64-byte segments with the luma source added.  o[16]: this lane's outputs; xl[q]: the luma source of (row, chunk) =
// (lane / 4 + 16 q, lane & 3); my: the luma column of the colour matrix for this plane.
__device__ __forceinline__ void blk_store_plane(const Geom &g, lds_float *otile, float *plane, const BlkRows &rows, const float (&o)[kBlk],
                                                const f4 (&xl)[4], float my, int first_col, int lane) {
    constexpr int kChunks = kBlk / 4, kRows = 64 / kChunks;
    lds_float *mine = otile + lane * kBlk;
    const int sw = (lane >> CM_TILE_SWZ) & (kChunks - 1);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) *(lds_f4 *)(mine + 4 * (c ^ sw)) = f4{o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int chunk = lane & (kChunks - 1);
    const int col = first_col + 4 * chunk;
#pragma unroll
    for (int q = 0; q < kChunks; ++q) {
        const int row = lane / kChunks + kRows * q;
        const int quad = chunk ^ ((row >> CM_TILE_SWZ) & (kChunks - 1));
        f4 v = *(const lds_f4 *)(otile + row * kBlk + 4 * quad);
        v += my * xl[q];
        if (rows.off[q] != 0xffffffffu && col < g.Wp)
            __builtin_nontemporal_store(v, (blk_global_f4 *)plane + rows.off[q] + (col >> 2));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// S: the tuned filter shape (even shifts of the band-pass and the detector low-pass: PAL-BG).  PAL-D front end, one line of
// history (the comb), plain luma strip by re-modulation; no notch, no minavg, float32 rows whose width is a multiple of 4.
// QE, QL: the pair delays of the band-pass and of the detector low-pass (compile-time here: they fix how many outputs a
// block carries over to the next output tile; the host checks them against the plan).
template <class S, int QE, int QL>
__global__ __launch_bounds__(64, 1) void demod_blk_kernel(const PassArgs<S> args, const BlkTiles *tiles) {
    static_assert(!S::ODD_E && !S::ODD_L && !S::RT, "the blocked decoder is built for the tuned even-shift shapes");
    typedef DemodK<float, S> K;
    constexpr int SP = S::SP, DEPTH = 1;
    __shared__ __attribute__((aligned(16))) float lds_store[kBlkSlots * kBlkSlotBytes / 4 + 3 * 64 * kBlk];
    blk_lds_byte *ops = (blk_lds_byte *)lds_store;
    lds_float *itile = (lds_float *)lds_store + kBlkSlots * kBlkSlotBytes / 4;
    lds_float *ltile = itile + 64 * kBlk;
    lds_float *otile = ltile + 64 * kBlk;
    const Geom &g = args.g;
    const K &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const float *xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
    // luma source row: the own row, or the previous call's where the plan says so (cm_kernels.h: run_pair)
    const float *lp = g.in + lc.frame * g.in_frame_stride +
                      (long long)(((g.luma_prev_bits >> lc.regime) & 1) ? lc.prev_row : lc.src_row) * g.Wp;
    const float *op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    BlkRows xrows, lrows, orows;
    xrows.init(g.in, xp, lane);
    lrows.init(g.in, lp, lane);
    orows.init(g.out, op, lane);
    StageBK<S> kb;
    kb.load(k);
    LaneKPk lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        LaneK<float> l1 = g.lanes[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
        apply_frame_rotation(g, lc.frame, l1);
        lk.load(l1, DEPTH, false);
        const float inv4 = 1.f / (kBlkScale * kBlkScale * kBlkScale * kBlkScale);    // the base pair arrives times kBlkScale^4
#pragma unroll
        for (int j = 0; j < 2; ++j) { lk.ks[j] *= inv4; lk.kc[j] *= inv4; }
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    const BlkTiles tl = tiles[lane];
    BlkAddr ad;
    ad.init(lane);
    BlkSlots sl;
    sl.init();
    for (int i = lane * 16; i < kBlkSlots * kBlkSlotBytes; i += 64 * 16) *(blk_lds_u4 *)(ops + i) = (blk_u4){0u, 0u, 0u, 0u};
    // taps: g[j] = c[j < 10 ? j : 19 - j]; the products behind the window use g[17 .. 19] = c[2], c[1], c[0]
    const float g17 = kBlkScale * k.taps.c[2], g18 = kBlkScale * k.taps.c[1], g19 = kBlkScale * k.taps.c[0];
    const float c0s = kBlkScale * k.taps.c0;      // centre tap of the scaled chains
    const float my0 = k.m[0][0], my1 = k.m[1][0], my2 = k.m[2][0];

    // ---- state of the line -----------------------------------------------------------------------------------------------
    BlkFir fx, fb, fe, fqc, fqs;
    fx.reset(); fb.reset(); fe.reset(); fqc.reset(); fqs.reset();
    IirState<float, S::NE> bpf;
    bpf.reset();
    IirStatePk<S::NL> lpf;
    lpf.reset();
    DemodBackPk<S, DEPTH, false, false> back;
    back.reset();
    float xh[10], beh[9], eh[10];            // x[t - 10 ..], band-pass even outputs [t - 9 ..], e[t - 10 ..] before the block
    pf2 qeh[9];                              // detector low-pass even outputs [t - 9 ..]
    pf2 base_c = {0.f, 0.f}, nb_c = {0.f, 0.f};      // own / neighbour base pair of the last step of the previous block
    pf2 uvd[SP > 0 ? SP : 1];
#pragma unroll
    for (int j = 0; j < 10; ++j) xh[j] = eh[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) { beh[j] = 0.f; qeh[j] = pf2{0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < (SP > 0 ? SP : 1); ++j) uvd[j] = pf2{0.f, 0.f};
    FrontLatch<float> fla;
    fla.reset();
    pf2 p_last = {0.f, 0.f}, uv_last = {0.f, 0.f};

    // ---- stream geometry (cm_kernels.h: run_pair) --------------------------------------------------------------------------
    const int W = g.W, Wp = g.Wp;
    constexpr int q_e = QE, q_l = QL;
    constexpr int front_off = 10 + q_e + 9 + 10;        // detector pair index nd = t - front_off
    constexpr int lat_front = front_off + q_l + 9;
    constexpr int lat_out = lat_front + 1 + SP;         // n7 = t - lat_out
    constexpr int kCarry = (kBlk - (lat_out & (kBlk - 1))) & (kBlk - 1);   // outputs of a block that belong to the next output tile
    constexpr int lat_tile = lat_out + kCarry;          // the tile flushed in block tb starts at tb - lat_tile
    static_assert(kCarry <= 4, "more carried outputs than the kernel is laid out for");
    float cr[kCarry > 0 ? kCarry : 1], cg[kCarry > 0 ? kCarry : 1], cb[kCarry > 0 ? kCarry : 1];
#pragma unroll
    for (int j = 0; j < (kCarry > 0 ? kCarry : 1); ++j) cr[j] = cg[j] = cb[j] = 0.f;
    const int T = ((Wp - 1) & ~(kBlk - 1)) + lat_tile + kBlk;
    // interior blocks: tb >= lat_tile keeps every stage index >= 0, tb + 15 <= W - 2 keeps them below every end-of-row latch
    int t_mid0 = lat_tile, t_mid1 = W >= kBlk + 1 ? ((W - kBlk - 1) & ~(kBlk - 1)) + kBlk : 0;
    if (t_mid1 < t_mid0) t_mid0 = t_mid1 = 0;          // short rows: the guarded body runs everything

    const lds_float *xrow = itile + lane * kBlk;
    blk_fill(g, itile, g.in, xrows, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef CM_BLK_DIAG
    unsigned long long diag_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, diag_last = 0;
    int diag_n = 0;
#endif
    auto block = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#ifdef CM_BLK_DIAG
        if (!EDGE) { diag_last = __builtin_readcyclecounter(); ++diag_n; }
#endif
        // ---- this block's input samples; the tile is refilled for the next block right away, the luma tile for this block's flush
        // (the tile is complete: every block waits for its fills before it stores)
        float xs[kBlk];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < kBlk / 4; ++q) {
            const f4 v = *(const lds_f4 *)(xrow + 4 * q);
            xs[4 * q] = v.x; xs[4 * q + 1] = v.y; xs[4 * q + 2] = v.z; xs[4 * q + 3] = v.w;
        }
        if (EDGE) {
#pragma unroll
            for (int s = 0; s < kBlk; ++s)
                if (tb + s >= W) xs[s] = 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (!EDGE || tb + kBlk < W) blk_fill(g, itile, g.in, xrows, tb / kBlk + 1, lane);
        const int tile0 = tb - lat_tile;                 // first column of the output tile this block completes
        if (!EDGE || (tile0 >= 0 && tile0 < Wp)) blk_fill(g, ltile, g.in, lrows, tile0 / kBlk, lane);

        CM_BLK_STAMP(0);
        // ---- up2(x): odd phase on the matrix pipe, even phase = centre tap of x[t - 10]
        float ao[kBlk];
        fx.run(xs, ao, tl, g17, g18, g19, ops, ad, sl);
        sl.next();
        CM_BLK_STAMP(1);
        float bo[kBlk], be[kBlk];
#pragma unroll
        for (int s = 0; s < kBlk; ++s) {
            const int n1 = tb + s - 10;
            float a_even = c0s * (s < 10 ? xh[s] : xs[s - 10]), a_odd = ao[s];
            float y0 = 0.f, y1 = 0.f;
            if (!EDGE || (n1 >= 0 && n1 < W + q_e)) {
                if (EDGE) {
                    if (n1 == W - 1) fla.a_last = a_odd;
                    if (n1 >= W) a_even = a_odd = fla.a_last;
                }
                y0 = iir_bp<false>(bpf, k.ext, a_even);
                y1 = iir_bp<false>(bpf, k.ext, a_odd);
            }
            if (EDGE && (n1 - q_e < 0 || n1 - q_e >= W)) y0 = y1 = 0.f;
            be[s] = y0;
            bo[s] = y1;
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) xh[j] = xs[kBlk - 10 + j];
        CM_BLK_STAMP(2);
        // ---- dn2 -> e
        float ev[kBlk];
        fb.run(bo, ev, tl, g17, g18, g19, ops, ad, sl);
        sl.next();
        CM_BLK_STAMP(3);
#pragma unroll
        for (int s = 0; s < kBlk; ++s) {
            ev[s] = __builtin_fmaf(c0s, s < 9 ? beh[s] : be[s - 9], ev[s]);
            if (EDGE) {
                const int n3 = tb + s - 10 - q_e - 9;
                if (n3 < 0 || n3 >= W) ev[s] = 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) beh[j] = be[kBlk - 9 + j];
        // ---- up2(e): the pair the product detectors multiply
        float mo[kBlk];
        fe.run(ev, mo, tl, g17, g18, g19, ops, ad, sl);
        sl.next();
        CM_BLK_STAMP(4);
        // ---- detectors: products with the phase-free carriers, low-pass on the pair (cos, sin)
        pf2 qe[kBlk];
        float qoc[kBlk], qos[kBlk];
#pragma unroll
        for (int s = 0; s < kBlk; ++s) {
            const int nd = tb + s - front_off;
            const float m_even = c0s * (s < 10 ? eh[s] : ev[s - 10]), m_odd = mo[s];
            int ci = nd;
            if (EDGE) ci = nd < 0 ? 0 : (nd > W - 1 ? W - 1 : nd);
            const f4 c = ((const_f4 *)g.carrier4)[ci];
            pf2 p_e = pk_mul_bs<0>(pf2{m_even, m_even}, pf2{c.x, c.y});
            pf2 p_o = pk_mul_bs<0>(pf2{m_odd, m_odd}, pf2{c.z, c.w});
            pf2 y0 = {0.f, 0.f}, y1 = {0.f, 0.f};
            if (!EDGE || (nd >= 0 && nd < W + q_l)) {
                if (EDGE) {
                    if (nd == W - 1) p_last = p_o;
                    if (nd >= W) p_e = p_o = p_last;
                }
                y0 = iir_sym_pk<0, S::NL>(lpf, kb.lpf, p_e);
                y1 = iir_sym_pk<0, S::NL>(lpf, kb.lpf, p_o);
            }
            if (EDGE && (nd - q_l < 0 || nd - q_l >= W)) y0 = y1 = pf2{0.f, 0.f};
            qe[s] = y0;
            qoc[s] = y1.x;
            qos[s] = y1.y;
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) eh[j] = ev[kBlk - 10 + j];
        CM_BLK_STAMP(5);
        // ---- dn2 of both channels -> the line's base pair
        float rc[kBlk], rs[kBlk];
        fqc.run(qoc, rc, tl, g17, g18, g19, ops, ad, sl);
        sl.next();
        fqs.run(qos, rs, tl, g17, g18, g19, ops, ad, sl);
        sl.next();
        CM_BLK_STAMP(6);
        // ---- back end: comb combination with the neighbouring lane's pair, pre-correction low-pass, re-modulation, matrix.
        // As in the streaming decoder it handles the base pair of the step before (n6 = t - lat_front - 1).  Outputs land in
        // tile registers: positions 0 .. kCarry - 1 come from the previous block.
        pf2 bases[kBlk], nbs[kBlk];
#pragma unroll
        for (int s = 0; s < kBlk; ++s) {
            const pf2 qc = s < 9 ? qeh[s] : qe[s - 9];
            bases[s] = pf2{__builtin_fmaf(c0s, qc.x, rc[s]), __builtin_fmaf(c0s, qc.y, rs[s])};
            nbs[s] = pf2{lane_from(idx1, bases[s].x), lane_from(idx1, bases[s].y)};     // 32 exchanges in flight, one wait
        }
        CM_BLK_STAMP(7);
        float o_r[kBlk], o_g[kBlk], o_b[kBlk];
#pragma unroll
        for (int j = 0; j < kCarry; ++j) { o_r[j] = cr[j]; o_g[j] = cg[j]; o_b[j] = cb[j]; }
#pragma unroll
        for (int s = 0; s < kBlk; ++s) {
            const int t = tb + s;
            const pf2 base = bases[s], nb = nbs[s];
            const int n6 = t - lat_front - 1, n7 = n6 - SP;
            const pf2 uv = back.combine(lk, base_c, nb_c, nb_c);
            base_c = base;
            nb_c = nb;
            int ci = n7;
            if (EDGE) ci = n7 < 0 ? 0 : (n7 > W - 1 ? W - 1 : n7);
            const f2 cb2 = ((const_f2 *)g.carrier2)[ci];
            const pf2 sc = back.remod(lk, pf2{cb2.x, cb2.y});
            const pf2 uv_d = SP > 0 ? uvd[SP > 0 ? SP - 1 : 0] : uv;
            const Rgb<float> o = back.template step<EDGE>(k, kb, lk, uv_last, n6, uv, uv_d, 0.f, sc);   // the luma source joins at the flush
#pragma unroll
            for (int j = SP - 1; j > 0; --j) uvd[j] = uvd[j - 1];
            if (SP > 0) uvd[0] = uv;
            if (s + kCarry < kBlk) { o_r[s + kCarry] = o.r; o_g[s + kCarry] = o.g; o_b[s + kCarry] = o.b; }
            else { cr[s + kCarry - kBlk] = o.r; cg[s + kCarry - kBlk] = o.g; cb[s + kCarry - kBlk] = o.b; }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) qeh[j] = qe[kBlk - 9 + j];
        CM_BLK_STAMP(8);
        // ---- the output tile [tile0, tile0 + 16) of all 64 rows.  The fills of this block have landed before its stores go
        // out, so the next block never waits behind stores.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (!EDGE || (tile0 >= 0 && tile0 < Wp)) {
            f4 xl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xl[q] = *(const lds_f4 *)(ltile + q * 256 + 4 * lane);
            blk_store_plane(g, otile, g.out, orows, o_r, xl, my0, tile0, lane);
            blk_store_plane(g, otile, g.out + g.out_plane_stride, orows, o_g, xl, my1, tile0, lane);
            blk_store_plane(g, otile, g.out + 2 * g.out_plane_stride, orows, o_b, xl, my2, tile0, lane);
        }
        CM_BLK_STAMP(9);
    };

    // one loop, two bodies
#pragma nounroll
    for (int tb = 0; tb < T; tb += kBlk) {
        if (tb >= t_mid0 && tb < t_mid1) block(std::false_type(), tb);
        else block(std::true_type(), tb);
    }
#ifdef CM_BLK_DIAG
    if (blockIdx.x == 4000 && lane == 0)
        printf("blk diag: %d interior blocks; cycles per block: in %llu | fx %llu | bp %llu | fb %llu | fe %llu | det %llu | fq %llu | nb %llu | back %llu | flush %llu\n", diag_n,
               diag_t[0] / diag_n, diag_t[1] / diag_n, diag_t[2] / diag_n, diag_t[3] / diag_n, diag_t[4] / diag_n, diag_t[5] / diag_n, diag_t[6] / diag_n,
               diag_t[7] / diag_n, diag_t[8] / diag_n, diag_t[9] / diag_n);
#endif
}

}  // namespace cm
#endif

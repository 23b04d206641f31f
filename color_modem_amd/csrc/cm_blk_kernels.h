// cm_blk_kernels.h - time-blocked PAL delay-line decoder with the half-band FIRs on the matrix pipe (gfx950).
//
// Same arithmetic, same lane = scan line layout and the same per-sample schedule (stream indices, FilterFunction edges) as
// the streaming decoder of cm_kernels.h / cm_stages.h, but a lane advances its line 32 samples at a time and holds the
// block in registers:
//   * the five 20-tap half-band FIR chains of the path (resample_poly up / down by 2) are Toeplitz products on
//     v_mfma_f32_32x32x16_f16: time on M, the 64 lines of the wave on N in two halves, a 48-sample window on K.  Data and
//     taps are split into two float16 pieces each (hi.hi + hi.lo + lo.hi, float32 accumulation; measured error 2.5e-7 of
//     full scale, the same as the float32 fmaf chain: profiles/r02_ubench_mfma_f16_fir.txt); v_permlane32_swap moves a
//     lane's samples into the operand layout and the results back.  Per chain and 32 samples: 18 MFMAs + about 100 vector
//     instructions instead of 640 v_fma_f32;
//   * the recursive filters walk the block sample by sample with their state in registers (the scalar / packed section
//     code of cm_stages.h / cm_stages_pk.h);
//   * one wavefront per 64 calls and ONE WAVE PER SIMD (the block and every filter state of a line need about 400 VGPRs):
//     no hand-over, no barriers; the matrix pipe runs beside the vector pipe of the same wave;
//   * the luma source sample x[n7] is not carried through the pipeline: the flush of an output tile re-reads the input row
//     segment it covers (128-byte row segments, an L2 hit) and adds it, so no delay ring is needed and the output leaves as
//     128-byte row segments.
// Scales: every FIR runs on taps times kBlkScale (keeps the low float16 pieces of the small taps normal); the centre taps
// carry the same factor, so stage outputs are kBlkScale^n times the streaming decoder's and the factor of the base pair
// (kBlkScale^4) is divided out of the per-line combination coefficients when a lane loads them.
#ifndef CM_BLK_KERNELS_H
#define CM_BLK_KERNELS_H

#include "cm_kernels.h"

namespace cm {

typedef _Float16 blk_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 blk_h2 __attribute__((ext_vector_type(2)));
typedef float blk_f16v __attribute__((ext_vector_type(16)));
typedef unsigned blk_u4 __attribute__((ext_vector_type(4)));

constexpr int kBlk = 32;                 // samples per block
constexpr float kBlkScale = 1.f;         // taps enter the matrix pipe times this (the un-normalised sections of the recursive filters already lift the streams by 10 - 1000: a larger factor overflows float16 at the detector low-pass)

// Toeplitz tiles of the 20-tap FIR y[t] = sum_j g[j] x[t - j] over the window [block start - 16, block start + 32):
// k-step c, lane l (output t = l & 31, half h = l >> 5), element j: window sample s = 16 c + 8 h + j, tap index t + 16 - s.
// [3 k-steps][hi, lo][64 lanes] fragments of 8 float16 = 6 KiB, built by the host (cm_api.hip: build_blk_tiles).
struct BlkTiles {
    blk_h8 hi[3], lo[3];
};

__device__ __forceinline__ blk_h8 blk_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
    blk_u4 v = {a, b, c, d};
    return __builtin_bit_cast(blk_h8, v);
}
// two samples -> packed float16 high and low pieces
__device__ __forceinline__ void blk_split2(float x0, float x1, unsigned &hi, unsigned &lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    blk_h2 ph = {h0, h1}, pl = {l0, l1};
    hi = __builtin_bit_cast(unsigned, ph);
    lo = __builtin_bit_cast(unsigned, pl);
}

// State of one FIR chain between blocks: the last k-step of the previous block in operand form and its samples 13 .. 15
// (the 6 products that reach behind the 48-sample window are added on the vector pipe).
struct BlkFir {
    unsigned hi[8], lo[8];
    float p13, p14, p15;
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < 8; ++i) hi[i] = lo[i] = 0u;
        p13 = p14 = p15 = 0.f;
    }
    // in[32] -> out[32] = kBlkScale * (FIR of the stream); g17 .. g19: the last three taps times kBlkScale
    __device__ __forceinline__ void run(const float (&in)[kBlk], float (&out)[kBlk], const BlkTiles &tl, float g17, float g18, float g19) {
        unsigned nhi[16], nlo[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) blk_split2(in[2 * i], in[2 * i + 1], nhi[i], nlo[i]);
        // operand form of the two new k-steps: registers [8 c .. 8 c + 3] <- lines 0-31, [8 c + 4 .. 8 c + 7] <- lines 32-63
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                auto r = __builtin_amdgcn_permlane32_swap(nhi[8 * c + i], nhi[8 * c + 4 + i], false, false);
                nhi[8 * c + i] = r[0]; nhi[8 * c + 4 + i] = r[1];
                auto q = __builtin_amdgcn_permlane32_swap(nlo[8 * c + i], nlo[8 * c + 4 + i], false, false);
                nlo[8 * c + i] = q[0]; nlo[8 * c + 4 + i] = q[1];
            }
        blk_f16v dl, dh;
#pragma unroll
        for (int i = 0; i < 16; ++i) dl[i] = dh[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned *wh = c == 0 ? hi : nhi + 8 * (c - 1), *wl = c == 0 ? lo : nlo + 8 * (c - 1);
            const blk_h8 l_hi = blk_frag(wh[0], wh[1], wh[2], wh[3]), l_lo = blk_frag(wl[0], wl[1], wl[2], wl[3]);
            const blk_h8 u_hi = blk_frag(wh[4], wh[5], wh[6], wh[7]), u_lo = blk_frag(wl[4], wl[5], wl[6], wl[7]);
            dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl.lo[c], l_hi, dl, 0, 0, 0);
            dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl.lo[c], u_hi, dh, 0, 0, 0);
            dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl.hi[c], l_lo, dl, 0, 0, 0);
            dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl.hi[c], u_lo, dh, 0, 0, 0);
            dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl.hi[c], l_hi, dl, 0, 0, 0);
            dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl.hi[c], u_hi, dh, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (clang folds __builtin_bit_cast of a vector ELEMENT expression to element 0: go through scalars)
            const float lo_f = dl[r], hi_f = dh[r];
            auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_f), __float_as_uint(hi_f), false, false);
            const int t = (r & 3) + 8 * (r >> 2);
            out[t] = __uint_as_float(s[0]);
            out[t + 4] = __uint_as_float(s[1]);
        }
        out[0] = __builtin_fmaf(g17, p15, __builtin_fmaf(g18, p14, __builtin_fmaf(g19, p13, out[0])));
        out[1] = __builtin_fmaf(g18, p15, __builtin_fmaf(g19, p14, out[1]));
        out[2] = __builtin_fmaf(g19, p15, out[2]);
#pragma unroll
        for (int i = 0; i < 8; ++i) { hi[i] = nhi[8 + i]; lo[i] = nlo[8 + i]; }
        p13 = in[13]; p14 = in[14]; p15 = in[15];
    }
};

struct BlkArgs {
    Geom g;
    const BlkTiles *tiles;      // [64 lanes]
};

// Row-wise read-back of the 32-sample output tile: 8 rows x 128 bytes per wave-instruction, with the luma source row segment
// of the same pixels re-read from the input and added (m_y[p] = luma column of the colour matrix).
__device__ __forceinline__ void blk_flush(const Geom &g, const lds_float *otile, const float *op, const float *xp, int first_col, int lane,
                                          float my0, float my1, float my2) {
    __builtin_amdgcn_wave_barrier();
    constexpr int kTile = kBlk, kChunks = kTile / 4, kRows = 64 / kChunks;
    const int chunk = lane & (kChunks - 1);
    const int col = first_col + 4 * chunk;
#pragma nounroll
    for (int q = 0; q < kChunks; ++q) {
        const int row = lane / kChunks + kRows * q;
        typedef __attribute__((address_space(1))) f4 global_f4;
        global_f4 *dst = (global_f4 *)(unsigned long long)ptr_from(row * 4, op);
        const f4 *src = (const f4 *)ptr_from(row * 4, xp);
        const int quad = chunk ^ ((row >> CM_TILE_SWZ) & (kChunks - 1));
        if (dst != nullptr && col < g.Wp) {
            const f4 x = src[col >> 2];
            dst += col >> 2;
            f4 v0 = *(const lds_f4 *)(otile + 0 * 64 * kTile + row * kTile + 4 * quad);
            f4 v1 = *(const lds_f4 *)(otile + 1 * 64 * kTile + row * kTile + 4 * quad);
            f4 v2 = *(const lds_f4 *)(otile + 2 * 64 * kTile + row * kTile + 4 * quad);
            v0 += my0 * x;
            v1 += my1 * x;
            v2 += my2 * x;
            __builtin_nontemporal_store(v0, &dst[0]);
            __builtin_nontemporal_store(v1, &dst[g.out_plane_stride >> 2]);
            __builtin_nontemporal_store(v2, &dst[(2 * g.out_plane_stride) >> 2]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// S: the tuned filter shape (even shifts of the band-pass and the detector low-pass: PAL-BG).  PAL-D front end, one line of
// history (the comb), plain luma strip by re-modulation; no notch, no minavg, float32 rows whose width is a multiple of 4.
// QE, QL: the pair delays of the band-pass and of the detector low-pass (compile-time here: the aligned flush of the output
// tile falls on one fixed sample of a block; the host checks them against the plan).
template <class S, int QE, int QL>
__global__ __launch_bounds__(64, 1) void demod_blk_kernel(const PassArgs<S> args, const BlkTiles *tiles) {
    static_assert(!S::ODD_E && !S::ODD_L && !S::RT, "the blocked decoder is built for the tuned even-shift shapes");
    typedef DemodK<float, S> K;
    constexpr int SP = S::SP, DEPTH = 1;
    __shared__ __attribute__((aligned(16))) float lds_store[64 * kBlk + 3 * 64 * kBlk];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + 64 * kBlk;
    const Geom &g = args.g;
    const K &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
    // luma source row of the flush: the own row, or the previous call's where the plan says so (cm_kernels.h: run_pair)
    const float *lp = g.in + lc.frame * g.in_frame_stride +
                      (long long)(((g.luma_prev_bits >> lc.regime) & 1) ? lc.prev_row : lc.src_row) * g.Wp;
    const float *op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
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
    lds_float *otile = otile_base + lane * kBlk;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kBlk / 4 - 1)) << 2;

    // ---- stream geometry (cm_kernels.h: run_pair) --------------------------------------------------------------------------
    const int W = g.W, Wp = g.Wp;
    constexpr int q_e = QE, q_l = QL;
    constexpr int front_off = 10 + q_e + 9 + 10;        // detector pair index nd = t - front_off
    constexpr int lat_front = front_off + q_l + 9;
    constexpr int lat_out = lat_front + 1 + SP;         // n7 = t - lat_out
    const int T = (Wp + lat_out + kBlk - 1) & ~(kBlk - 1);
    // interior blocks: tb >= lat_out keeps every stage index >= 0, tb + 31 <= W - 2 keeps them below every end-of-row latch
    int t_mid0 = (lat_out + kBlk - 1) & ~(kBlk - 1), t_mid1 = W >= kBlk + 1 ? ((W - kBlk - 1) & ~(kBlk - 1)) + kBlk : 0;
    if (t_mid1 < t_mid0) t_mid0 = t_mid1 = 0;          // short rows: the guarded body runs everything

    const lds_float *xrow = itile + lane * kBlk;
    fill_tile<kBlk>(g, itile, xp, 0, lane);

    auto block = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        // ---- this block's input samples; the tile is refilled for the next block right away
        float xs[kBlk];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        if (tb + kBlk < W) fill_tile<kBlk>(g, itile, xp, (tb >> 5) + 1, lane);

        // ---- up2(x): odd phase on the matrix pipe, even phase = centre tap of x[t - 10]
        float ao[kBlk];
        fx.run(xs, ao, tl, g17, g18, g19);
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
        // ---- dn2 -> e
        float ev[kBlk];
        fb.run(bo, ev, tl, g17, g18, g19);
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
        fe.run(ev, mo, tl, g17, g18, g19);
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
        // ---- dn2 of both channels -> the line's base pair
        float rc[kBlk], rs[kBlk];
        fqc.run(qoc, rc, tl, g17, g18, g19);
        fqs.run(qos, rs, tl, g17, g18, g19);
        // ---- back end: comb combination with the neighbouring lane's pair, pre-correction low-pass, re-modulation, matrix.
        // As in the streaming decoder it handles the base pair of the step before (n6 = t - lat_front - 1).
#pragma unroll
        for (int s = 0; s < kBlk; ++s) {
            const int t = tb + s;
            const pf2 qc = s < 9 ? qeh[s] : qe[s - 9];
            const pf2 base = pf2{__builtin_fmaf(c0s, qc.x, rc[s]), __builtin_fmaf(c0s, qc.y, rs[s])};
            const pf2 nb = pf2{lane_from(idx1, base.x), lane_from(idx1, base.y)};
            const int n6 = t - lat_front - 1, n7 = n6 - SP;
            const pf2 uv = back.combine(lk, base_c, nb_c, nb_c);
            base_c = base;
            nb_c = nb;
            int ci = n7;
            if (EDGE) ci = n7 < 0 ? 0 : (n7 > W - 1 ? W - 1 : n7);
            const f2 cb = ((const_f2 *)g.carrier2)[ci];
            const pf2 sc = back.remod(lk, pf2{cb.x, cb.y});
            const pf2 uv_d = SP > 0 ? uvd[SP > 0 ? SP - 1 : 0] : uv;
            const Rgb<float> o = back.template step<EDGE>(k, kb, lk, uv_last, n6, uv, uv_d, 0.f, sc);   // the luma source joins at the flush
#pragma unroll
            for (int j = SP - 1; j > 0; --j) uvd[j] = uvd[j - 1];
            if (SP > 0) uvd[0] = uv;
            if (!EDGE || (n7 >= 0 && n7 < W)) put_rgb<false, kBlk>(otile, wpos, n7, o);
            if (((s - lat_out) & (kBlk - 1)) == kBlk - 1 && n7 >= 0)      // one fixed sample of a block completes an output tile
                blk_flush(g, otile_base, op, lp, n7 & ~(kBlk - 1), lane, my0, my1, my2);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) qeh[j] = qe[kBlk - 9 + j];
    };

    // one loop, two bodies (each body is some 5000 instructions: a third copy would not fit the instruction cache at all)
#pragma nounroll
    for (int tb = 0; tb < T; tb += kBlk) {
        if (tb >= t_mid0 && tb < t_mid1) block(std::false_type(), tb);
        else block(std::true_type(), tb);
    }
    if (Wp & (kBlk - 1)) blk_flush(g, otile_base, op, lp, Wp & ~(kBlk - 1), lane, my0, my1, my2);     // the row's last, partial tile
}

}  // namespace cm
#endif

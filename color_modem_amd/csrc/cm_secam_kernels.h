// cm_secam_kernels.h - device-side lane drivers of the SECAM decoder and encoder (gfx950).
//
// Same execution model and data movement as cm_kernels.h (one lane = one scan line, input rows
// staged through a 64 x 32-sample LDS tile filled by global_load_lds, outputs transposed through an
// LDS tile into 64-byte row segments).  Differences:
//   * the chroma stream starts with the mirrored pre-roll cc[m] = x[P - m], m < P (secam.py:283-284),
//     read sample by sample out of the first input tile before the aligned main loop starts;
//   * luma (band-stop at 1x rate) runs on a second, delayed visit of the row so that it meets the
//     FM-decoded colour-difference sample of the same pixel without a delay line;
//   * the previous line's colour-difference signal comes from the neighbouring lane.
#ifndef CM_SECAM_KERNELS_H
#define CM_SECAM_KERNELS_H

#include "cm_kernels.h"
#include "cm_mod_kernels.h"

namespace cm {

struct SecamDemodArgs {
    Geom g;                    // g.lanes -> SecamDemodLaneK<float> table, g.carrier4 -> FM reference {cos, sin} pairs,
                               // g.carrier2 -> dc table (cm_plan.h: build_fm_dc)
    SecamDemodK<float> k;
};

// U8: the ImageModem byte boundary fused in, as in the PAL / NTSC decoders (cm_kernels.h: PassCfg::U8): composite bytes
// enter through (5 (byte / 255) - 1) / 3, interleaved RGB bytes leave; the tiles hold bytes and the strides count bytes.
template <bool U8>
__global__ __launch_bounds__(64, 2) void secam_demod_kernel(const SecamDemodArgs args) {
    constexpr int kTile = 16, DEPTH = 1;
    __shared__ __attribute__((aligned(16))) float lds_store[U8 ? (kLdsIn + 3 * 64 * kTile) / 4 : kLdsIn + 3 * 64 * kTile];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + (U8 ? kLdsIn / 4 : kLdsIn);
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const Geom &g = args.g;
    SecamDemodK<float> k = args.k;
    typedef SecamDemodPk::VP VP;
    if (VP::VT) pin_block(k.taps);
    if (VP::VB) pin_block(k.bpf, false);
    SecamDemodKPk kp;      // the quadrature low-pass runs on the pair (I, Q) (cm_stages_pk.h)
    kp.load(k);
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *xp, *op;
    if (U8) {
        xp = (const float *)((const unsigned char *)g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.W);
        op = lc.store_ok ? (const float *)((unsigned char *)g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride)
                         : nullptr;
    } else {
        xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
        op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    }
    SecamDemodLaneK<float> lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamDemodLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    SecamDemodPk st;
    st.reset();
    float chw[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) chw[j] = 0.f;
    lds_float *otile = U8 ? (lds_float *)((lds_u8 *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const lds_float *xrow = itile + lane * kInTile;
    const lds_u8 *xrow8 = (const lds_u8 *)itile + lane * kInTile;

    const int W = g.W, P = k.preroll, Lc = W + P;
    const int lat = SecamDemod<float>::latency(k);          // chroma sample n = m - lat (the packed form keeps the schedule)
    const int d_luma = lat + 1 - P - k.s_y;                 // luma filter input x[xi - d_luma] at main-loop sample xi
    const int lat_out = d_luma + k.s_y;                     // output sample n' = xi - lat_out
    const int s_flush = (lat_out + 3) & 3;
    float own_prev = 0.f, nb_prev = 0.f;

    // one step of the stream: cc = cc[m]; x_l = x[n' + s_y] for the luma filter
    auto step = [&](int m, float cc, float x_l, int sub) {
        int m2 = m - k.s_b - 10;
        m2 = m2 < 0 ? 0 : (m2 > Lc - 1 ? Lc - 1 : m2);
        f4 c = ((const_f4 *)g.carrier4)[m2];
        int m4 = m - lat + P;                               // row-stream sample the decimator completes in this step
        m4 = m4 < 0 ? 0 : (m4 > Lc - 1 ? Lc - 1 : m4);
        const float dc = ((const __attribute__((address_space(4))) float *)g.carrier2)[m4];
        float ch_out;
        float own = st.chroma_step(k, kp, lk, m, cc, chw[sub], pf2{c.x, c.y}, pf2{c.z, c.w}, dc, ch_out);
        chw[10 + sub] = ch_out;
        const int n = m - 1 - lat;                          // the back end runs one sample behind the exchange
        float luma = st.luma_step(k, n, x_l);
        Rgb<float> o = st.finish(k, lk, luma, own_prev, nb_prev);
        own_prev = own;
        nb_prev = lane_from(idx1, own);
        if (n >= 0 && n < W) put_rgb<U8, kTile>(otile, wpos, n, o);
    };
    auto shift_window = [&]() {
#pragma unroll
        for (int j = 0; j < 10; ++j) chw[j] = chw[j + 4];
    };

    if (U8) fill_tile_u8(g, itile, xp, 0, lane); else fill_tile(g, itile, xp, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- pre-roll: cc[m] = x[P - m] for m < P, in bodies of 4 steps so that the windows keep their phase
    const int m_start = -((4 - (P & 3)) & 3);
    for (int mb = m_start; mb < P; mb += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = mb + s;
            float cc = 0.f;
            if (m >= 0) {
                int xi = P - m;
                if (xi > W - 1) xi = W - 1;
                if (P < kInTile)   // the mirrored samples lie in the first input tile
                    cc = U8 ? __builtin_fmaf((float)xrow8[xi], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : xrow[xi];
                else               // long pre-rolls (sampling rates above ~27 MHz): straight from the row, once per line
                    cc = U8 ? __builtin_fmaf((float)((const unsigned char *)xp)[xi], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : xp[xi];
            }
            step(m, cc, 0.f, s);
        }
        shift_window();
    }
    // ---- main loop over the row samples xi = m - P
    auto read_x = [&](int first) -> f4 {
        f4 v;
        if (U8) v = decode_bytes(*(const lds_u32 *)(xrow8 + (first & (kInTile - 1))));
        else v = *(const lds_f4 *)(xrow + (first & (kInTile - 1)));
        if (first + 3 >= W) {
            if (first >= W) v.x = 0.f;
            if (first + 1 >= W) v.y = 0.f;
            if (first + 2 >= W) v.z = 0.f;
            if (first + 3 >= W) v.w = 0.f;
        }
        return v;
    };
    auto read_luma = [&](int first) -> f4 { return load_luma<U8>(xp, first, true, W); };
    const int T = (g.Wp + lat_out + 3) & ~3;
    f4 xv = read_x(0);
    f4 nl = read_luma(-d_luma);
    for (int xb = 0; xb < T; xb += 4) {
        const f4 lw = nl;
        nl = read_luma(xb + 4 - d_luma);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            step(P + xb + s, xv[s], lw[s], s);
            if (s == s_flush) {
                const int n = xb + s - lat_out;
                if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) {
                    if (U8) flush_tile_u8(g, otile_base, op, n & ~(kTile - 1), lane);
                    else flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
                }
            }
        }
        shift_window();
        const int nxt = xb + 4;
        if ((nxt & (kInTile - 1)) == 0 && nxt < W) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        xv = read_x(nxt);
        if ((nxt & (kInTile - 1)) == kInTile - 4 && nxt + 4 < W) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (U8) fill_tile_u8(g, itile, xp, (nxt >> 5) + 1, lane); else fill_tile(g, itile, xp, (nxt >> 5) + 1, lane);
        }
    }
}

struct SecamModArgs {
    Geom g;                    // g.lanes -> SecamModLaneK<float, double> table
    SecamModK<float, double> k;
};

// SP = shift of the pre-correction low-pass (register window of the luma delay); DEPTH = 1: line averaging
// RT: SP is the size of the luma delay window, the delay itself is k.s_p <= SP (other sampling rates)
template <int SP, int DEPTH, bool U8 = false, bool RT = false>
__global__ __launch_bounds__(64, 2) void secam_mod_kernel(const SecamModArgs args) {
    constexpr int kTile = kSecamModTile;
    __shared__ __attribute__((aligned(16))) float lds_store[U8 ? kModLdsFloatsU8 : mod_lds_floats<kTile>()];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + (U8 ? kInTile3Bytes / 4 : kLdsIn3);
    const Geom &g = args.g;
    const SecamModK<float, double> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *rp, *op;
    mod_rows<U8>(g, lc, rp, op);
    SecamModLaneK<float, double> lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamModLaneK<float, double> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    SecamMod<float, double> st;
    st.reset();
    float yw[SP + 4];
#pragma unroll
    for (int j = 0; j < SP + 4; ++j) yw[j] = 0.f;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int sp = RT ? k.s_p : SP;
    const int T = (g.Wp + sp + 3) & ~3;
    f4 cur[3], nxt[3];
    first_tile3<U8>(g, itile, rp, lane, nxt);
    for (int tb = 0; tb < T; tb += 4) {
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        next_tile3x<U8>(g, itile, rp, lane, tb + 4, nxt);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
            float y = fmaf_(k.e[0][0], r, fmaf_(k.e[0][1], gg, k.e[0][2] * b));
            float dr = fmaf_(k.e[1][0], r, fmaf_(k.e[1][1], gg, k.e[1][2] * b));
            float db = fmaf_(k.e[2][0], r, fmaf_(k.e[2][1], gg, k.e[2][2] * b));
            float d = lk.own_is_db != 0.f ? db : dr;     // secam.py:262-271 (the line being modulated decides)
            if (DEPTH >= 1) {
                // the previous call's components: its lane computed them with its own parity, so exchange both
                float yp = lane_from(idx1, y), drp = lane_from(idx1, dr), dbp = lane_from(idx1, db);
                float dp = lk.own_is_db != 0.f ? dbp : drp;
                y = fmaf_(lk.wy0, y, lk.wy1 * yp);
                d = fmaf_(lk.wc0, d, lk.wc1 * dp);
            }
            yw[SP + s] = y;
            const int n7 = t - sp;
            float y_d = yw[s];                 // luma of sample t - SP
            if (RT) {                          // ... t - s_p: uniform selects instead of a dynamic register index
#pragma unroll
                for (int j = 0; j < SP; ++j)
                    if (sp == j) y_d = yw[SP - j + s];
            }
            float comp = st.step(k, lk, t, y_d, d);
            put_composite<U8, kTile>(g, otile_base, op, lane, wpos, n7, comp);
        }
#pragma unroll
        for (int j = 0; j < SP; ++j) yw[j] = yw[j + 4];
    }
}

}  // namespace cm
#endif

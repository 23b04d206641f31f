// cm_wrap_kernels.h - SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem (ref comb.py:96-113 on top of
// pal.py:79-234).  These stacks mix the two front ends on the second line of every run (call 0 is the plain decode, call 1
// averages it with the first delay-line decode) or reach back three lines, so they do not fit the per-line coefficient
// tables of the fused decoders.  They run as TWO streaming kernels per batch instead (+ the sparse first-line pass):
//
//   1. the inner decoder's kernel in component mode with strip_chroma = False - exactly what the wrapper asks its backend
//      for (comb.py:97, 101) - over ALL calls of ALL runs of the batch, results in call order ([frame][call][3][W] scratch:
//      Geom::out_calls); call 0 of every run comes from the backend's plain decoder (comb.py:48-49) as a sparse pass;
//   2. comb_wrap_back_kernel below: one lane = one call, lanes = consecutive calls (the previous call's components are the
//      neighbouring lane's), streaming along the row like the modulators:
//        (u, v) = avg / minavg(last, curr), luma source = last / curr          comb.py:102-104
//        y -= backend.modulate_components(frame, line - 2 own_delay, 0, u, v)   comb.py:105-107  (QamModCore, the inner
//                                                                               modulator's pre-correction filter + carrier)
//        y = notch(y)                                                           comb.py:108-110  (one biquad, zero state)
//        decode_components                                                      comb.py:121-122
//      and the output tile / row-segment stores of the decoders, straight into the caller's frames.
// Round 2 ran this as a Python loop over fields with five launches and four scratch tensors each (5 workgroups per launch).
#ifndef CM_WRAP_KERNELS_H
#define CM_WRAP_KERNELS_H

#include "cm_kernels.h"
#include "cm_mod_kernels.h"

namespace cm {

template <int NP>
struct WrapBackArgs {
    Geom g;                   // g.lanes -> the backend modulator's ModLaneK table, g.carrier2 -> its carrier table
    ModK<float, NP> k;        // the backend modulator's pre-correction filter (its matrix is not used)
    SosK<float, 1> notch;
    float notch_gain;         // 0: no notch
    float m[9];               // decode matrix (identity in component mode)
    int own_delay;            // 1: luma source = the previous call's luma, re-modulation at line - 2 (comb.py:102, 105)
    int minavg;               // 1: comb.py:13-15 instead of comb.py:9-10; 2: (u, v) of the component buffer are final (avg= callables, averaged by the caller)
    int strip;                // 0: demodulate_components(strip_chroma = False)
};

// SP: shift of the pre-correction low-pass (register windows of the delayed signals); RT: SP is the window size, the delay
// itself is k.s_p <= SP (other sampling rates).  U8: interleaved RGB bytes leave (image.py:7-8, 84).  MINAVG / NOTCH: the
// wrapper's avg=comb.minavg / notch= (compile-time: the interior bodies carry no wave-uniform branch for them).
template <int NP, int SP, bool U8 = false, bool RT = false, bool MINAVG = false, bool NOTCH = false>
__global__ __launch_bounds__(64, 2) void comb_wrap_back_kernel(const WrapBackArgs<NP> args) {
    constexpr int kTile = 16, DEPTH = 1;
    __shared__ __attribute__((aligned(16))) float lds_store[kLdsIn3 + (U8 ? 64 * 3 * kTile / 4 : 3 * 64 * kTile)];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + kLdsIn3;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const Geom &g = args.g;
    const ModK<float, NP> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * row_stride;
    const float *op;
    if (U8) op = lc.store_ok ? (const float *)((unsigned char *)g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride) : nullptr;
    else op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    const bool first = lc.kk == 0;                       // call 0 of a run: (y, u, v) = curr, not stripped (comb.py:97-99)
    const bool strip = args.strip != 0 && !first;
    const float strip_f = strip ? 1.f : 0.f;
    const bool take_prev_y = args.own_delay != 0 && !first;
    ModLaneK<float> lk;
    {
        int lm = lc.line - 2 * args.own_delay;           // the line the wrapper re-modulates at; unused where first
        if (lm < 0) lm &= 1;
        const int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const ModLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lm];
        float rc, rs;
        if (frame_turn(g, lc.frame, rc, rs)) {
            turn(lk.cph, lk.sph, rc, rs);
            turn(lk.vcph, lk.vsph, rc, rs);
        }
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    IirState<float, NP> pre_u, pre_v;
    pre_u.reset();
    pre_v.reset();
    float u_last = 0.f, v_last = 0.f;
    IirState<float, 1> notch;
    notch.reset();
    float yw[SP + 4], uw[SP + 4], vw[SP + 4];
#pragma unroll
    for (int j = 0; j < SP + 4; ++j) yw[j] = uw[j] = vw[j] = 0.f;
    lds_float *otile = U8 ? (lds_float *)((lds_u8 *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int sp = RT ? k.s_p : SP;
    // RT: which of the SP window positions is the delay - as opaque scalars: compared against the literals 0 .. SP - 1, hipcc turns the
    // select chain below back into a dynamic index and the three windows into 144 bytes of scratch memory per lane
    int sel[SP];
#pragma unroll
    for (int j = 0; j < SP; ++j) {
        sel[j] = j;
        if (RT) asm volatile("" : "+s"(sel[j]));
    }
    const int T = (g.Wp + sp + 3) & ~3;
    const int s_flush = (sp + 3) & 3;                    // the step of a body whose output sample n7 = t - sp ends a quad
    // interior bodies: 0 <= n7 and t < W - 1 for the four steps - no zero-extension, no latch, every output inside the row
    int tb_mid0 = (sp + 3) & ~3, tb_mid1 = (W - 4) & ~3;
    if (tb_mid1 <= tb_mid0) tb_mid0 = tb_mid1 = 0;
    f4 cur[3], nxt[3];
    first_tile3<false>(g, itile, rp, lane, nxt);
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        next_tile3x<false>(g, itile, rp, lane, tb + 4, nxt);
        const const_f2 *carp = (const_f2 *)g.carrier2 + (tb - sp);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            const float cy = cur[0][s], cu = cur[1][s], cv = cur[2][s];
            const float py = lane_from(idx1, cy), pu = lane_from(idx1, cu), pv = lane_from(idx1, cv);
            float u = MINAVG ? minavg_(pu, cu) : 0.5f * (pu + cu);
            float v = MINAVG ? minavg_(pv, cv) : 0.5f * (pv + cv);
            if (first || args.minavg == 2) { u = cu; v = cv; }
            const float ys = take_prev_y ? py : cy;
            yw[SP + s] = ys; uw[SP + s] = u; vw[SP + s] = v;
            const int n7 = t - sp;
            f2 cc;
            if (EDGE) cc = ((const_f2 *)g.carrier2)[n7 < 0 ? 0 : (n7 > W - 1 ? W - 1 : n7)];
            else cc = carp[s];
            float y_d = yw[s], u_d = uw[s], v_d = vw[s];          // the signals at n7 = t - SP
            if (RT) {                                              // ... = t - s_p: uniform selects instead of a dynamic index
#pragma unroll
                for (int j = 0; j < SP; ++j)
                    if (sp == sel[j]) { y_d = yw[SP - j + s]; u_d = uw[SP - j + s]; v_d = vw[SP - j + s]; }
            }
            // backend.modulate_components(frame, line - 2 own_delay, 0, u, v) (QamModCore::step, cm_stages.h)
            float wu = 0.f, wv = 0.f;
            if (!EDGE || (t >= 0 && t < W + sp)) {
                if (EDGE) {
                    if (t == W - 1) { u_last = u; v_last = v; }
                    if (t >= W) { u = u_last; v = v_last; }
                }
                wu = iir_gen<false>(pre_u, k.pre, u);
                wv = iir_gen<false>(pre_v, k.pre, v);
            }
            const float sn = fmaf_(lk.sph, cc.x, lk.cph * cc.y);
            const float cs = fmaf_(lk.vcph, cc.x, -(lk.vsph * cc.y));
            const float remod = fmaf_(sn, wu, cs * wv);
            float y = fmaf_(-strip_f, remod, y_d);
            if (NOTCH) {     // luma[0 .. W) from a zero state (FilterFunction, shift 0), where the wrapper strips (comb.py:108-110)
                if (!EDGE || (n7 >= 0 && n7 < W)) {
                    const float yn = iir_sym<false>(notch, args.notch, y) * args.notch_gain;
                    if (strip) y = yn;
                }
            }
            Rgb<float> o;
            o.r = fmaf_(args.m[0], y, fmaf_(args.m[1], u_d, args.m[2] * v_d));
            o.g = fmaf_(args.m[3], y, fmaf_(args.m[4], u_d, args.m[5] * v_d));
            o.b = fmaf_(args.m[6], y, fmaf_(args.m[7], u_d, args.m[8] * v_d));
            if (!EDGE || (n7 >= 0 && n7 < W)) put_rgb<U8, kTile>(otile, wpos, n7, o);
            if (s == s_flush && n7 >= 0 && ((n7 & (kTile - 1)) == kTile - 1 || n7 == g.Wp - 1)) {
                if (U8) flush_tile_u8(g, otile_base, op, n7 & ~(kTile - 1), lane);
                else flush_tile<kTile>(g, otile_base, op, n7 & ~(kTile - 1), lane);
            }
        }
#pragma unroll
        for (int j = 0; j < SP; ++j) { yw[j] = yw[j + 4]; uw[j] = uw[j + 4]; vw[j] = vw[j + 4]; }
    };
    int tb = 0;
    for (; tb < tb_mid0; tb += 4) body(std::true_type(), tb);
    for (; tb < tb_mid1; tb += 4) body(std::false_type(), tb);
    for (; tb < T; tb += 4) body(std::true_type(), tb);
}

// composite bytes -> level-decoded float rows (image.py:24-25, 62): the wrapped combs' byte entry point decodes the frame once
// (the inner decoder's component output has no byte form), 1 + 4 bytes per pixel
// (the top `quads_per_frame` quads of frames in_frame_bytes apart -> compact float frames: the whole frame when the two agree)
__global__ __launch_bounds__(256) void decode_level_kernel(const unsigned char *in, float *out, long long n_quads, long long quads_per_frame,
                                                           long long in_frame_bytes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_quads) return;
    const long long frame = i / quads_per_frame, q = i - frame * quads_per_frame;
    ((f4 *)out)[i] = decode_bytes(*(const unsigned *)(in + frame * in_frame_bytes + 4 * q));
}

}  // namespace cm
#endif

// cm_am_stages.h - per-scanline streaming stages of the amplitude-modulated line-sequential standards:
// Proto-SECAM 1957 (ref color_modem/color/protosecam.py:27-112) and NIIR / SECAM-IV (ref color_modem/color/niir.py:10-202).
//
// Same execution model as cm_stages.h: ONE LANE OWNS ONE SCANLINE and walks it one 1x-rate sample per step; every filter of
// the reference is a streaming recurrence whose state lives in the lane's registers.  What is new here is the rate: these
// decoders run their recursive filters at THREE times the sampling rate, between scipy.signal.resample_poly(x, 3, 1) and
// resample_poly(x, 1, 3) (61-tap Kaiser(5) FIR, h = firwin(61, 1 / 3)):
//   * Up3: transposed-form polyphase interpolator - one input sample updates 58 partial sums and completes three outputs
//     (61 FMAs, no register moves);
//   * Dn3: transposed-form decimator - one triple updates the partial sums of the 20 pending outputs and completes one;
//   * FF3 / FF1: utils.FilterFunction.__call__ (utils.py:28-36) around a cascade of second-order sections at the 3x / 1x
//     rate: the input sequence is extended by `shift` copies of its last sample, the first `shift` outputs are dropped -
//     index bookkeeping on the stream (a triple of delay per 3 samples of shift plus a hold of up to two samples).
// The header compiles for the device (T = float, hipcc) and for the host (T = double / float, g++: tests/sim/cm_sim_am.cpp
// checks the schedule against the numpy oracle without a GPU; the product never runs it).
#ifndef CM_AM_STAGES_H
#define CM_AM_STAGES_H

#include "cm_stages.h"

namespace cm {

constexpr int kAmTaps = 61;       // firwin(2 * 10 * 3 + 1, 1 / 3)
constexpr int kAmHalf = 10;       // the FIRs delay a stream by 10 samples of the 1x rate (30 of the 3x rate)

template <typename T>
struct Taps3 {
    T h[kAmTaps];
};
constexpr int tap3i(int i) { return i <= 30 ? i : 60 - i; }      // h[i] = h[60 - i] (checked by the host: cm_am_plan.h: taps3_third_band)

// firwin(61, 1 / 3) is a THIRD-BAND filter: h[30 + 3 k] = sinc(k) = 0 for k != 0 (numerically ~1e-17; the host refuses a tap set where
// they are not below 1e-12 of the centre tap, cm_am_plan.h: taps3_third_band).  The streaming forms below never touch those 20 taps
// (round 4): the interpolator's phase 0 is the input delayed by 10 samples times 3 h[30], the decimator skips the samples z[3 q] but one.

// resample_poly(x, 3, 1): y[m] = sum_i 3 h[i] xu[m + 30 - i], xu[3 n] = x[n].  push(x[t], x[t - 10]) completes y[3 (t - 10) + j], j = 0..2.
// s(e) = partial sum of y[3 t - 30 + e] from the inputs before x[t], kept for e = 1, 2 mod 3 only (e = 3 i + r -> s[2 i + r - 1]); k.h holds 3 h.
template <typename T>
struct Up3 {
    T s[38];
    CM_HD void reset() {
#pragma unroll
        for (int e = 0; e < 38; ++e) s[e] = T(0);
    }
    static constexpr int slot(int e) { return 2 * (e / 3) + (e % 3) - 1; }
    // (fma3: the three-address form - with the two-address v_fmac hipcc rotates the partial sums through one v_mov each; V: the taps sit
    // in vector registers (pin_taps3) or in scalar ones; tap3i: h is symmetric, the device keeps 21 values, not 41)
    template <bool V = true>
    CM_HD void push(const Taps3<T> &k, T x, T x_del, T out[3]) {
        out[0] = k.h[30] * x_del;
        out[1] = fma3<V>(k.h[1], x, s[slot(1)]);
        out[2] = fma3<V>(k.h[2], x, s[slot(2)]);
#pragma unroll
        for (int e = 4; e < 58; ++e)
            if (e % 3 != 0) s[slot(e - 3)] = fma3<V>(k.h[tap3i(e)], x, s[slot(e)]);
        s[slot(55)] = k.h[2] * x;
        s[slot(56)] = k.h[1] * x;
    }
};

// resample_poly(z, 1, 3): y[n] = sum_i h[i] z[3 n + 30 - i].  push(z[3 q .. 3 q + 2]) completes y[q - 10] (times 3 when
// given the interpolator's taps 3 h, as the stages below do).
// s[d] = partial sum of y[q - 10 + d] from the triples before q.
template <typename T>
struct Dn3 {
    T s[20];
    CM_HD void reset() {
#pragma unroll
        for (int d = 0; d < 20; ++d) s[d] = T(0);
    }
    // Every pending output takes its products in the order z[2], z[1] (, z[0]: the centre tap only); the passes below run that
    // order across all of them, so that no instruction reads the result of the one right before it (three-address form: no v_mov).
    template <bool V = true>
    CM_HD T push(const Taps3<T> &k, const T z[3]) {
        const T out = s[0];
#pragma unroll
        for (int d = 1; d < 20; ++d) s[d] = fma3<V>(k.h[tap3i(3 * d - 2)], z[2], s[d]);
        const T last = k.h[2] * z[2];
        s[10] = fma3<V>(k.h[30], z[0], s[10]);
#pragma unroll
        for (int d = 1; d < 20; ++d) s[d - 1] = fma3<V>(k.h[tap3i(3 * d - 1)], z[1], s[d]);
        s[19] = fma3<V>(k.h[1], z[1], last);
        return out;
    }
};

enum { AM_FORM_BP = 0, AM_FORM_SYM = 1, AM_FORM_GEN = 2 };

template <int FORM, bool V = false, typename T, int NSEC>
CM_HD T am_iir(IirState<T, NSEC> &st, const SosK<T, NSEC> &k, T x) {
    if (FORM == AM_FORM_BP) return iir_bp<V>(st, k, x);
    if (FORM == AM_FORM_SYM) return iir_sym<V>(st, k, x);
    return iir_gen<V>(st, k, x);
}

// Delay bookkeeping of one FilterFunction whose sequence runs at `rate` samples per step.
struct FFGeom {
    int32_t shift;   // FilterFunction._shift (>= 0)
    int32_t q;       // steps of delay: ceil(shift / rate)
    int32_t r;       // rate * q - shift: how far the output group straddles the filter's groups (0 .. rate - 1)
};

// FilterFunction at the 3x rate.  Sequence a[m], m in [0, L); the filter runs over m in [0, L + shift) (a[L - 1] repeated
// beyond L) from a zero state, output o[m - shift].  step(n1, in) takes a[3 n1 .. 3 n1 + 2] and returns o[3 (n1 - q) + j],
// zero outside [0, L).
template <typename T, int NSEC>
struct FF3 {
    IirState<T, NSEC> st;
    T last, h1, h2;   // a[L - 1]; the raw outputs Y[3 n1 - 1], Y[3 n1 - 2] of the previous step
    CM_HD void reset() {
        st.reset();
        last = h1 = h2 = T(0);
    }
    // EDGE = false: the caller guarantees 0 <= 3 (n1 - q) and 3 n1 + 2 < L - 1 (the interior of a row: no latch, no clamp,
    // every output inside the sequence)
    // R: g.r known at compile time (0..2; the interior loops of the wave pairs are instantiated per value, so that the choice of
    // registers below costs no instruction), -1: read from g
    template <int FORM, bool EDGE = true, bool V = false, int R = -1>
    CM_HD void step(const SosK<T, NSEC> &k, const FFGeom &g, int L, int n1, const T in[3], T out[3]) {
        T y[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (EDGE) {
                const int m = 3 * n1 + j;
                y[j] = T(0);
                if (m >= 0 && m < L + g.shift) {
                    if (m == L - 1) last = in[j];
                    y[j] = am_iir<FORM, V>(st, k, m < L ? in[j] : last);
                }
            } else {
                y[j] = am_iir<FORM, V>(st, k, in[j]);
            }
        }
        T o0, o1, o2;
        const int r = R >= 0 ? R : g.r;
        if (r == 0) { o0 = y[0]; o1 = y[1]; o2 = y[2]; }
        else if (r == 1) { o0 = h1; o1 = y[0]; o2 = y[1]; }
        else { o0 = h2; o1 = h1; o2 = y[0]; }
        h2 = y[1];
        h1 = y[2];
        if (EDGE) {
            const int p = 3 * (n1 - g.q);
            out[0] = (p >= 0 && p < L) ? o0 : T(0);
            out[1] = (p + 1 >= 0 && p + 1 < L) ? o1 : T(0);
            out[2] = (p + 2 >= 0 && p + 2 < L) ? o2 : T(0);
        } else {
            out[0] = o0; out[1] = o1; out[2] = o2;
        }
    }
};

// FilterFunction at the 1x rate: step(i, a[i]) returns o[i - shift] (zero outside [0, L)).
template <typename T, int NSEC>
struct FF1 {
    IirState<T, NSEC> st;
    T last;
    CM_HD void reset() {
        st.reset();
        last = T(0);
    }
    // EDGE = false: the caller guarantees shift <= i < L - 1
    template <int FORM, bool EDGE = true>
    CM_HD T step(const SosK<T, NSEC> &k, int shift, int L, int i, T in) {
        if (!EDGE) return am_iir<FORM>(st, k, in);
        T y = T(0);
        if (i >= 0 && i < L + shift) {
            if (i == L - 1) last = in;
            y = am_iir<FORM>(st, k, i < L ? in : last);
        }
        const int p = i - shift;
        return (p >= 0 && p < L) ? y : T(0);
    }
};

// =============================================================================================
// Proto-SECAM decoder (ref protosecam.py:92-112).  Streams at step t (x_now = x[t], zero outside the row):
//   n1 = t - 10            triple U(n1) = resample_poly(x, 3, 1)[3 n1 ..]
//   chroma: n2 = n1 - q_e  band-pass (_extract_chroma_up) -> |.| -> n3 = n2 - q_p low-pass (_chroma_up_post_demod_filter)
//           -> resample_poly(., 1, 3): n4 = n3 - 10, so chroma[t - lat_c], lat_c = 20 + q_e + q_p
//   luma:   m2 = n1 - q_r  band-stop (_remove_chroma_up) -> resample_poly(., 1, 3): luma[t - lat_y], lat_y = 20 + q_r
// Every gain (sections, pi / 2, the 8 of protosecam.py:103) is folded into chroma_gain / luma_gain by the host.
// =============================================================================================
template <typename T>
struct ProtoDemodK {
    int32_t width;
    FFGeom ge, gr, gp;
    Taps3<T> taps;               // 3 h: the interpolator's taps; the decimators use them too (their 1 / 3 is in the gains)
    SosK<T, 3> ext;              // band-pass, numerators 1 - z^-2
    SosK<T, 3> rem;              // band-stop, numerators 1 + b1 z^-1 + z^-2
    SosK<T, 2> post;             // low-pass, general numerators (a first-order section when the order is odd)
    T chroma_gain, luma_gain;    // chroma = chroma_gain * stream - 1,  luma = luma_gain * stream
    T m[3][3];                   // (r, g, b) = m . (luma, dr, db)
};

template <typename T>
struct ProtoDemod {
    Up3<T> up;
    FF3<T, 3> ext, rem;
    FF3<T, 2> post;
    Dn3<T> dn_c, dn_y;
    CM_HD void reset() {
        up.reset(); ext.reset(); rem.reset(); post.reset(); dn_c.reset(); dn_y.reset();
    }
    CM_HD static int lat_chroma(const ProtoDemodK<T> &k) { return 2 * kAmHalf + k.ge.q + k.gp.q; }
    CM_HD static int lat_luma(const ProtoDemodK<T> &k) { return 2 * kAmHalf + k.gr.q; }
    // luma = luma[t - lat_luma], chroma = chroma[t - lat_chroma] (meaningful inside the row); x_del = x[t - 10] (zero outside the row)
    template <bool EDGE = true>
    CM_HD void step(const ProtoDemodK<T> &k, int t, T x_now, T x_del, T &luma, T &chroma) {
        const int L = 3 * k.width, n1 = t - kAmHalf;
        T u[3], c1[3], c2[3], y1[3];
        up.push(k.taps, x_now, x_del, u);
        ext.template step<AM_FORM_BP, EDGE>(k.ext, k.ge, L, n1, u, c1);
#pragma unroll
        for (int j = 0; j < 3; ++j) c1[j] = c1[j] < T(0) ? -c1[j] : c1[j];     // protosecam.py:98 (the factor pi / 2 is in chroma_gain)
        post.template step<AM_FORM_GEN, EDGE>(k.post, k.gp, L, n1 - k.ge.q, c1, c2);
        chroma = fmaf_(k.chroma_gain, dn_c.push(k.taps, c2), T(-1));
        rem.template step<AM_FORM_SYM, EDGE>(k.rem, k.gr, L, n1, u, y1);
        luma = k.luma_gain * dn_y.push(k.taps, y1);
    }
};

// =============================================================================================
// Proto-SECAM encoder (ref protosecam.py:74-90).  The caller forms (luma, d) - d the colour-difference signal this line
// carries, after the encoder-side line averaging if any - and feeds the two paths with the delays that make them meet:
//   chroma: d[i] -> pre-correction low-pass (FilterFunction at 1x, shift s_c) -> 0.125 (1 + .): sample i - s_c
//   luma:   luma[i] -> resample_poly(., 3, 1) -> band-stop at 3x -> resample_poly(., 1, 3): sample i - (20 + q_r)
//           (premod_luma_filter off: sample i)
// composite[n] = luma[n] + cos(phi + n step) * chroma[n]                                       protosecam.py:87-90
// =============================================================================================
template <typename T>
struct ProtoModK {
    int32_t width, luma_filter;
    int32_t s_c;                 // shift of the pre-correction low-pass
    FFGeom gr;
    Taps3<T> taps;               // 3 h
    SosK<T, 2> pre;
    SosK<T, 3> rem;
    T pre_gain, luma_gain;
    T e[3][3];                   // (luma, dr, db) = e . (r, g, b)
};

template <typename T>
struct ProtoMod {
    FF1<T, 2> pre;
    Up3<T> up;
    FF3<T, 3> rem;
    Dn3<T> dn;
    CM_HD void reset() {
        pre.reset(); up.reset(); rem.reset(); dn.reset();
    }
    CM_HD static int lat_luma(const ProtoModK<T> &k) { return k.luma_filter ? 2 * kAmHalf + k.gr.q : 0; }
    CM_HD static int lat_chroma(const ProtoModK<T> &k) { return k.s_c; }
    // i_c, d: index and value of the colour-difference sample fed now; i_y, luma likewise.  Returns the filtered luma of
    // sample i_y - lat_luma through luma_out and 0.125 (1 + chroma) of sample i_c - lat_chroma through chroma_out.
    // luma_del: the luma sample fed ten steps ago (index i_y - 10; zero outside the row).
    template <bool EDGE = true>
    CM_HD void step(const ProtoModK<T> &k, int i_c, T d, int i_y, T luma, T luma_del, T &luma_out, T &chroma_out) {
        const T c = pre.template step<AM_FORM_GEN, EDGE>(k.pre, k.s_c, k.width, i_c, d);
        chroma_out = fmaf_(T(0.125) * k.pre_gain, c, T(0.125));
        if (k.luma_filter) {
            T u[3], y1[3];
            up.push(k.taps, (!EDGE || (i_y >= 0 && i_y < k.width)) ? luma : T(0),
                    (!EDGE || (i_y - kAmHalf >= 0 && i_y - kAmHalf < k.width)) ? luma_del : T(0), u);
            rem.template step<AM_FORM_SYM, EDGE>(k.rem, k.gr, 3 * k.width, i_y - kAmHalf, u, y1);
            luma_out = k.luma_gain * dn.push(k.taps, y1);
        } else {
            luma_out = luma;
        }
    }
};

// =============================================================================================
// NIIR / SECAM-IV decoder (ref niir.py:106-164 + _remove_offset :63-67 + decode_components :52-61).
// Streams at step t (x_now = composite[t], zero outside the row); L = 3 W:
//   n1 = t - 10        U(n1) = resample_poly(x, 3, 1)
//   n2 = n1 - q_b      M(n2) = _demodulate_upsampled_filter (band-pass), without its gain
//   n3 = n2 - q_l      S(n3) = _demodulate_upsampled_baseband_filter(|M|), without the gains
//                      P(n3) = phasemod_up = c_pm M(n3) / S(n3) (the caller delays M by q_l steps); the previous call's P
//                      comes from the neighbouring lane, or, on the first line of a run, is the band-passed synthetic
//                      reference carrier of niir.py:107-110 (NiirSyn below)
//   n4 = n3 - 1        the carrier's derivative (niir.py:127-129) needs the next 3x sample: the products are formed one
//                      triple late
//   n5 = n4 - 10       the five decimated streams: sinphi, cosphi, saturation, sincarrier, coscarrier
// so the output sample is n5 = t - lat, lat = 21 + q_b + q_l.
// =============================================================================================
template <typename T>
struct NiirDemodK {
    int32_t width;
    FFGeom gb, gl;
    Taps3<T> taps;               // 3 h
    SosK<T, 3> bp;               // band-pass, numerators 1 - z^-2
    SosK<T, 2> lp;               // low-pass, general numerators
    T c_pm;                      // phasemod_up = c_pm * M / S
    T g_b;                       // gain of the band-pass: the synthetic reference is g_b * M
    T sat_gain;                  // saturation = sat_gain * decimated S
    T alt_scale;                 // altcarrier_up[p] = alt_scale * (carrier_up[p + 1] - carrier_up[p - 1]) = 1.5 / carrier step
    T third;                     // the decimators run on 3 h
    T m[3][3];                   // (r, g, b) = m . (luma, db, dr)
};

// up3 -> band-pass -> |.| -> low-pass of one line
template <typename T>
struct NiirFront {
    Up3<T> up;
    FF3<T, 3> bp;
    FF3<T, 2> lp;
    CM_HD void reset() {
        up.reset(); bp.reset(); lp.reset();
    }
    // m_out = M(n2), n2 = t - 10 - q_b;  s_out = S(n3), n3 = n2 - q_l   (both zero outside [0, L)); x_del = x[t - 10] (zero outside the row)
    // VT / VI: the taps / the sections' coefficients sit in vector registers (the float64 instance: taps scalar, sections vector)
    template <bool EDGE = true, bool VT = true, bool VI = false>
    CM_HD void step(const NiirDemodK<T> &k, int t, T x_now, T x_del, T m_out[3], T s_out[3]) {
        const int L = 3 * k.width, n1 = t - kAmHalf;
        T u[3], a[3];
        up.template push<VT>(k.taps, x_now, x_del, u);
        bp.template step<AM_FORM_BP, EDGE, VI>(k.bp, k.gb, L, n1, u, m_out);
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] = m_out[j] < T(0) ? -m_out[j] : m_out[j];     // niir.py:113 (pi / 2 is in the constants)
        lp.template step<AM_FORM_GEN, EDGE, VI>(k.lp, k.gl, L, n1 - k.gb.q, a, s_out);
    }
};

// the synthetic phase reference of the first line of a run (niir.py:107-110): band-passed +-sin(phi + n step), not normalised
template <typename T>
struct NiirSyn {
    Up3<T> up;
    FF3<T, 3> bp;
    CM_HD void reset() {
        up.reset(); bp.reset();
    }
    template <bool EDGE = true>
    CM_HD void step(const NiirDemodK<T> &k, int t, T x_syn, T x_syn_del, T m_out[3]) {
        T u[3];
        up.push(k.taps, x_syn, x_syn_del, u);
        bp.template step<AM_FORM_BP, EDGE>(k.bp, k.gb, 3 * k.width, t - kAmHalf, u, m_out);
    }
};

// a / b and 1 / sqrt(x).  On the device: v_rcp_f32 / v_rsq_f32 (1 ulp) instead of the IEEE sequences (10 - 12 instructions
// each, six divisions and two square roots per pixel in the NIIR decoder).  Where they are used a relative error of the
// quotient scales a phasor or a (sin, cos) pair as a whole - the angle, which is what the decoder is after, does not see it.
CM_HD float am_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
// float64: the reciprocal from v_rcp_f32 and one Newton step (relative error ~1e-14; the IEEE sequence is ~30 instructions).  A zero or
// infinite divisor keeps the seed (+-inf / 0): a * seed is numpy's inf / nan / 0 there, where the reference divides by zero (niir.py:115)
CM_HD double am_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float rf = __builtin_amdgcn_rcpf((float)b);
    const double r = (double)rf;
    const double rn = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return a * (__builtin_amdgcn_classf(rf, 0x204 | 0x060) ? r : rn);      // class: +-inf | +-0
#else
    return a / b;
#endif
}
CM_HD float am_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / std::sqrt(x);
#endif
}
CM_HD double am_rsqrt(double x) { return 1.0 / std::sqrt(x); }

// phasemod_up of a triple: c_pm * M / S inside the sequence, zero outside (the decimators zero-extend)
template <bool EDGE = true, typename T>
CM_HD void niir_phasemod(const NiirDemodK<T> &k, int n3, const T m[3], const T s[3], T p[3]) {
    const int L = 3 * k.width;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = 3 * n3 + j;
        p[j] = (!EDGE || (q >= 0 && q < L)) ? am_div(k.c_pm * m[j], s[j]) : T(0);
    }
}

template <typename T>
struct NiirOut {
    T sinphi, cosphi, sat, sinc, cosc;    // the decimated streams at sample n5 (saturation with its gain)
};

template <typename T>
struct NiirBack {
    Dn3<T> dn_s, dn_c, dn_sat, dn_sc, dn_cc;
    T c1[3], h1[3], s1[3], c2_2;          // carrier, hue-modulated signal and S of the previous triple; carrier[3 n3 - 4]
    CM_HD void reset() {
        dn_s.reset(); dn_c.reset(); dn_sat.reset(); dn_sc.reset(); dn_cc.reset();
#pragma unroll
        for (int j = 0; j < 3; ++j) c1[j] = h1[j] = s1[j] = T(0);
        c2_2 = T(0);
    }
    // own / prev: phasemod_up triples n3 of this call and of the previous one; s: S(n3); alt: is_alternate_line
    template <bool EDGE = true>
    CM_HD NiirOut<T> step(const NiirDemodK<T> &k, int n3, const T own[3], const T prev[3], const T s[3], bool alt) {
        const int L = 3 * k.width, n4 = n3 - 1;
        T c[3], h[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                       // niir.py:117-124
            c[j] = alt ? own[j] : prev[j];
            h[j] = alt ? prev[j] : own[j];
        }
        T ac[3], su[3], cu[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 3 * n4 + j;
            const T before = j == 0 ? c2_2 : c1[j - 1], after = j == 2 ? c[0] : c1[j + 1];
            ac[j] = (!EDGE || (p >= 1 && p <= L - 2)) ? k.alt_scale * (after - before) : T(0);       // niir.py:126-129
            su[j] = h1[j] * c1[j];                                                         // niir.py:131-132
            cu[j] = h1[j] * ac[j];
        }
        NiirOut<T> o;
        o.sinphi = dn_s.push(k.taps, su);
        o.cosphi = dn_c.push(k.taps, cu);
        o.sat = k.sat_gain * dn_sat.push(k.taps, s1);
        o.sinc = k.third * dn_sc.push(k.taps, c1);
        o.cosc = k.third * dn_cc.push(k.taps, ac);
        c2_2 = c1[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { c1[j] = c[j]; h1[j] = h[j]; s1[j] = s[j]; }
        return o;
    }
};

// ---- round 4: the hue path in float64 ------------------------------------------------------------------------------------------
// The decoder takes the hue as the ANGLE of the decimated pair (sinphi, cosphi) = resample_poly(huemod_up * {carrier_up, altcarrier_up}, 1, 3)
// (niir.py:131-137).  Where the hue turns quickly inside the decimator's window that pair gets short - every absolute error in it is
// divided by its length - and float32 rounding of the 3x-rate front end (interpolator, band-pass, low-pass: ~1e-7 of full scale) showed as
// isolated samples beyond 1e-5 (round 3: 4e-5 of all decoded samples on random pictures, worst 4e-3).  So everything whose ABSOLUTE error
// reaches that pair runs in float64: NiirFront<double>, the quotient M / S, the products and these two decimators.  What only scales an
// amplitude - the saturation, the re-modulation carriers, niir_finish - stays float32.  (TP = double on the device; the host simulator
// also instantiates TP = float to show what the split buys: tests/test_sim_am.py.)
template <typename TP>
struct NiirHue {
    Dn3<TP> dn_s, dn_c;
    TP c1[3], h1[3], c2_2;                // carrier and hue-modulated signal of the previous triple; carrier[3 n3 - 4]
    CM_HD void reset() {
        dn_s.reset(); dn_c.reset();
#pragma unroll
        for (int j = 0; j < 3; ++j) c1[j] = h1[j] = TP(0);
        c2_2 = TP(0);
    }
    // own / prev: phasemod_up triples n3 of this call and of the previous one.  Returns the decimated pair at sample n5 = n3 - 11 and the
    // carrier / its derivative of triple n4 = n3 - 1 (the inputs of the two re-modulation decimators, niir.py:145-146).
    template <bool EDGE = true, bool V = true>
    CM_HD void step(const Taps3<TP> &taps, TP alt_scale, int width, int n3, const TP own[3], const TP prev[3], bool alt, TP &sinphi, TP &cosphi,
                    TP car[3], TP acar[3]) {
        const int L = 3 * width, n4 = n3 - 1;
        TP c[3], h[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                       // niir.py:117-124
            c[j] = alt ? own[j] : prev[j];
            h[j] = alt ? prev[j] : own[j];
        }
        TP su[3], cu[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 3 * n4 + j;
            const TP before = j == 0 ? c2_2 : c1[j - 1], after = j == 2 ? c[0] : c1[j + 1];
            acar[j] = (!EDGE || (p >= 1 && p <= L - 2)) ? alt_scale * (after - before) : TP(0);       // niir.py:126-129
            car[j] = c1[j];
            su[j] = h1[j] * c1[j];                                                                      // niir.py:131-132
            cu[j] = h1[j] * acar[j];
        }
        sinphi = dn_s.template push<V>(taps, su);
        cosphi = dn_c.template push<V>(taps, cu);
        c2_2 = c1[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { c1[j] = c[j]; h1[j] = h[j]; }
    }
};

// per-line constants of the decoder (computed in float64 by the caller from AmLine)
template <typename T>
struct NiirLineK {
    T sin_shift, cos_shift;      // of +line_shift on ordinary lines, -line_shift on alternate ones (niir.py:120, 124)
    T sin_ps, cos_ps;            // of the re-modulation phase (niir.py:151-157)
    bool alt;
};

CM_HD float am_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);      // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence (about 10 instructions)
#else
    return std::sqrt(x);
#endif
}
CM_HD double am_sqrt(double x) { return std::sqrt(x); }

// One row of a colour matrix: a x + b y + c z - except that a UNIT row (the component protocol: modulate_components /
// demodulate_components hand (luma, db, dr) over as they are) returns its component untouched, the sign of a zero included.  That sign
// is data here: the NIIR encoder gives a pair its 0.1 pedestal at the hue arctan2(db, dr) (niir.py:42-49, 195), the decoder returns
// saturation * sin / cos(hue) with saturation = max(r - 0.1, 0) (niir.py:63-67) - on grey pixels a pair of ZEROS whose signs are those
// of sin / cos(hue) - and arctan2(0, -0) is pi: a comb wrapper that re-modulates the decoder's pair through the encoder
// (comb.py:105-106 around niir.py:82-83) gets its pedestal at +0.1 or -0.1 by that sign.  (-0) + (+0) = +0: a multiply-add chain loses it.
template <typename T>
CM_HD T row3(T a, T b, T c, T x, T y, T z) {
    if (a == T(1) && b == T(0) && c == T(0)) return x;
    if (a == T(0) && b == T(1) && c == T(0)) return y;
    if (a == T(0) && b == T(0) && c == T(1)) return z;
    return fmaf_(a, x, fmaf_(b, y, c * z));
}
CM_HD float copysign_(float m, float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_copysignf(m, s);
#else
    return std::copysign(m, s);
#endif
}
CM_HD double copysign_(double m, double s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_copysign(m, s);
#else
    return std::copysign(m, s);
#endif
}

// niir.py:134-163, 63-67, 52-61 on the decimated streams of one sample: comp = composite[n5]
template <typename T>
CM_HD Rgb<T> niir_finish(const NiirDemodK<T> &k, const NiirLineK<T> &lk, const NiirOut<T> &o, T comp, bool strip) {
    const T inv = am_rsqrt(o.cosphi * o.cosphi + o.sinphi * o.sinphi);
    const T c1 = o.cosphi * inv, s1 = o.sinphi * inv;
    const T s2 = -c1 * lk.sin_shift - s1 * lk.cos_shift;                 // niir.py:139-140
    const T c2 = s1 * lk.sin_shift - c1 * lk.cos_shift;
    T db = o.sat * s2, dr = o.sat * c2;
    const T r2 = db * db + dr * dr;
    const T r = am_sqrt(r2);
    T luma = comp;
    if (strip) {
        const T u = lk.alt ? -r : db, v = lk.alt ? T(0) : dr;             // niir.py:148-156
        const T u2 = u * lk.cos_ps - v * lk.sin_ps, v2 = u * lk.sin_ps + v * lk.cos_ps;
        luma = comp - (u2 * o.sinc + v2 * o.cosc);
    }
    // _remove_offset: the saturation loses its pedestal, the hue stays (atan2(0, 0) = 0 gives (0, 0))
    const T keep = r > T(0) ? (r - T(0.1) > T(0) ? am_div(r - T(0.1), r) : T(0)) : T(0);
    db *= keep;
    dr *= keep;
    Rgb<T> out;
    out.r = row3(k.m[0][0], k.m[0][1], k.m[0][2], luma, db, dr);       // (a unit row - demodulate_components - keeps the sign of a zero)
    out.g = row3(k.m[1][0], k.m[1][1], k.m[1][2], luma, db, dr);
    out.b = row3(k.m[2][0], k.m[2][1], k.m[2][2], luma, db, dr);
    return out;
}

// =============================================================================================
// NIIR encoder (ref niir.py:78-90, 42-49, 69-76; HueCorrectingNiirModem :181-202).  The caller forms (luma, db, dr) of the
// modulated line; add_offset / hue_correct give the two colour-difference signals their pedestal, both pass the
// pre-correction low-pass (FilterFunction at 1x, shift s_c), and
//   composite[n] = luma[n] + db[n] sin(phi + n step) + dr[n] cos(phi + n step)          ordinary lines
//                = luma[n] - sqrt(db^2 + dr^2)[n] sin(phi + n step)                        alternate lines
// =============================================================================================
template <typename T>
struct NiirModK {
    int32_t width, s_c;
    SosK<T, 2> pre;
    T pre_gain;
    T e[3][3];                   // (luma, db, dr) = e . (r, g, b)
    double ed[6];                // the db and dr rows in float64 (niir_chroma_f64)
};

// niir.py:42-49 (noise off): saturation + 0.1 at the same hue
template <typename T>
CM_HD void niir_add_offset(T &db, T &dr) {
    const T r = am_sqrt(db * db + dr * dr);
    if (r > T(0)) {
        const T f = (r + T(0.1)) / r;
        db *= f;
        dr *= f;
    } else {            // arctan2(+-0, +0) = +-0: sin 0, cos 1; arctan2(+-0, -0) = +-pi: cos -1 (row3 above: the sign of a zero is data)
        db = T(0);
        dr = copysign_(T(0.1), dr);
    }
}
// niir.py:42-49 with noise_level != 0: the pedestal comes from the clean saturation, the hue from the noisy pair
// (n_b, n_r = (numpy.random.random_sample - 0.5) * noise_level, drawn by the host in the reference's call order)
template <typename T>
CM_HD void niir_add_offset_noise(T &db, T &dr, T n_b, T n_r) {
    const T sat = am_sqrt(db * db + dr * dr) + T(0.1);
    db += n_b;
    dr += n_r;
    const T r = am_sqrt(db * db + dr * dr);
    if (r > T(0)) {
        db = sat * db / r;
        dr = sat * dr / r;
    } else {
        db = T(0);
        dr = copysign_(sat, dr);
    }
}
// niir.py:187-198: saturation-weighted mean hue of this call (db, dr) and the previous one (pdb, pdr), the previous call's
// saturation + 0.1
template <typename T>
CM_HD void niir_hue_correct(T db, T dr, T pdb, T pdr, T &odb, T &odr, T n_b = T(0), T n_r = T(0)) {
    const T ls = am_sqrt(pdb * pdb + pdr * pdr), s = am_sqrt(db * db + dr * dr);
    T div = ls + s;
    if (div == T(0)) div = T(1);
    T adb = (pdb * ls + db * s) / div, adr = (pdr * ls + dr * s) / div;
    if (n_b != T(0)) adb += n_b;      // niir.py:192-194: noise on the mean (no noise: nothing is added - (-0) + 0 would lose the sign the hue of a zero pair hangs on)
    if (n_r != T(0)) adr += n_r;
    const T ra = am_sqrt(adb * adb + adr * adr), ep = ls + T(0.1);
    if (ra > T(0)) {
        odb = ep * adb / ra;
        odr = ep * adr / ra;
    } else {
        odb = T(0);
        odr = copysign_(ep, adr);
    }
}

// ---- small saturation: the pedestal's hue in float64 ---------------------------------------------------------------------------
// The encoder gives every pixel the saturation r + 0.1 AT THE HUE OF (db, dr) (niir.py:42-49, 187-198).  On grey pixels (db, dr) =
// niir.py:35-36 are rounding residues of ~1e-17 of three products, on nearly grey ones small differences of them - and the pedestal,
// a tenth of full scale, points where THEY point.  In float32 that angle is noise below a saturation of ~1e-3 (6e-8 of the products
// against r), and no arithmetic but the reference's own reproduces the residues: (db, dr) are therefore formed again in float64, term
// by term in the reference's order (c0 r + c1 g - c2 b as written: every product and sum rounded on its own, no contraction), and the
// pedestal with them, wherever the float32 saturation is below 1e-2 (reference-generated vectors: tests/golden/am_mod_niir*_grey.npz).
// BYTES: the pixel came in as bytes (ImageModem.modulate, image.py:43-45): the reference forms byte / 255.0 in float64, and so does this
// (x is the kernel's float32 byte / 255: the byte comes back exactly)
template <bool BYTES>
CM_HD void niir_chroma_f64(const double *ed, float rf, float gf, float bf, double &db, double &dr) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    double r = (double)rf, g = (double)gf, b = (double)bf;
    if (BYTES) {
        r = (double)(int)(rf * 255.f + 0.5f) / 255.0;
        g = (double)(int)(gf * 255.f + 0.5f) / 255.0;
        b = (double)(int)(bf * 255.f + 0.5f) / 255.0;
    }
    const double p0 = ed[0] * r, p1 = ed[1] * g, p2 = ed[2] * b;
    const double q0 = ed[3] * r, q1 = ed[4] * g, q2 = ed[5] * b;
    db = (p0 + p1) + p2;
    dr = (q0 + q1) + q2;
    // the component protocol (unit rows): the pair as it came in, the sign of a zero included (row3 above)
    if (ed[0] == 0.0 && ed[1] == 1.0 && ed[2] == 0.0) db = g;
    if (ed[3] == 0.0 && ed[4] == 0.0 && ed[5] == 1.0) dr = b;
}
// niir.py:42-49 in float64 (sin / cos of arctan2(db, dr) = db / r, dr / r; arctan2(0, 0) = 0)
// (no contraction in these two either: where two nearly grey pixels of opposite hue meet, the mean of niir.py:190-191 cancels down to the
// rounding residue of its own products and sums - measured on the device: 6e-5 with fused multiply-adds, 7e-8 without)
CM_HD void niir_add_offset_f64(double db, double dr, double nb, double nr, bool noisy, double &odb, double &odr) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double sat = sqrt(db * db + dr * dr) + 0.1;
    if (noisy) { db += nb; dr += nr; }
    const double r = sqrt(db * db + dr * dr);
    if (r > 0.0) { odb = sat * db / r; odr = sat * dr / r; }
    else { odb = 0.0; odr = copysign_(sat, dr); }
}
// niir.py:187-198 in float64
CM_HD void niir_hue_correct_f64(double db, double dr, double pdb, double pdr, double nb, double nr, double &odb, double &odr) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double ls = sqrt(pdb * pdb + pdr * pdr), s = sqrt(db * db + dr * dr);
    double div = ls + s;
    if (div == 0.0) div = 1.0;
    double adb = (pdb * ls + db * s) / div, adr = (pdr * ls + dr * s) / div;
    if (nb != 0.0) adb += nb;
    if (nr != 0.0) adr += nr;
    const double ra = sqrt(adb * adb + adr * adr), ep = ls + 0.1;
    if (ra > 0.0) { odb = ep * adb / ra; odr = ep * adr / ra; }
    else { odb = 0.0; odr = copysign_(ep, adr); }
}
#ifndef CM_NIIR_SMALL_SAT2
#define CM_NIIR_SMALL_SAT2 1e-4f
#endif
constexpr float kNiirSmallSat2 = CM_NIIR_SMALL_SAT2;      // (saturation 1e-2)^2
// the plain / noisy encoder's pair of one pixel: (db, dr) in, the pair with its pedestal out; (r, g, b): the pixel itself
template <bool BYTES = false, typename T>
CM_HD void niir_offset_pixel(const double *ed, float r, float g, float b, T &db, T &dr, T nb, T nr, bool noisy) {
    const T qb = noisy ? db + nb : db, qr = noisy ? dr + nr : dr;
    const bool small = qb * qb + qr * qr < T(kNiirSmallSat2);
    if (noisy) niir_add_offset_noise(db, dr, nb, nr);
    else niir_add_offset(db, dr);
    if (small) {
        double d0, d1, o0, o1;
        niir_chroma_f64<BYTES>(ed, r, g, b, d0, d1);
        niir_add_offset_f64(d0, d1, (double)nb, (double)nr, noisy, o0, o1);
        db = T(o0);
        dr = T(o1);
    }
}
// the hue-correcting encoder's: (db, dr) of this call and (pdb, pdr) of the previous one in; (pr, pg, pb): the previous call's pixel
template <bool BYTES = false, typename T>
CM_HD void niir_hue_pixel(const double *ed, float r, float g, float b, float pr, float pg, float pb, T db, T dr, T pdb, T pdr, T nb, T nr, T &odb,
                          T &odr) {
    niir_hue_correct(db, dr, pdb, pdr, odb, odr, nb, nr);
    const T ls = am_sqrt(pdb * pdb + pdr * pdr), s = am_sqrt(db * db + dr * dr);
    T div = ls + s;
    if (div == T(0)) div = T(1);
    const T adb = (pdb * ls + db * s) / div + nb, adr = (pdr * ls + dr * s) / div + nr;
    if (adb * adb + adr * adr < T(kNiirSmallSat2)) {
        double d0, d1, p0, p1, o0, o1;
        niir_chroma_f64<BYTES>(ed, r, g, b, d0, d1);
        niir_chroma_f64<BYTES>(ed, pr, pg, pb, p0, p1);
        niir_hue_correct_f64(d0, d1, p0, p1, (double)nb, (double)nr, o0, o1);
        odb = T(o0);
        odr = T(o1);
    }
}

template <typename T>
struct NiirMod {
    FF1<T, 2> pre_b, pre_r;
    CM_HD void reset() {
        pre_b.reset(); pre_r.reset();
    }
    // i: index of the (offset) colour-difference samples fed now; returns the chroma of sample i - s_c given the carrier
    // {sin, cos}(phi + (i - s_c) step)
    CM_HD T step(const NiirModK<T> &k, int i, T db, T dr, bool alt, T sn, T cs) {
        const T b = k.pre_gain * pre_b.template step<AM_FORM_GEN>(k.pre, k.s_c, k.width, i, db);
        const T r = k.pre_gain * pre_r.template step<AM_FORM_GEN>(k.pre, k.s_c, k.width, i, dr);
        return alt ? -am_sqrt(b * b + r * r) * sn : fmaf_(b, sn, r * cs);                 // niir.py:73-76
    }
};

// =============================================================================================
// Packed float32 forms for the wave-pair decoders (cm_am_kernels.h: niir_demod_pair_kernel, proto_demod_pair_kernel; device
// only).  The 61-tap FIR is symmetric: its 31 distinct taps sit in 16 VGPR pairs, tap(i) = c[i' >> 1][i' & 1] with
// i' = min(i, 60 - i); a decimator that runs on a PAIR of signals (sin / cos products, the two re-modulation carriers, chroma /
// luma) executes exactly the scalar decimator's FMAs, two at a time (v_pk_fma_f32, op_sel picks the tap out of its pair).
// =============================================================================================
#if defined(__HIPCC__)
}  // namespace cm
#include "cm_stages_pk.h"
namespace cm {
// (third-band: the 21 taps that are not zero - i <= 30, i mod 3 != 0, and the centre tap - in 11 pairs)
constexpr int tap3_slot(int I) { return (I <= 30 ? I : 60 - I) == 30 ? 20 : 2 * ((I <= 30 ? I : 60 - I) / 3) + ((I <= 30 ? I : 60 - I) % 3) - 1; }
struct TapsPk3 {
    pf2 c[11];
    template <int J>
    __device__ __forceinline__ void load_j(const Taps3<float> &t) {
        constexpr int i0 = J == 10 ? 30 : 3 * J + 1, i1 = J == 10 ? 30 : 3 * J + 2;      // slots 2 J, 2 J + 1
        c[J] = pf2{take(t.h[i0]), take(t.h[i1])};
        pin_pair(c[J]);
        if constexpr (J + 1 < 11) load_j<J + 1>(t);
    }
    __device__ __forceinline__ void load(const Taps3<float> &t) { load_j<0>(t); }
};
template <int I> __device__ __forceinline__ pf2 tap3_fma(const TapsPk3 &k, pf2 x, pf2 acc) {
    constexpr int i = tap3_slot(I);
    return pk_fma_c<i & 1>(k.c[i >> 1], x, acc);
}
template <int I> __device__ __forceinline__ pf2 tap3_mul(const TapsPk3 &k, pf2 x) {
    constexpr int i = tap3_slot(I);
    return pk_mul_c<i & 1>(k.c[i >> 1], x);
}
template <int I> __device__ __forceinline__ float tap3(const TapsPk3 &k) {
    constexpr int i = tap3_slot(I);
    return (i & 1) ? k.c[i >> 1].y : k.c[i >> 1].x;
}
// Dn3 on a pair of signals
struct Dn3Pk {
    pf2 s[20];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int d = 0; d < 20; ++d) s[d] = pf2{0.f, 0.f};
    }
    // two passes over the pending outputs (z2, then z1 - the scalar decimator's order per output - and the centre tap on z0 in between;
    // no packed instruction reads the result of the one right before it - hipcc pads such pairs with an s_nop)
    template <int D> __device__ __forceinline__ void pass2(const TapsPk3 &k, pf2 z2) {
        s[D] = tap3_fma<3 * D - 2>(k, z2, s[D]);
        if constexpr (D < 19) pass2<D + 1>(k, z2);
    }
    template <int D> __device__ __forceinline__ void pass1(const TapsPk3 &k, pf2 z1) {
        s[D - 1] = tap3_fma<3 * D - 1>(k, z1, s[D]);
        if constexpr (D < 19) pass1<D + 1>(k, z1);
    }
    __device__ __forceinline__ pf2 push(const TapsPk3 &k, pf2 z0, pf2 z1, pf2 z2) {
        const pf2 out = s[0];
        pass2<1>(k, z2);
        const pf2 last = tap3_mul<58>(k, z2);
        s[10] = tap3_fma<30>(k, z0, s[10]);
        pass1<1>(k, z1);
        s[19] = tap3_fma<59>(k, z1, last);
        return out;
    }
};
// Dn3 on one signal with the taps out of the pairs
struct Dn3S {
    float s[20];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int d = 0; d < 20; ++d) s[d] = 0.f;
    }
    template <int D>
    __device__ __forceinline__ void upd(const TapsPk3 &k, float z0, float z1, float z2) {
        float a = fma3<true>(tap3<3 * D - 2>(k), z2, s[D]);
        if constexpr (D == 10) a = fma3<true>(tap3<30>(k), z0, a);
        s[D - 1] = fma3<true>(tap3<3 * D - 1>(k), z1, a);
        if constexpr (D < 19) upd<D + 1>(k, z0, z1, z2);
    }
    __device__ __forceinline__ float push(const TapsPk3 &k, const float z[3]) {
        const float out = s[0];
        upd<1>(k, z[0], z[1], z[2]);
        s[19] = fma3<true>(tap3<59>(k), z[1], tap3<58>(k) * z[2]);
        return out;
    }
};
// ---- stage A of proto_demod_pair_kernel in packed float32 (round 5; the interior bodies) ---------------------------------------------
// d = x[XH] + s
template <int XH>
__device__ __forceinline__ pf2 pk_add_bx(pf2 x, pf2 s) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (XH == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(s));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(s));
#else
    d = x[XH] + s;
#endif
    return d;
}
// d = (k.y, k.x) * x[XH]
template <int XH>
__device__ __forceinline__ pf2 pk_mul_xk_swap(pf2 x, pf2 k) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (XH == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(d) : "v"(x), "v"(k));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(k));
#else
    d = pf2{k.y * x[XH], k.x * x[XH]};
#endif
    return d;
}
// d = (k.x * y.x - x, k.y * y.y + x), x = xs[XH] (XH = -1: the pair xs itself): the second state of a (band-pass | band-stop) section pair
template <int XH>
__device__ __forceinline__ pf2 pk_fma_pm(pf2 k, pf2 y, pf2 xs) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (XH < 0) asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1]" : "=v"(d) : "v"(k), "v"(y), "v"(xs));
    else if (XH == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0] neg_lo:[0,0,1]" : "=v"(d) : "v"(k), "v"(y), "v"(xs));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1]" : "=v"(d) : "v"(k), "v"(y), "v"(xs));
#else
    const float xl = XH < 0 ? xs.x : xs[XH < 0 ? 0 : XH], xh = XH < 0 ? xs.y : xs[XH < 0 ? 0 : XH];
    d = pf2{__builtin_fmaf(k.x, y.x, -xl), __builtin_fmaf(k.y, y.y, xh)};
#endif
    return d;
}
// Up3 with the partial sums in pairs: pair i = (s(3 i + 1), s(3 i + 2)) = slots (2 i, 2 i + 1) of Up3::s, so one step moves every pair
// down by one and the two taps of a pair sit in one pair of TapsPk3 (the upper half of the symmetric filter: the same pairs, halves
// swapped by op_sel).  The same 40 products and sums as Up3::push, in its order per output, two per instruction.
struct Up3Pk {
    pf2 s[19];
    __device__ __forceinline__ void from(const Up3<float> &u) {
#pragma unroll
        for (int i = 0; i < 19; ++i) s[i] = pf2{take(u.s[2 * i]), take(u.s[2 * i + 1])};
    }
    __device__ __forceinline__ void to(Up3<float> &u) const {
#pragma unroll
        for (int i = 0; i < 19; ++i) { u.s[2 * i] = s[i].x; u.s[2 * i + 1] = s[i].y; }
    }
    template <int XH, int I> __device__ __forceinline__ void chain(const TapsPk3 &k, pf2 xv) {
        if constexpr (I <= 9) s[I - 1] = pk_fma_xk<XH, 0, false>(xv, k.c[I], s[I]);
        else s[I - 1] = pk_fma_xk<XH, 1, false>(xv, k.c[19 - I], s[I]);
        if constexpr (I < 18) chain<XH, I + 1>(k, xv);
    }
    // x = xv[XH]; returns (out[1], out[2]) of Up3::push - out[0] = 3 h[30] x[t - 10] is the caller's
    template <int XH> __device__ __forceinline__ pf2 push(const TapsPk3 &k, pf2 xv) {
        const pf2 out = pk_fma_xk<XH, 0, false>(xv, k.c[0], s[0]);
        chain<XH, 1>(k, xv);
        s[18] = pk_mul_xk_swap<XH>(xv, k.c[0]);
        return out;
    }
};
// The band-pass (numerators 1 - z^-2) and the band-stop (1 + b1 z^-1 + z^-2) of the Proto-SECAM decoder run over the same samples:
// section j of both as one pair (.x band-pass, .y band-stop) in the band-stop's four-operation form - iir_sym - with b1 = 0 and the
// sign of x in the second state flipped for the band-pass half, which is iir_bp's arithmetic (t = 0 * x + s2 = s2).
struct BpSymK3 {
    pf2 na1[3], na2[3], b1[3];
    __device__ __forceinline__ void load(const SosK<float, 3> &bp, const SosK<float, 3> &sym) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            na1[j] = pf2{take(bp.na1[j]), take(sym.na1[j])};
            na2[j] = pf2{take(bp.na2[j]), take(sym.na2[j])};
            b1[j] = pf2{0.f, take(sym.b1[j])};
            pin_pair(na1[j]); pin_pair(na2[j]); pin_pair(b1[j]);
        }
    }
};
struct BpSymPk3 {
    pf2 s1[3], s2[3], h1, h2;      // h1, h2: the raw outputs of the previous step (FF3::h1, h2)
    __device__ __forceinline__ void from(const FF3<float, 3> &bp, const FF3<float, 3> &sym) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {      // (take: element by element - merged into vector loads the states would sit in scratch memory)
            s1[j] = pf2{take(bp.st.s1[j]), take(sym.st.s1[j])};
            s2[j] = pf2{take(bp.st.s2[j]), take(sym.st.s2[j])};
        }
        h1 = pf2{take(bp.h1), take(sym.h1)};
        h2 = pf2{take(bp.h2), take(sym.h2)};
    }
    __device__ __forceinline__ void to(FF3<float, 3> &bp, FF3<float, 3> &sym) const {
#pragma unroll
        for (int j = 0; j < 3; ++j) { bp.st.s1[j] = s1[j].x; sym.st.s1[j] = s1[j].y; bp.st.s2[j] = s2[j].x; sym.st.s2[j] = s2[j].y; }
        bp.h1 = h1.x; sym.h1 = h1.y;
        bp.h2 = h2.x; sym.h2 = h2.y;
    }
    // one sample x = xs[XH] through both cascades: (band-pass output, band-stop output)
    template <int XH> __device__ __forceinline__ pf2 step(const BpSymK3 &k, pf2 xs) {
        pf2 y = pk_add_bx<XH>(xs, s1[0]);
        pf2 t = pk_fma_xk<XH, 0, false>(xs, k.b1[0], s2[0]);
        s1[0] = pk_fma(k.na1[0], y, t);
        s2[0] = pk_fma_pm<XH>(k.na2[0], y, xs);
#pragma unroll
        for (int j = 1; j < 3; ++j) {
            const pf2 x = y;
            y = pk_add(x, s1[j]);
            t = pk_fma(k.b1[j], x, s2[j]);
            s1[j] = pk_fma(k.na1[j], y, t);
            s2[j] = pk_fma_pm<-1>(k.na2[j], y, x);
        }
        return y;
    }
};
// ---- Dn3 on ONE signal, two steps per call, packed along the accumulator index (round 5) ----------------------------------------------
// One step of Dn3 is s'[d - 1] = b_d z1 + (a_d z2 + s[d]) for d = 1 .. 20 (a_d = h[3 d - 2], b_d = h[3 d - 1], s[20] = 0; d = 10 takes the
// centre tap on z0 in between), so the accumulators move down by one per step and a pair (s[d], s[d + 1]) is no pair of the next step's
// registers - but after TWO steps everything has moved by two: pair P[j] = (s[2 j], s[2 j + 1]) becomes the new P[j - 1] through four
// v_pk_fma (a, b of step 1 with the tap pairs (2 j + 2, 2 j + 3), then a, b of step 2 with (2 j + 1, 2 j + 2)): per output the products
// and sums of Dn3::push in its order, 40 packed for 80 scalar instructions.  The symmetric filter (a_d = b_(21 - d)) halves the tap pairs
// again: the upper ones are the lower ones of the other kind with their halves swapped (op_sel).
struct Dn3TwoK {
    pf2 a1[5], b1[5], a2[5], b2[5], za, zb;      // a1[j] = (a(2 j + 2), a(2 j + 3)), a2[j] = (a(2 j + 1), a(2 j + 2)); za = (0, a(1)), zb = (0, b(1))
    float h30;
    static constexpr int ia(int d) { return tap3i(3 * d - 2); }
    static constexpr int ib(int d) { return tap3i(3 * d - 1); }
    template <int J> __device__ __forceinline__ void load_j(const Taps3<float> &t) {
        a1[J] = pf2{take(t.h[ia(2 * J + 2)]), take(t.h[ia(2 * J + 3)])};
        b1[J] = pf2{take(t.h[ib(2 * J + 2)]), take(t.h[ib(2 * J + 3)])};
        a2[J] = pf2{take(t.h[ia(2 * J + 1)]), take(t.h[ia(2 * J + 2)])};
        b2[J] = pf2{take(t.h[ib(2 * J + 1)]), take(t.h[ib(2 * J + 2)])};
        pin_pair(a1[J]); pin_pair(b1[J]); pin_pair(a2[J]); pin_pair(b2[J]);
        if constexpr (J + 1 < 5) load_j<J + 1>(t);
    }
    __device__ __forceinline__ void load(const Taps3<float> &t) {
        load_j<0>(t);
        za = pf2{0.f, take(t.h[ia(1)])};
        zb = pf2{0.f, take(t.h[ib(1)])};
        pin_pair(za); pin_pair(zb);
        h30 = take(t.h[30]);
    }
};
struct Dn3Two {
    pf2 p[10];
    __device__ __forceinline__ void from(const Dn3<float> &d) {
#pragma unroll
        for (int i = 0; i < 10; ++i) p[i] = pf2{take(d.s[2 * i]), take(d.s[2 * i + 1])};
    }
    __device__ __forceinline__ void to(Dn3<float> &d) const {
#pragma unroll
        for (int i = 0; i < 10; ++i) { d.s[2 * i] = p[i].x; d.s[2 * i + 1] = p[i].y; }
    }
    // T = kpair * z[XH] + T with the tap pair of (step, kind, j): the stored pair, or for j >= 5 the other kind's mirror pair swapped
    template <int STEP, bool B, int J, int XH>
    __device__ __forceinline__ pf2 tap(const Dn3TwoK &k, pf2 z, pf2 t) const {
        if constexpr (J < 5) {
            const pf2 &c = STEP == 1 ? (B ? k.b1[J] : k.a1[J]) : (B ? k.b2[J] : k.a2[J]);
            return pk_fma_xk<XH, 0, false>(z, c, t);
        } else if constexpr (STEP == 1) {
            if constexpr (J == 9) return pk_fma_xk<XH, 1, false>(z, B ? k.za : k.zb, t);       // (a20, a21 = 0) = swapped (0, b1)
            else return pk_fma_xk<XH, 1, false>(z, B ? k.a1[8 - J] : k.b1[8 - J], t);           // a(d) = b(21 - d)
        } else {
            return pk_fma_xk<XH, 1, false>(z, B ? k.a2[9 - J] : k.b2[9 - J], t);
        }
    }
    template <int J>
    __device__ __forceinline__ void pair(const Dn3TwoK &k, pf2 z12, pf2 w12, float z0, float w0) {
        pf2 t;
        if constexpr (J < 9) t = p[J + 1]; else t = pf2{0.f, 0.f};
        t = tap<1, false, J, 1>(k, z12, t);
        if constexpr (J == 4) t.x = fma3<true>(k.h30, z0, t.x);      // the centre tap: d = 10 of step 1 is this pair's low half ...
        t = tap<1, true, J, 0>(k, z12, t);
        t = tap<2, false, J, 1>(k, w12, t);
        if constexpr (J == 4) t.y = fma3<true>(k.h30, w0, t.y);      // ... and of step 2 its high half
        t = tap<2, true, J, 0>(k, w12, t);
        p[J] = t;
        if constexpr (J < 9) pair<J + 1>(k, z12, w12, z0, w0);
    }
    // z: the triple of the first step as z0 and (z1, z2), w: of the second; returns (Dn3::push(z), Dn3::push(w))
    __device__ __forceinline__ pf2 push2(const Dn3TwoK &k, float z0, pf2 z12, float w0, pf2 w12) {
        pf2 out;
        out.x = p[0].x;
        out.y = fma3<true>(k.zb.y, z12.x, fma3<true>(k.za.y, z12.y, p[0].y));      // s'[0] = b1 z1 + (a1 z2 + s[1])
        pair<0>(k, z12, w12, z0, w0);
        return out;
    }
};
// NiirBack for stage B of the wave pair: the four phase decimators as two packed pairs - (sin, cos) products | (carrier, its
// derivative); the fifth (saturation) runs in stage A, which has the low-passed envelope at hand
struct NiirBackPk {
    Dn3Pk dn_sc, dn_car;
    float c1[3], h1[3], c2_2;
    __device__ __forceinline__ void reset() {
        dn_sc.reset(); dn_car.reset();
#pragma unroll
        for (int j = 0; j < 3; ++j) c1[j] = h1[j] = 0.f;
        c2_2 = 0.f;
    }
    // sat: stage A's k.sat_gain * Dn3(S of the previous triple) of this step
    template <bool EDGE = true>
    __device__ __forceinline__ NiirOut<float> step(const NiirDemodK<float> &k, const TapsPk3 &kp, int n3, const float own[3], const float prev[3],
                                                    float sat, bool alt) {
        const int L = 3 * k.width, n4 = n3 - 1;
        float c[3], h[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                       // niir.py:117-124
            c[j] = alt ? own[j] : prev[j];
            h[j] = alt ? prev[j] : own[j];
        }
        pf2 sc[3], car[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 3 * n4 + j;
            const float before = j == 0 ? c2_2 : c1[j - 1], after = j == 2 ? c[0] : c1[j + 1];
            const float ac = (!EDGE || (p >= 1 && p <= L - 2)) ? k.alt_scale * (after - before) : 0.f;       // niir.py:126-129
            sc[j] = pf2{h1[j] * c1[j], h1[j] * ac};                                                           // niir.py:131-132
            car[j] = pf2{c1[j], ac};
        }
        NiirOut<float> o;
        const pf2 r1 = dn_sc.push(kp, sc[0], sc[1], sc[2]);
        const pf2 r2 = dn_car.push(kp, car[0], car[1], car[2]);
        o.sinphi = r1.x;
        o.cosphi = r1.y;
        o.sat = sat;
        o.sinc = k.third * r2.x;
        o.cosc = k.third * r2.y;
        c2_2 = c1[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { c1[j] = c[j]; h1[j] = h[j]; }
        return o;
    }
};
#endif

// ---- line geometry and sub-carrier start phase, float64 (line.py:57-65, utils.py:82-88) ----------------------------------
struct AmLine {
    int32_t line_shift, even_first, odd_first, frame_cycle;
    double frame_phase_shift, line_phase_shift;
    CM_HD int analog_line(int line) const {
        const int a = line + line_shift;
        // Python's floor division / modulo on possibly negative numbers (line - 2 of the first line of a field)
        const int half = (a >= 0) ? a / 2 : -((-a + 1) / 2);
        return ((a - 2 * half) == 0 ? even_first : odd_first) + half;
    }
    CM_HD bool alternate(long long frame, int line) const {
        const int al = analog_line(line);
        const int pa = ((al % 2) + 2) % 2, pf = (int)(((frame % 2) + 2) % 2);
        return pa == pf;
    }
    CM_HD double start_phase(long long frame, int line) const {
        const double two_pi = 6.283185307179586476925286766559;
        const int ref = even_first < odd_first ? even_first : odd_first;
        const long long fm = ((frame % frame_cycle) + frame_cycle) % frame_cycle;
        const double a = (double)fm * frame_phase_shift;
        const double b = (double)(analog_line(line) - ref) * line_phase_shift;
        // Python's float % (exact remainder, sign of the divisor)
        double fa = fmod(a, two_pi), fb = fmod(b, two_pi);
        if (fa < 0.0) fa += two_pi;
        if (fb < 0.0) fb += two_pi;
        double s = fmod(fa + fb, two_pi);
        if (s < 0.0) s += two_pi;
        return s;
    }
};

}  // namespace cm
#endif

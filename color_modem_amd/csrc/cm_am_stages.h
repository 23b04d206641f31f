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

// resample_poly(x, 3, 1): y[m] = sum_i 3 h[i] xu[m + 30 - i], xu[3 n] = x[n].  push(x[t]) completes y[3 (t - 10) + j], j = 0..2.
// s[e] = partial sum of y[3 t - 30 + e] from the inputs before x[t].  k.h holds 3 h.
template <typename T>
struct Up3 {
    T s[58];
    CM_HD void reset() {
#pragma unroll
        for (int e = 0; e < 58; ++e) s[e] = T(0);
    }
    CM_HD void push(const Taps3<T> &k, T x, T out[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = fmaf_(k.h[j], x, s[j]);
#pragma unroll
        for (int e = 3; e < 58; ++e) s[e - 3] = fmaf_(k.h[e], x, s[e]);
#pragma unroll
        for (int e = 58; e < 61; ++e) s[e - 3] = k.h[e] * x;
    }
};

// resample_poly(z, 1, 3): y[n] = sum_i h[i] z[3 n + 30 - i].  push(z[3 q .. 3 q + 2]) completes y[q - 10].
// s[d] = partial sum of y[q - 10 + d] from the triples before q.
template <typename T>
struct Dn3 {
    T s[20];
    CM_HD void reset() {
#pragma unroll
        for (int d = 0; d < 20; ++d) s[d] = T(0);
    }
    CM_HD T push(const Taps3<T> &k, const T z[3]) {
        const T out = fmaf_(k.h[0], z[0], s[0]);
#pragma unroll
        for (int d = 1; d < 20; ++d)
            s[d - 1] = fmaf_(k.h[3 * d], z[0], fmaf_(k.h[3 * d - 1], z[1], fmaf_(k.h[3 * d - 2], z[2], s[d])));
        s[19] = fmaf_(k.h[60], z[0], fmaf_(k.h[59], z[1], k.h[58] * z[2]));
        return out;
    }
};

enum { AM_FORM_BP = 0, AM_FORM_SYM = 1, AM_FORM_GEN = 2 };

template <int FORM, typename T, int NSEC>
CM_HD T am_iir(IirState<T, NSEC> &st, const SosK<T, NSEC> &k, T x) {
    if (FORM == AM_FORM_BP) return iir_bp<false>(st, k, x);
    if (FORM == AM_FORM_SYM) return iir_sym<false>(st, k, x);
    return iir_gen<false>(st, k, x);
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
    template <int FORM>
    CM_HD void step(const SosK<T, NSEC> &k, const FFGeom &g, int L, int n1, const T in[3], T out[3]) {
        T y[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int m = 3 * n1 + j;
            y[j] = T(0);
            if (m >= 0 && m < L + g.shift) {
                if (m == L - 1) last = in[j];
                y[j] = am_iir<FORM>(st, k, m < L ? in[j] : last);
            }
        }
        T o0, o1, o2;
        if (g.r == 0) { o0 = y[0]; o1 = y[1]; o2 = y[2]; }
        else if (g.r == 1) { o0 = h1; o1 = y[0]; o2 = y[1]; }
        else { o0 = h2; o1 = h1; o2 = y[0]; }
        h2 = y[1];
        h1 = y[2];
        const int p = 3 * (n1 - g.q);
        out[0] = (p >= 0 && p < L) ? o0 : T(0);
        out[1] = (p + 1 >= 0 && p + 1 < L) ? o1 : T(0);
        out[2] = (p + 2 >= 0 && p + 2 < L) ? o2 : T(0);
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
    template <int FORM>
    CM_HD T step(const SosK<T, NSEC> &k, int shift, int L, int i, T in) {
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
    Taps3<T> up, dn;             // 3 h and h
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
    // luma = luma[t - lat_luma], chroma = chroma[t - lat_chroma] (meaningful inside the row)
    CM_HD void step(const ProtoDemodK<T> &k, int t, T x_now, T &luma, T &chroma) {
        const int L = 3 * k.width, n1 = t - kAmHalf;
        T u[3], c1[3], c2[3], y1[3];
        up.push(k.up, x_now, u);
        ext.template step<AM_FORM_BP>(k.ext, k.ge, L, n1, u, c1);
#pragma unroll
        for (int j = 0; j < 3; ++j) c1[j] = c1[j] < T(0) ? -c1[j] : c1[j];     // protosecam.py:98 (the factor pi / 2 is in chroma_gain)
        post.template step<AM_FORM_GEN>(k.post, k.gp, L, n1 - k.ge.q, c1, c2);
        chroma = fmaf_(k.chroma_gain, dn_c.push(k.dn, c2), T(-1));
        rem.template step<AM_FORM_SYM>(k.rem, k.gr, L, n1, u, y1);
        luma = k.luma_gain * dn_y.push(k.dn, y1);
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
    Taps3<T> up, dn;
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
    CM_HD void step(const ProtoModK<T> &k, int i_c, T d, int i_y, T luma, T &luma_out, T &chroma_out) {
        const T c = pre.template step<AM_FORM_GEN>(k.pre, k.s_c, k.width, i_c, d);
        chroma_out = fmaf_(T(0.125) * k.pre_gain, c, T(0.125));
        if (k.luma_filter) {
            T u[3], y1[3];
            up.push(k.up, (i_y >= 0 && i_y < k.width) ? luma : T(0), u);
            rem.template step<AM_FORM_SYM>(k.rem, k.gr, 3 * k.width, i_y - kAmHalf, u, y1);
            luma_out = k.luma_gain * dn.push(k.dn, y1);
        } else {
            luma_out = luma;
        }
    }
};

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

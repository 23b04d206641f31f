// cm_am_plan.h - host side: cm_am_desc (include/color_modem_hip.h) -> the uniform blocks of cm_am_stages.h.
// Shared by the library (T = float) and by the host simulator of tests/sim (T = double / float).
#ifndef CM_AM_PLAN_H
#define CM_AM_PLAN_H

#include <cmath>
#include <string>
#include <vector>

#include "../../include/color_modem_hip.h"
#include "cm_am_stages.h"
#include "cm_plan.h"

namespace cm {

inline FFGeom ff_geom(int shift, int rate) {
    FFGeom g;
    g.shift = shift;
    g.q = (shift + rate - 1) / rate;
    g.r = rate * g.q - shift;
    return g;
}

template <typename T>
void load_taps3(const cm_am_desc &d, Taps3<T> &taps) {
    for (int i = 0; i < kAmTaps; ++i) taps.h[i] = T(3.0 * d.resample_fir3[i]);    // resample_poly scales the interpolator by `up`
}

inline bool am_shifts_ok(const cm_am_desc &d, std::string &err) {
    const cm_iir_desc *f[4] = {&d.precorrect, &d.bandpass_up, &d.bandstop_up, &d.lowpass_up};
    for (const cm_iir_desc *x : f)
        if (x->shift < 0) { err = "negative FilterFunction shift is not used on this path"; return false; }
    return true;
}

// firwin(61, 1 / 3) as the streaming forms use it (cm_am_stages.h: Up3, Dn3): symmetric, and zero at 30 + 3 k, k != 0
inline bool taps3_third_band(const cm_am_desc &d, std::string &err) {
    const double c = std::fabs(d.resample_fir3[30]);
    for (int i = 0; i < kAmTaps; ++i) {
        if (i % 3 == 0 && i != 30 && std::fabs(d.resample_fir3[i]) > 1e-12 * c) { err = "resample_fir3 is not a third-band filter (taps 30 + 3 k must vanish)"; return false; }
        if (std::fabs(d.resample_fir3[i] - d.resample_fir3[kAmTaps - 1 - i]) > 1e-12 * c) { err = "resample_fir3 is not symmetric"; return false; }
    }
    return true;
}

template <typename T>
bool build_proto_demod_k(const cm_am_desc &d, ProtoDemodK<T> &k, std::string &err) {
    if (!am_shifts_ok(d, err) || !taps3_third_band(d, err)) return false;
    k.width = d.width;
    k.ge = ff_geom(d.bandpass_up.shift, 3);
    k.gr = ff_geom(d.bandstop_up.shift, 3);
    k.gp = ff_geom(d.lowpass_up.shift, 3);
    load_taps3(d, k.taps);
    double g_e, g_r, g_p;
    if (!convert_sos<T, 3>(d.bandpass_up, FORM_BP, k.ext, g_e, err, "bandpass_up", true)) return false;
    if (!convert_sos<T, 3>(d.bandstop_up, FORM_SYM, k.rem, g_r, err, "bandstop_up", true)) return false;
    if (!convert_sos<T, 2>(d.lowpass_up, FORM_GEN, k.post, g_p, err, "lowpass_up", true)) return false;
    k.chroma_gain = T(8.0 * 0.5 * M_PI * std::fabs(g_e) * g_p / 3.0);     // protosecam.py:98, 103; the decimator runs on 3 h
    k.luma_gain = T(g_r / 3.0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) k.m[i][j] = T(d.decode_matrix[3 * i + j]);
    return true;
}

template <typename T>
bool build_proto_mod_k(const cm_am_desc &d, ProtoModK<T> &k, std::string &err) {
    if (!am_shifts_ok(d, err) || !taps3_third_band(d, err)) return false;
    k.width = d.width;
    k.luma_filter = d.premod_luma_filter ? 1 : 0;
    k.s_c = d.precorrect.shift;
    k.gr = ff_geom(d.bandstop_up.shift, 3);
    load_taps3(d, k.taps);
    double g_c, g_r;
    if (!convert_sos<T, 2>(d.precorrect, FORM_GEN, k.pre, g_c, err, "precorrect", true)) return false;
    if (!convert_sos<T, 3>(d.bandstop_up, FORM_SYM, k.rem, g_r, err, "bandstop_up", true)) return false;
    k.pre_gain = T(g_c);
    k.luma_gain = T(g_r / 3.0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) k.e[i][j] = T(d.encode_matrix[3 * i + j]);
    return true;
}

template <typename T>
bool build_niir_demod_k(const cm_am_desc &d, NiirDemodK<T> &k, std::string &err) {
    if (!am_shifts_ok(d, err) || !taps3_third_band(d, err)) return false;
    k.width = d.width;
    k.gb = ff_geom(d.bandpass_up.shift, 3);
    k.gl = ff_geom(d.lowpass_up.shift, 3);
    load_taps3(d, k.taps);
    double g_b, g_l;
    if (!convert_sos<T, 3>(d.bandpass_up, FORM_BP, k.bp, g_b, err, "bandpass_up", true)) return false;
    if (!convert_sos<T, 2>(d.lowpass_up, FORM_GEN, k.lp, g_l, err, "lowpass_up", true)) return false;
    const double sat_up = 0.5 * M_PI * std::fabs(g_b) * g_l;          // saturation_up = sat_up * S   (niir.py:113-114)
    k.c_pm = T(g_b / sat_up);
    k.g_b = T(g_b);
    k.sat_gain = T(sat_up / 3.0);
    k.alt_scale = T(0.5 * 3.0 / d.carrier_phase_step);                // niir.py:126-129
    k.third = T(1.0 / 3.0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) k.m[i][j] = T(d.decode_matrix[3 * i + j]);
    return true;
}

// The phase reference of the first line of a run (niir.py:107-110): _demodulate_upsampled_filter(resample_poly(+-sin(phi + n step), 3, 1)).
// The chain is linear - FilterFunction's tail padding too - and +-sin(phi + n step) = +-(sin(phi) cos(n step) + cos(phi) sin(n step)), so the
// reference of ANY line is +-(sin(phi) R_c + cos(phi) R_s) with two sequences of the plan, and so are the two decimations the decoder takes of it
// where it is the carrier (niir.py:145-146): D = Dn3(R), A = Dn3(altcarrier(R)) (taps 3 h, as the kernels' decimators).
// out = [R_c | R_s] 3 W samples each, then [D_c | D_s | A_c | A_s] W each; float64.
inline bool build_niir_syn(const cm_am_desc &d, std::vector<double> &out, std::string &err) {
    NiirDemodK<double> k;
    if (!build_niir_demod_k<double>(d, k, err)) return false;
    const int W = d.width, L = 3 * W;
    out.assign(10 * (size_t)W, 0.0);
    for (int which = 0; which < 2; ++which) {
        auto x = [&](int t) { return (t < 0 || t >= W) ? 0.0 : (which ? std::sin((double)t * d.carrier_phase_step) : std::cos((double)t * d.carrier_phase_step)); };
        double *R = out.data() + (size_t)which * L;
        NiirSyn<double> sy;
        sy.reset();
        for (int t = 0; t < W + kAmHalf + k.gb.q; ++t) {
            double m[3];
            sy.step(k, t, x(t), x(t - kAmHalf), m);
            const int n2 = t - kAmHalf - k.gb.q;
            for (int j = 0; j < 3; ++j) {
                const int q = 3 * n2 + j;
                if (q >= 0 && q < L) R[q] = k.g_b * m[j];
            }
        }
        auto r_at = [&](int q) { return (q >= 0 && q < L) ? R[q] : 0.0; };
        auto a_at = [&](int q) { return (q >= 1 && q <= L - 2) ? k.alt_scale * (r_at(q + 1) - r_at(q - 1)) : 0.0; };     // niir.py:126-129
        double *D = out.data() + 2 * (size_t)L + (size_t)which * W, *A = D + 2 * (size_t)W;
        for (int n = 0; n < W; ++n) {
            double sd = 0.0, sa = 0.0;
            for (int i = 0; i < kAmTaps; ++i) {
                sd += k.taps.h[i] * r_at(3 * n + 30 - i);
                sa += k.taps.h[i] * a_at(3 * n + 30 - i);
            }
            D[n] = sd;
            A[n] = sa;
        }
    }
    return true;
}

template <typename T>
bool build_niir_mod_k(const cm_am_desc &d, NiirModK<T> &k, std::string &err) {
    if (!am_shifts_ok(d, err)) return false;
    k.width = d.width;
    k.s_c = d.precorrect.shift;
    double g_c;
    if (!convert_sos<T, 2>(d.precorrect, FORM_GEN, k.pre, g_c, err, "precorrect", true)) return false;
    k.pre_gain = T(g_c);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) k.e[i][j] = T(d.encode_matrix[3 * i + j]);
    for (int i = 0; i < 6; ++i) k.ed[i] = d.encode_matrix[3 + i];
    return true;
}

// per-line constants of the NIIR decoder (niir.py:117-124, 148-157), float64
template <typename T>
NiirLineK<T> niir_line_k(const cm_am_desc &d, const AmLine &ln, long long frame, int line) {
    NiirLineK<T> lk;
    lk.alt = ln.alternate(frame, line);
    const double shift = lk.alt ? -d.line_phase_shift : d.line_phase_shift;
    const double ps = (lk.alt ? 0.0 : d.line_phase_shift) + M_PI - d.bandpass_phase_shift;
    lk.sin_shift = T(std::sin(shift));
    lk.cos_shift = T(std::cos(shift));
    lk.sin_ps = T(std::sin(ps));
    lk.cos_ps = T(std::cos(ps));
    return lk;
}

inline AmLine am_line(const cm_am_desc &d) {
    AmLine g;
    g.line_shift = d.line_shift;
    g.even_first = d.even_first;
    g.odd_first = d.odd_first;
    g.frame_cycle = d.frame_cycle < 1 ? 1 : d.frame_cycle;
    g.frame_phase_shift = d.frame_phase_shift;
    g.line_phase_shift = d.line_phase_shift;
    return g;
}

}  // namespace cm
#endif

// cm_plan.h - host-side conversion of a cm_plan_desc (float64, reference conventions) into the
// uniform coefficient blocks and tables the streaming stages consume (cm_stages.h conventions:
// sections normalised to b0 = 1, FIR taps 2h, all gains folded into the per-lane constants).
// Shared by the library (T = float) and by tests/sim (T = float / double).
#ifndef CM_PLAN_H
#define CM_PLAN_H

#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "../../include/color_modem_hip.h"
#include "cm_stages.h"

#ifndef CM_SECAM_FIRST_ORDER      /* 0: the SECAM encoder's first-order sections in the general five-operation form (A/B: profiles/r06_secam_mod_bound.txt) */
#define CM_SECAM_FIRST_ORDER 1
#endif

namespace cm {

enum SectionForm { FORM_BP, FORM_SYM, FORM_GEN };

// Normalise scipy sections to b0 = 1, check the numerator form, return the product of the b0's.
// pad = true (run-time shapes): a shorter cascade is completed with identity sections - numerator equal to the denominator
// and a zero state that stays exactly zero: 1 - z^-2 over 1 - z^-2 (BP), 1 + z^-2 over 1 + z^-2 (SYM), 1 over 1 (GEN).
template <typename T, int NSEC>
bool convert_sos(const cm_iir_desc &d, SectionForm form, SosK<T, NSEC> &out, double &gain, std::string &err,
                 const char *name, bool pad = false) {
    if (pad ? d.n_sections > NSEC : d.n_sections != NSEC) {
        err = std::string(name) + ": section count does not match the kernel instance";
        return false;
    }
    gain = 1.0;
    for (int j = 0; j < NSEC; ++j) {
        out.na1[j] = out.b1[j] = T(0);
        out.na2[j] = form == FORM_BP ? T(1) : (form == FORM_SYM ? T(-1) : T(0));   // -a2
        out.b2[j] = form == FORM_BP ? T(-1) : (form == FORM_SYM ? T(1) : T(0));
    }
    for (int j = 0; j < d.n_sections; ++j) {
        const double *s = d.sos[j];
        if (s[0] == 0.0 || std::fabs(s[3] - 1.0) > 1e-12) {
            err = std::string(name) + ": section is not normalised (b0 = 0 or a0 != 1)";
            return false;
        }
        double b1 = s[1] / s[0], b2 = s[2] / s[0];
        gain *= s[0];
        const double tol = 1e-9;
        if (form == FORM_BP && (std::fabs(b1) > tol || std::fabs(b2 + 1.0) > tol)) {
            err = std::string(name) + ": expected numerator 1 - z^-2 (band-pass section)";
            return false;
        }
        if (form == FORM_SYM && std::fabs(b2 - 1.0) > tol) {
            err = std::string(name) + ": expected numerator 1 + b1 z^-1 + z^-2";
            return false;
        }
        out.na1[j] = T(-s[4]);
        out.na2[j] = T(-s[5]);
        out.b1[j] = T(b1);
        out.b2[j] = T(b2);
    }
    return true;
}

// A filter the variant does not have (n_sections == 0) becomes the identity: all-zero sections, gain 1.
template <typename T, int NSEC>
bool convert_sos_optional(const cm_iir_desc &d, SectionForm form, SosK<T, NSEC> &out, double &gain, std::string &err,
                          const char *name) {
    if (d.n_sections != 0) return convert_sos<T, NSEC>(d, form, out, gain, err, name);
    gain = 1.0;
    for (int j = 0; j < NSEC; ++j) out.na1[j] = out.na2[j] = out.b1[j] = out.b2[j] = T(0);
    return true;
}

inline int pair_delay(int shift) { return (shift + 1) / 2; }

struct DemodScales {
    double base;  // true base pair = base * kernel base pair
    double pre;   // gain of the pre-correction low-pass
};

// Build the uniform block of the QAM-family demodulators.  `pald`: front LPF = pald_lp.
template <typename T, class S>
bool build_demod_k(const cm_plan_desc &d, bool pald, bool need_bsf, DemodK<T, S> &k, DemodScales &sc, std::string &err) {
    k.width = d.width;
    for (int i = 0; i < 10; ++i) k.taps.c[i] = T(2.0 * d.resample_fir[2 * i + 1]);
    k.taps.c0 = T(2.0 * d.resample_fir[20]);
    double g_e, g_r = 0.0, g_l, g_p;
    if (!convert_sos<T, S::NE>(d.extract2x, FORM_BP, k.ext, g_e, err, "extract2x", S::RT)) return false;
    if (need_bsf) {
        if (!convert_sos<T, S::NR>(d.remove2x, FORM_SYM, k.rem, g_r, err, "remove2x", S::RT)) return false;
    } else {
        for (int j = 0; j < S::NR; ++j) k.rem.na1[j] = k.rem.na2[j] = k.rem.b1[j] = k.rem.b2[j] = T(0);
    }
    const cm_iir_desc &lp = pald ? d.pald_lp : d.demod_lp;
    if (!convert_sos<T, S::NL>(lp, FORM_SYM, k.lpf, g_l, err, pald ? "pald_lp" : "demod_lp", S::RT)) return false;
    if (!convert_sos<T, S::NP>(d.precorrect, FORM_GEN, k.pre, g_p, err, "precorrect", S::RT)) return false;
    if (S::RT) {
        if (d.precorrect.shift > S::SP) { err = "pre-correction shift beyond the window of the run-time shape"; return false; }
    } else if ((d.extract2x.shift & 1) != (S::ODD_E ? 1 : 0) || (lp.shift & 1) != (S::ODD_L ? 1 : 0) ||
               (need_bsf && (d.remove2x.shift & 1) != (S::ODD_R ? 1 : 0)) || d.precorrect.shift != S::SP) {
        err = "filter shifts do not match the kernel instance";
        return false;
    }
    k.odd_e = d.extract2x.shift & 1;
    k.odd_l = lp.shift & 1;
    k.odd_r = d.remove2x.shift & 1;
    if (d.extract2x.shift < 0 || lp.shift < 0 || d.remove2x.shift < 0 || d.precorrect.shift < 0) {
        err = "negative FilterFunction shift is not used on this path";
        return false;
    }
    k.q_e = pair_delay(d.extract2x.shift);
    k.q_l = pair_delay(lp.shift);
    k.q_r = pair_delay(d.remove2x.shift);
    k.s_p = d.precorrect.shift;
    // every decimator output is doubled (taps are 2h)
    if (pald)
        sc.base = g_e * 0.5 * g_l * 0.5;  // dn2 after the band-pass, dn2 after the low-pass
    else
        sc.base = g_e * 2.0 * g_l * 0.5;  // detector factor 2 (qam.py:51-52), one dn2
    sc.pre = g_p;
    k.luma_gain = T(g_r * 0.5);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) k.m[i][j] = T(d.decode_matrix[3 * i + j]);
    double g_n;
    if (!convert_sos_optional<T, 1>(d.notch, FORM_SYM, k.notch, g_n, err, "notch")) return false;
    if (d.notch.n_sections && d.notch.shift != 0) { err = "notch: only shift 0 is built"; return false; }
    k.notch_gain = d.notch.n_sections ? T(g_n) : T(0);   // 0 switches the notch off in instances that carry it
    return true;
}

// cos/sin of m * cps for m = 0 .. 2W-1, interleaved {C[m], S[m]}
template <typename T>
std::vector<T> build_carrier(double cps, int width) {
    std::vector<T> t(4 * (size_t)width);
    for (int m = 0; m < 2 * width; ++m) {
        t[2 * m] = T(std::cos(m * cps));
        t[2 * m + 1] = T(std::sin(m * cps));
    }
    return t;
}

// Convert one lane-table entry, folding the kernel's scale conventions in.
template <typename T>
LaneK<T> convert_lane(const double *e, const DemodScales &sc) {
    LaneK<T> l;
    l.sph = T(e[2] * sc.pre);
    l.cph = T(e[3] * sc.pre);
    l.vsph = T(e[2] * sc.pre * e[16]);
    l.vcph = T(e[3] * sc.pre * e[16]);
    for (int j = 0; j < 3; ++j) {
        l.cu[j][0] = T(e[4 + 2 * j] * sc.base);
        l.cu[j][1] = T(e[5 + 2 * j] * sc.base);
        l.cv[j][0] = T(e[10 + 2 * j] * sc.base);
        l.cv[j][1] = T(e[11 + 2 * j] * sc.base);
        l.cu2[j][0] = T(e[20 + 2 * j] * sc.base);
        l.cu2[j][1] = T(e[21 + 2 * j] * sc.base);
        l.cv2[j][0] = T(e[26 + 2 * j] * sc.base);
        l.cv2[j][1] = T(e[27 + 2 * j] * sc.base);
    }
    return l;
}

// ---- SECAM ------------------------------------------------------------------------------------
template <typename T>
bool build_secam_demod_k(const cm_plan_desc &d, SecamDemodK<T> &k, std::string &err) {
    const cm_secam_desc &s = d.secam;
    if (!s.present) { err = "SECAM constants missing"; return false; }
    k.width = d.width;
    k.preroll = s.preroll;
    for (int i = 0; i < 10; ++i) k.taps.c[i] = T(2.0 * d.resample_fir[2 * i + 1]);
    k.taps.c0 = T(2.0 * d.resample_fir[20]);
    double g_b, g_bell, g_l, g_y, g_d;
    if (!convert_sos<T, 3>(s.chroma_bp, FORM_BP, k.bpf, g_b, err, "chroma_bp", true)) return false;   // lower orders at low sampling rates: identity-padded
    if (!convert_sos_optional<T, 1>(s.bell, FORM_BP, k.bell, g_bell, err, "bell")) return false;
    k.has_bell = s.bell.n_sections != 0;   // secam.py:167-170: variants with bell_kn == bell_kd have no bell filter
    if (!convert_sos<T, 3>(s.fm_lp, FORM_SYM, k.lpf, g_l, err, "fm_lp", true)) return false;
    if (!convert_sos<T, 3>(s.luma_bs, FORM_SYM, k.ybs, g_y, err, "luma_bs", true)) return false;
    if (!convert_sos_optional<T, 1>(s.lf_rev, FORM_GEN, k.deemph, g_d, err, "lf_rev")) return false;   // secam.py:173-177
    if (s.bell.shift != 0 || s.lf_rev.shift != 0 || s.fm_lp.shift < 0 || s.chroma_bp.shift < 0 || s.luma_bs.shift < 0) {
        err = "SECAM filter shifts outside what the kernel is built for (bell 0, de-emphasis 0)";
        return false;
    }
    if (k.deemph.na2[0] != T(0) || k.deemph.b2[0] != T(0)) { err = "SECAM de-emphasis: a first-order section is what the kernels carry (secam.py:173-177)"; return false; }
    if (g_b * g_bell * g_l <= 0.0) { err = "SECAM chroma path gain must be positive (the discriminator drops it)"; return false; }
    k.s_b = s.chroma_bp.shift;
    k.q_l = pair_delay(s.fm_lp.shift);
    k.odd_l = s.fm_lp.shift & 1;
    k.s_y = s.luma_bs.shift;
    k.two_over_pi = T(2.0 / 3.141592653589793238462643383279502884);
    k.luma_gain = T(g_y);
    for (int i = 0; i < 3; ++i) {
        k.m[i][0] = T(d.decode_matrix[3 * i]);
        k.m[i][1] = T(d.decode_matrix[3 * i + 1] * g_d);
        k.m[i][2] = T(d.decode_matrix[3 * i + 2] * g_d);
    }
    return true;
}

// The float64 band-pass + bell of the row ends (cm_stages.h: SecamBp64): the same sections in float64 and the lengths of
// head and tail from the slowest pole of the two filters (a2 = r^2: tau = -2 / ln(a2) samples).
inline bool build_secam_bp64(const cm_plan_desc &d, SecamBp64 &e, std::string &err) {
    const cm_secam_desc &s = d.secam;
    double g_b, g_bell;
    if (!convert_sos<double, 3>(s.chroma_bp, FORM_BP, e.bpf, g_b, err, "chroma_bp", true)) return false;
    if (!convert_sos_optional<double, 1>(s.bell, FORM_BP, e.bell, g_bell, err, "bell")) return false;
    double a2 = 0.0;
    for (int j = 0; j < s.chroma_bp.n_sections && j < 3; ++j) a2 = std::fmax(a2, -e.bpf.na2[j]);
    if (s.bell.n_sections) a2 = std::fmax(a2, -e.bell.na2[0]);
    if (!(a2 > 0.0 && a2 < 1.0)) { err = "SECAM chroma band-pass: pole radius outside (0, 1)"; return false; }
    const double tau = -2.0 / std::log(a2);
    e.head = (int)std::ceil(4.0 * tau);
    e.tail = (int)std::ceil(3.0 * tau);
    return true;
}

// e = {fsc, fdev, own_is_db, w_prev} of the line (normalised frequencies)
template <typename T>
SecamDemodLaneK<T> convert_secam_demod_lane(const double *e, const cm_secam_desc &s) {
    SecamDemodLaneK<T> l;
    l.scale = T(0.5 / e[1]);
    l.off2 = T(2.0 * (s.fm_fc - e[0]));
    l.lo = T(2.0 * (s.flimit_min - e[0]));
    l.hi = T(2.0 * (s.flimit_max - e[0]));
    l.own_is_db = T(e[2]);
    l.w_prev = T(e[3]);
    return l;
}

// dc[n] = fc (2 sum_k h[k] [0 <= 2 n + 20 - k < 2 lc] - 2): what the decimator makes of the constant fc of
// frequencies_up (length 2 lc, zero-padded by resample_poly) beyond 2 fc, n = 0 .. lc - 1 (float64 arithmetic)
template <typename T>
std::vector<T> build_fm_dc(const cm_plan_desc &d, int lc) {
    std::vector<T> t((size_t)lc);
    for (int n = 0; n < lc; ++n) {
        double acc = 0.0;
        for (int k = 0; k < 41; ++k) {
            const long long j = 2LL * n + 20 - k;
            if (j >= 0 && j < 2LL * lc) acc += d.resample_fir[k];
        }
        t[n] = T(d.secam.fm_fc * (2.0 * acc - 2.0));
    }
    return t;
}

template <typename T, typename TD>
bool build_secam_mod_k(const cm_plan_desc &d, SecamModK<T, TD> &k, std::string &err) {
    const cm_secam_desc &s = d.secam;
    if (!s.present) { err = "SECAM constants missing"; return false; }
    double g1, g2;
    if (!convert_sos<TD, 2>(s.pre_lp, FORM_GEN, k.pre_lp, g1, err, "pre_lp", true)) return false;   // order 2 at low sampling rates (SECAM-A on 405 lines)
    if (!convert_sos_optional<TD, 1>(s.lf_pre, FORM_GEN, k.lf_pre, g2, err, "lf_pre")) return false;
    if (s.lf_pre.shift != 0 || s.pre_lp.shift < 0) { err = "SECAM encoder filter shifts outside what the kernel is built for"; return false; }
    {   // first-order sections (b2 = a2 = 0) run in their own three-operation form (cm_stages.h: SecamMod::step); pre_lp's second-order
        // section goes in front of its first-order one (the cascade's order changes its rounding at the float64 level only)
        auto first_order = [](const SosK<TD, 2> &f, int j) { return f.na2[j] == TD(0) && f.b2[j] == TD(0); };
        if (first_order(k.pre_lp, 0) && !first_order(k.pre_lp, 1)) {
            std::swap(k.pre_lp.na1[0], k.pre_lp.na1[1]); std::swap(k.pre_lp.na2[0], k.pre_lp.na2[1]);
            std::swap(k.pre_lp.b1[0], k.pre_lp.b1[1]); std::swap(k.pre_lp.b2[0], k.pre_lp.b2[1]);
        }
        k.pre_tail_first = (CM_SECAM_FIRST_ORDER && first_order(k.pre_lp, 1)) ? 1 : 0;
        k.lf_first = (CM_SECAM_FIRST_ORDER && k.lf_pre.na2[0] == TD(0) && k.lf_pre.b2[0] == TD(0)) ? 1 : 0;
    }
    k.width = d.width;
    k.s_p = s.pre_lp.shift;
    k.gain = TD(g1 * g2);
    k.f_min = TD(s.flimit_min);
    k.f_max = TD(s.flimit_max);
    k.f0 = TD(s.bell_f0);
    k.pi = TD(3.141592653589793238462643383279502884);
    k.two_pi = TD(2.0 * 3.141592653589793238462643383279502884);
    k.m0 = T(s.m0);
    k.kn = T(s.bell_kn);
    k.kd = T(s.bell_kd);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) k.e[i][j] = T(d.encode_matrix[3 * i + j]);
    return true;
}

template <typename T, typename TD>
SecamModLaneK<T, TD> convert_secam_mod_lane(const double *e) {
    SecamModLaneK<T, TD> l;
    l.fsc = TD(e[0]);
    l.fdev = TD(e[1]);
    l.own_is_db = T(e[2]);
    l.start_phase = T(e[3]);
    l.wy0 = T(e[4]); l.wy1 = T(e[5]); l.wc0 = T(e[6]); l.wc1 = T(e[7]);
    return l;
}

// {cos, sin}(k * pi * fc / 2) for the 2x samples k = 0 .. 2 Lc - 1 (secam.py:137-140, phase not wrapped)
template <typename T>
std::vector<T> build_fm_reference(double fc, int lc) {
    const int n = 2 * lc;
    std::vector<T> t(2 * (size_t)n);
    const double pi = 3.141592653589793238462643383279502884;
    const double stop = (n * pi * fc) / 2.0;
    const double delta = stop / n;
    for (int i = 0; i < n; ++i) {
        t[2 * i] = T(std::cos(i * delta));
        t[2 * i + 1] = T(std::sin(i * delta));
    }
    return t;
}

// ---- the filter-set shapes this build carries ------------------------------------------------
//                 NE NR NL NP  oddE   oddL   oddR  SP
typedef Sys<2, 2, 3, 1, false, false, false, 2> SysPal;   // PAL-BG @ 13.5 MHz: shifts 4 / 4 (6 for PAL-D) / 2 / 2
typedef Sys<3, 3, 3, 1, false, true, false, 2> SysNtsc;   // NTSC-M @ 13.5 MHz: shifts 6 / 5 / 4 / 2 (also NTSC-N, NTSC 3.61, PAL-M, PAL-N)
typedef Sys<3, 2, 3, 1, false, true, false, 2> SysNtscI;  // NTSC-I, NTSC 4.43 on 625 lines: narrower band-stop
typedef Sys<3, 2, 3, 1, false, true, true, 2> SysNtscSq; // NTSC-M with 640 / 704 samples per line (12.27 / 13.2 MHz): shifts 6 / 5 / 3 / 2 - an odd band-stop shift
// PAL-BG with 768 samples per line (14.75 MHz, the square-pixel raster): band-pass of three sections with shift 7, PAL-D low-pass
// shift 7 - and the plain first line of a field runs the detector low-pass with shift 4: the two passes of one launch differ
// in one parity, so this shape is a pair (main pass, first-line pass)
typedef Sys<3, 2, 3, 1, true, true, false, 2> SysPalSq;
typedef Sys<3, 2, 3, 1, true, false, false, 2> SysPalSqFirst;
typedef Sys<4, 3, 3, 2, true, true, true, 4> SysNtscA;    // NTSC-A (405 lines, 2.66 MHz sub-carrier): shifts 9 / 7 / 5 / 4
typedef Sys<4, 3, 3, 2, false, false, false, 12, true> SysAny;   // run-time shape: up to 4 / 3 / 3 / 2 sections, pre shift <= 12

struct SysSignature {
    int ne, nr, nl, np, odd_e, odd_l, odd_r, sp;
};
template <class S>
inline SysSignature signature_of() {
    return SysSignature{S::NE, S::NR, S::NL, S::NP, S::ODD_E, S::ODD_L, S::ODD_R, S::SP};
}
// signature a plan asks for; pald: front low-pass is pald_lp
inline SysSignature signature_wanted(const cm_plan_desc &d, bool pald) {
    const cm_iir_desc &lp = pald ? d.pald_lp : d.demod_lp;
    return SysSignature{d.extract2x.n_sections, d.remove2x.n_sections, lp.n_sections, d.precorrect.n_sections,
                        d.extract2x.shift & 1, lp.shift & 1, d.remove2x.shift & 1, d.precorrect.shift};
}
// can the run-time shape carry this filter set?
inline bool fits_any(const SysSignature &w) {
    return w.ne <= SysAny::NE && w.nr <= SysAny::NR && w.nl <= SysAny::NL && w.np <= SysAny::NP && w.sp <= SysAny::SP;
}
inline bool same_signature(const SysSignature &a, const SysSignature &b) {
    return a.ne == b.ne && a.nr == b.nr && a.nl == b.nl && a.np == b.np && a.odd_e == b.odd_e && a.odd_l == b.odd_l &&
           a.odd_r == b.odd_r && a.sp == b.sp;
}

}  // namespace cm
#endif

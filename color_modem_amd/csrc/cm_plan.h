// cm_plan.h - host-side conversion of a cm_plan_desc (float64, reference conventions) into the
// uniform coefficient blocks and tables the streaming stages consume (cm_stages.h conventions:
// sections normalised to b0 = 1, FIR taps 2h, all gains folded into the per-lane constants).
// Shared by the library (T = float) and by tests/sim (T = float / double).
#ifndef CM_PLAN_H
#define CM_PLAN_H

#include <cmath>
#include <string>
#include <vector>

#include "../../include/color_modem_hip.h"
#include "cm_stages.h"

namespace cm {

enum SectionForm { FORM_BP, FORM_SYM, FORM_GEN };

// Normalise scipy sections to b0 = 1, check the numerator form, return the product of the b0's.
template <typename T, int NSEC>
bool convert_sos(const cm_iir_desc &d, SectionForm form, SosK<T, NSEC> &out, double &gain, std::string &err,
                 const char *name) {
    if (d.n_sections != NSEC) {
        err = std::string(name) + ": section count does not match the kernel instance";
        return false;
    }
    gain = 1.0;
    for (int j = 0; j < NSEC; ++j) out.na1[j] = out.na2[j] = out.b1[j] = out.b2[j] = T(0);
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
    if (!convert_sos<T, S::NE>(d.extract2x, FORM_BP, k.ext, g_e, err, "extract2x")) return false;
    if (need_bsf) {
        if (!convert_sos<T, S::NR>(d.remove2x, FORM_SYM, k.rem, g_r, err, "remove2x")) return false;
    } else {
        for (int j = 0; j < S::NR; ++j) k.rem.na1[j] = k.rem.na2[j] = k.rem.b1[j] = k.rem.b2[j] = T(0);
    }
    const cm_iir_desc &lp = pald ? d.pald_lp : d.demod_lp;
    if (!convert_sos<T, S::NL>(lp, FORM_SYM, k.lpf, g_l, err, pald ? "pald_lp" : "demod_lp")) return false;
    if (!convert_sos<T, S::NP>(d.precorrect, FORM_GEN, k.pre, g_p, err, "precorrect")) return false;
    if ((d.extract2x.shift & 1) != (S::ODD_E ? 1 : 0) || (lp.shift & 1) != (S::ODD_L ? 1 : 0) ||
        (need_bsf && (d.remove2x.shift & 1) != (S::ODD_R ? 1 : 0)) || d.precorrect.shift != S::SP) {
        err = "filter shifts do not match the kernel instance";
        return false;
    }
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
    l.sth = T(e[0]);
    l.cth = T(e[1]);
    l.sph = T(e[2] * sc.pre);
    l.cph = T(e[3] * sc.pre);
    l.vsph = T(e[2] * sc.pre * e[16]);
    l.vcph = T(e[3] * sc.pre * e[16]);
    for (int j = 0; j < 3; ++j) {
        l.cu[j][0] = T(e[4 + 2 * j] * sc.base);
        l.cu[j][1] = T(e[5 + 2 * j] * sc.base);
        l.cv[j][0] = T(e[10 + 2 * j] * sc.base);
        l.cv[j][1] = T(e[11 + 2 * j] * sc.base);
    }
    return l;
}

// ---- the filter-set shapes this build carries ------------------------------------------------
//                 NE NR NL NP  oddE   oddL   oddR  SP
typedef Sys<2, 2, 3, 1, false, false, false, 2> SysPal;   // PAL-BG @ 13.5 MHz: shifts 4 / 4 (6 for PAL-D) / 2 / 2
typedef Sys<3, 3, 3, 1, false, true, false, 2> SysNtsc;   // NTSC-M @ 13.5 MHz: shifts 6 / 5 / 4 / 2

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
inline bool same_signature(const SysSignature &a, const SysSignature &b) {
    return a.ne == b.ne && a.nr == b.nr && a.nl == b.nl && a.np == b.np && a.odd_e == b.odd_e && a.odd_l == b.odd_l &&
           a.odd_r == b.odd_r && a.sp == b.sp;
}

}  // namespace cm
#endif

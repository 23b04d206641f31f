// cm_stages.h - per-scanline streaming stages of the colour demodulators/modulators.
//
// Execution model: ONE LANE OWNS ONE SCANLINE.  A 64-lane wavefront
// walks 64 consecutive "calls" (rows of one field, in the order the reference's ImageModem
// presents them) in lock-step, one pixel per step.  Every filter of the reference is applied
// as a streaming recurrence whose state lives in the lane's VGPRs:
//
//   * resample_poly 2x up / 2x down (41-tap Kaiser half-band FIR, scipy.signal.resample_poly
//     as called at ref qam.py:35-57, pal.py:72-77, secam.py:136-149) -> transposed-form FIR
//     chain, 19 accumulators, one 3-address FMA per tap and NO register moves;
//   * lfilter (ref utils.py:28-36) -> cascade of second-order sections in transposed direct
//     form II, float32 (direct form fails 1e-5, sections reach 2-6e-7);
//   * FilterFunction's delay compensation (append `shift` copies of the last sample, drop the
//     first `shift` outputs) -> index bookkeeping on the stream, no buffers.
//
// All coefficients are wave-uniform (SGPRs, or VGPR copies for the hot blocks: see VPolicy); the only cross-lane
// traffic is the comb filter's "previous line" term, exchanged at 1x rate after the base demodulation
// (linearity: demod(curr +- last) = demod(curr) +- demod(last)).
//
// The header compiles for the device (T = float, hipcc) and for the host (T = float or
// double, g++) - the host instantiation is used ONLY by tests/sim to check the streaming
// schedule against the oracle without a GPU; the product never runs it.
#ifndef CM_STAGES_H
#define CM_STAGES_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include <cmath>
#define CM_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define CM_HD inline
#endif

// Experiment switches (CM_EXP_*: timing ablations whose results are wrong by design; CM_DEV_ROLE: one stage of the pair
// only) exist for tools/dev_build.sh A/B builds.  They compile only together with -DCM_EXPERIMENTS, which
// cm_plan_describe reports and __graft_entry__.build() never sets: a stray -D cannot ship a wrong library silently.
#if !defined(CM_EXPERIMENTS) &&                                                                                              \
    (defined(CM_EXP_NO_LUMA) || defined(CM_EXP_NO_STORE) || defined(CM_EXP_PLAIN_STORE) || defined(CM_EXP_NO_FILL_WAIT) ||   \
     defined(CM_EXP_NO_BSKIP) || defined(CM_EXP_ROLE_SWAP) || defined(CM_EXP_NO_FIR) || defined(CM_EXP_NO_PKFIR) ||          \
     defined(CM_EXP_NO_LPF) || defined(CM_EXP_SECAM_ALWAYS_FAST) || defined(CM_EXP_SECAM_NO_LPF) ||                          \
     defined(CM_EXP_SECAM_NO_PHASE) || defined(CM_EXP_SECAM_MOD_F32) || defined(CM_DEV_ROLE))
#error "CM_EXP_* / CM_DEV_ROLE switches produce wrong results by design: build them with -DCM_EXPERIMENTS (tools/dev_build.sh)"
#endif

namespace cm {

// Compile-time shape of one colour system's filter set: section counts of the extract band-pass,
// remove band-stop, detector low-pass and pre-correction low-pass, the parity of the
// FilterFunction shifts of the three 2x-rate filters (odd: output pairs straddle input pairs)
// and the shift of the pre-correction filter (a register-window delay in the kernels).
// RT_ = true: a run-time shape.  The section counts and SP_ are then MAXIMA (the host pads a shorter cascade with
// identity sections, which cost arithmetic but no branch), the shift parities and the pre-correction shift are read from
// DemodK at run time.  One such instance serves every sampling rate (= image width) without a tuned instance.
// LUMA_ (round 6, the tuned shapes of the wide rasters, cm_shapes_wide.h): everything above is a compile-time constant as in the
// tuned shapes, but the pipeline latency grows with the sampling rate (46 steps at 720 samples per line, 63 at 1920) and with it
// the luma delay ring of the wave pair.  0: the tuned shapes' ring of a fixed size; 1: sized at launch by the plan's latency
// (dynamic LDS), like the run-time shape's; 2: no ring - stage A fetches the luma source samples a second time (one 16-byte
// load per lane and body, L2 hits) and the LDS that frees buys a workgroup per CU.
// YS_: slots of the band-stop luma ring of the wave pair (0: 32, the run-time shape 64).
template <int NE_, int NR_, int NL_, int NP_, bool ODD_E_, bool ODD_L_, bool ODD_R_, int SP_, bool RT_ = false, int LUMA_ = 0, int YS_ = 0>
struct Sys {
    static constexpr int NE = NE_, NR = NR_, NL = NL_, NP = NP_, SP = SP_;
    static constexpr bool ODD_E = ODD_E_, ODD_L = ODD_L_, ODD_R = ODD_R_, RT = RT_;
    static constexpr bool WIDE = LUMA_ != 0;             // a shape of cm_shapes_wide.h
    static constexpr bool DYN = RT_ || LUMA_ == 1;       // luma delay ring sized at launch
    static constexpr bool NORING = LUMA_ == 2;           // no luma delay ring: a second visit of the row
    static constexpr int YS = YS_ ? YS_ : (RT_ ? 64 : 32);
};

// ---- arithmetic helpers -------------------------------------------------------------------
// fma3: d = c * x + acc with d allowed to differ from acc.  On the device this must be the
// VOP3 form: hipcc otherwise picks the 2-address v_fmac and pays one v_mov per tap to rotate
// the accumulators (profiles/r01_ubench_valu.txt, "chain" rows).
// Where a wave-uniform coefficient block lives on the device.  Measured (profiles/r01_notes.md): vector
// instructions that read many different SGPRs issue at ~0.6e12 /s chip-wide at 2 waves per SIMD, the same
// instructions with the coefficient in a VGPR at ~0.9e12 /s.  The hot blocks (FIR taps, the 2x-rate
// low-pass and band-pass sections) are therefore pinned into VGPRs by the kernels; V selects the asm form.
// per front end: {FIR taps, detector low-pass, band-pass} -> 1 = VGPR
#ifndef CM_V_PALD
#define CM_V_PALD 1, 0, 0     /* 245 VGPRs without pinning: only the taps (100 of 202 instructions) fit */
#endif
#ifndef CM_V_QAM
#define CM_V_QAM 1, 1, 1
#endif
#ifndef CM_V_SECAM
#define CM_V_SECAM 1, 1, 1
#endif
#ifndef CM_V_SECAM_A       /* stage A of the SECAM wave pair: band-pass coefficients in SGPRs (165 instead of 172 VGPRs: 3 waves per SIMD) */
#define CM_V_SECAM_A 1, 1, 0
#endif
#ifndef CM_V_SECAM_B       /* stage B of the SECAM wave pair (its decimator's taps) */
#define CM_V_SECAM_B CM_V_SECAM
#endif
template <int VT_, int VL_, int VB_>
struct VPolicy {
    static constexpr bool VT = VT_ != 0, VL = VL_ != 0, VB = VB_ != 0;
};
template <bool V>
CM_HD float fma3(float c, float x, float acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    float d;
    if (V)
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(c), "v"(x), "v"(acc));
    else
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(acc));
    return d;
#else
    return std::fma(c, x, acc);
#endif
}
template <bool V>
CM_HD double fma3(double c, double x, double acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    if (V)
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(c), "v"(x), "v"(acc));
    else
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(acc));
    return d;
#else
    return c * x + acc;
#endif
}
// keep a value in a vector register (no-op on the host)
CM_HD void pin_vgpr(float &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#else
    (void)v;
#endif
}
CM_HD void pin_vgpr(double &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#else
    (void)v;
#endif
}
CM_HD float fmaf_(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}
CM_HD double fmaf_(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}

// ---- uniform coefficient blocks -----------------------------------------------------------
template <typename T>
struct Taps {  // 2*h[2i+1], i = 0..9 (h symmetric: tap 19-i equals tap i) and 2*h[20]
    T c[10];
    T c0;
};

template <typename T, int NSEC>
struct SosK {  // NSEC sections normalised to b0 = 1; gains are folded elsewhere by the host
    T na1[NSEC], na2[NSEC];  // NEGATED denominator coefficients
    T b1[NSEC], b2[NSEC];    // used by the SYM (b1) and GEN (b1, b2) forms
};

template <typename T>
CM_HD void pin_block(Taps<T> &t) {
#pragma unroll
    for (int i = 0; i < 10; ++i) pin_vgpr(t.c[i]);
    pin_vgpr(t.c0);
}
template <typename T, int N>
CM_HD void pin_block(SosK<T, N> &k, bool with_b1) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        pin_vgpr(k.na1[j]);
        pin_vgpr(k.na2[j]);
        if (with_b1) pin_vgpr(k.b1[j]);
    }
}

// ---- transposed-form half-band FIR --------------------------------------------------------
// s[j] carries the partial sum of the output that completes j+1 pushes from now.
template <typename T>
struct HalfbandChain {
    T s[19];
    CM_HD void reset() {
#pragma unroll
        for (int j = 0; j < 19; ++j) s[j] = T(0);
    }
    // 2x interpolation, odd phase: returns out[n] = sum_i c_i * x[n + 10 - i] once x[n + 10] = x
    // has been pushed (ref: resample_poly(x, 2, 1)[2n + 1]).
    template <bool V>
    CM_HD T push(const Taps<T> &k, T x) {
#ifdef CM_EXP_NO_FIR   /* timing experiment: the scalar half-band chains cost nothing (results are wrong) */
        return x + s[0];
#endif
        T out = fma3<V>(k.c[0], x, s[0]);
#pragma unroll
        for (int j = 0; j < 18; ++j) s[j] = fma3<V>(k.c[(j + 1) < 10 ? (j + 1) : 18 - j], x, s[j + 1]);
        s[18] = k.c[0] * x;
        return out;
    }
    // 2x decimation: push the pair (z[2m], z[2m+1]); returns 2 * resample_poly(z, 1, 2)[m - 9].
    template <bool V>
    CM_HD T push_pair(const Taps<T> &k, T even, T odd) {
        T out = push<V>(k, odd);
        s[8] = fmaf_(k.c0, even, s[8]);  // centre tap lands on output m
        return out;
    }
};

// ---- second-order-section cascades (transposed direct form II) ------------------------------
template <typename T, int MAXSEC>
struct IirState {
    T s1[MAXSEC], s2[MAXSEC];
    CM_HD void reset() {
#pragma unroll
        for (int j = 0; j < MAXSEC; ++j) s1[j] = s2[j] = T(0);
    }
};

// numerator 1 - z^-2 (Butterworth / Chebyshev-I band-pass sections)
template <bool V, typename T, int MAXSEC>
CM_HD T iir_bp(IirState<T, MAXSEC> &st, const SosK<T, MAXSEC> &k, T x) {
#pragma unroll
    for (int j = 0; j < MAXSEC; ++j) {
        T y = x + st.s1[j];
        st.s1[j] = fma3<V>(k.na1[j], y, st.s2[j]);
        st.s2[j] = fmaf_(k.na2[j], y, -x);
        x = y;
    }
    return x;
}
// numerator 1 + b1 z^-1 + z^-2 (low-pass / band-stop sections with zeros on the unit circle)
template <bool V, typename T, int MAXSEC>
CM_HD T iir_sym(IirState<T, MAXSEC> &st, const SosK<T, MAXSEC> &k, T x) {
#pragma unroll
    for (int j = 0; j < MAXSEC; ++j) {
        T y = x + st.s1[j];
        T t = fma3<V>(k.b1[j], x, st.s2[j]);
        st.s1[j] = fmaf_(k.na1[j], y, t);
        st.s2[j] = fmaf_(k.na2[j], y, x);
        x = y;
    }
    return x;
}
// general numerator 1 + b1 z^-1 + b2 z^-2 (also first-order sections: b2 = a2 = 0)
template <bool V, typename T, int MAXSEC>
CM_HD T iir_gen(IirState<T, MAXSEC> &st, const SosK<T, MAXSEC> &k, T x) {
#pragma unroll
    for (int j = 0; j < MAXSEC; ++j) {
        T y = x + st.s1[j];
        T t = fma3<V>(k.b1[j], x, st.s2[j]);
        st.s1[j] = fmaf_(k.na1[j], y, t);
        st.s2[j] = fmaf_(k.na2[j], y, k.b2[j] * x);
        x = y;
    }
    return x;
}

// ---- per-lane constants (host-computed, float64 -> T) ---------------------------------------
template <typename T>
struct LaneK {
    T sph, cph;  // sin/cos of the re-modulation phase times the pre-filter gain; both 0 when luma
                 // is passed through unstripped
    T vsph, vcph;  // the same two times the V-switch sign of the re-modulated line (ref pal.py:50-51)
    // R[k] = (Rs, Rc) = the line's phase-free base pair: detector products against sin / cos(m cps),
    // low-passed and decimated.  The detector phase of every contributing line and all filter gains
    // are folded into these coefficients by the host.
    T cu[3][2];  // u = sum_j cu[j][0] * Rs[k-j] + cu[j][1] * Rc[k-j]
    T cv[3][2];  // v likewise
    // second combination, read by MINAVG instances only: (u, v) = minavg(first, second), ref comb.py:13-15
    T cu2[3][2];
    T cv2[3][2];
};

// comb.py:13-15: sign * min(|a|, |b|) with sign = (1 - signbit(a)) - signbit(b): 0 when the signs differ
CM_HD float minavg_(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float m = __builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b));
    const unsigned differ = (__builtin_bit_cast(unsigned, a) ^ __builtin_bit_cast(unsigned, b)) & 0x80000000u;
    return differ ? 0.f : __builtin_copysignf(m, a);
#else
    const float m = std::fmin(std::fabs(a), std::fabs(b));
    return std::signbit(a) != std::signbit(b) ? 0.f : std::copysign(m, a);
#endif
}
CM_HD double minavg_(double a, double b) {
    const double m = std::fmin(std::fabs(a), std::fabs(b));
    return std::signbit(a) != std::signbit(b) ? 0.0 : std::copysign(m, a);
}

template <typename T>
struct Pair {
    T s, c;
};

// Values a FilterFunction stage latches when its input ends (utils.py:31-33 pads with them).  They are
// parameters, not members of the stage state, so that the kernels can keep them out of the registers
// that are live across the edge-free main loop.
template <typename T>
struct FrontLatch {
    T a_last, ps_last, pc_last;
    CM_HD void reset() { a_last = ps_last = pc_last = T(0); }
};
template <typename T>
struct BackLatch {
    T u_last, v_last;
    CM_HD void reset() { u_last = v_last = T(0); }
};

// ---- uniform parameters of the QAM-family demodulators --------------------------------------
template <typename T, class S>
struct DemodK {
    int32_t width;       // W
    int32_t q_e, q_l, q_r, s_p;  // pair delays of the 2x-rate filters (ceil(shift / 2)), pre shift
    int32_t odd_e, odd_l, odd_r; // shift parities of the 2x-rate filters (read by run-time shapes, S::RT)
    Taps<T> taps;
    SosK<T, S::NE> ext;   // ref qam.py:17 band-pass
    SosK<T, S::NR> rem;   // ref qam.py:17 band-stop
    SosK<T, S::NL> lpf;   // ref qam.py:18 (QAM front) or pal.py:67-69 (PAL-D front)
    SosK<T, S::NP> pre;   // ref qam.py:16
    T luma_gain;             // gain of the band-stop path (sections * 1/2 from the decimator)
    T m[3][3];               // (r, g, b) = m * (y, u, v)
    SosK<T, 1> notch;        // ref comb.py:18-20 (numerator b0 (1 + b1 z^-1 + z^-2), shift 0); NOTCH instances only
    T notch_gain;
};

// =============================================================================================
// The front ends are cut in two stages at a 2x-rate sample pair (Mid), so that a kernel can give each
// stage to a wavefront of its own (cm_kernels.h, "wave pair") and hand the pair over through LDS:
//   stage A (per front end)  x -> ... -> the pair the product detectors multiply
//   stage B (Detector)       pair * {sin, cos}(m cps) -> LPF -> dn2 = the line's base pair
// The one-wave composition (PalDFront / QamFront::step) is what tests/sim runs on the host.
// =============================================================================================
template <typename T>
struct Mid {
    T even, odd;
};

// Stage B of both front ends (ref qam.py:47-54 / pal.py:73-77): product detectors against the phase-free
// carriers sin / cos(m cps), m = 2 nd, 2 nd + 1, the detector low-pass with FilterFunction edge handling, dn2.
// nd = index of the incoming pair; the line's detector phase is a rotation of the resulting pair and lives in
// LaneK::cu / cv.  car = {C[2 nd], S[2 nd], C[2 nd + 1], S[2 nd + 1]}.
template <typename T, class S, class VP>
struct Detector {
    typedef DemodK<T, S> K;
    HalfbandChain<T> dn_s, dn_c;
    IirState<T, S::NL> lpf_s, lpf_c;
    T hold_s, hold_c;   // previous odd outputs for an odd shift

    CM_HD void reset() {
        dn_s.reset(); dn_c.reset(); lpf_s.reset(); lpf_c.reset();
        hold_s = hold_c = T(0);
    }
    template <bool EDGE>
    CM_HD Pair<T> step(const K &k, FrontLatch<T> &la, int nd, const Mid<T> &m, const T car[4]) {
        const int W = k.width;
        const int n5 = nd - k.q_l;
        const bool ODD_L = S::RT ? k.odd_l != 0 : S::ODD_L;
        T ps_e = m.even * car[1], pc_e = m.even * car[0];
        T ps_o = m.odd * car[3], pc_o = m.odd * car[2];
        T qs_e = T(0), qs_o = T(0), qc_e = T(0), qc_o = T(0);
        if (!EDGE || (nd >= 0 && nd < W + k.q_l)) {
            if (EDGE) {
                if (nd == W - 1) { la.ps_last = ps_o; la.pc_last = pc_o; }
                if (nd >= W) { ps_e = ps_o = la.ps_last; pc_e = pc_o = la.pc_last; }
            }
            T s0 = iir_sym<VP::VL>(lpf_s, k.lpf, ps_e);
            T s1 = iir_sym<VP::VL>(lpf_s, k.lpf, ps_o);
            T c0 = iir_sym<VP::VL>(lpf_c, k.lpf, pc_e);
            T c1 = iir_sym<VP::VL>(lpf_c, k.lpf, pc_o);
            if (ODD_L) {
                qs_e = hold_s; qs_o = s0; hold_s = s1;
                qc_e = hold_c; qc_o = c0; hold_c = c1;
            } else {
                qs_e = s0; qs_o = s1; qc_e = c0; qc_o = c1;
            }
        }
        if (EDGE && (n5 < 0 || n5 >= W)) qs_e = qs_o = qc_e = qc_o = T(0);
        Pair<T> out;
        out.s = dn_s.template push_pair<VP::VT>(k.taps, qs_e, qs_o);
        out.c = dn_c.template push_pair<VP::VT>(k.taps, qc_e, qc_o);
        return out;
    }
};

// =============================================================================================
// PAL-D front end: x -> E = dn2(BPF(up2 x)) -> up2 -> * {sin, cos}(theta + k cps) -> LPF -> dn2
// (ref pal.py:71-77 applied to ref qam.py:34-37), producing the line's own base pair
// (Ps, Pc)[n].  Gains (band-pass, low-pass, the two 1/2 of the decimators) are NOT applied
// here; the host folds them into LaneK::cu/cv.
//
// Stream indices at step t (one 1x sample per step):
//   n1 = t - 10        pair A(n1) = up2(x)[2 n1, 2 n1 + 1]
//   n2 = n1 - q_e      pair B(n2) = BPF output
//   n3 = n2 - 9        e[n3]
//   n4 = n3 - 10       pair U(n4) = up2(e)                      <- stage A ends here
//   n5 = n4 - q_l      pair Q(n5) = LPF output (two paths)
//   n6 = n5 - 9        (Ps, Pc)[n6]
// =============================================================================================
template <typename T, class S>
struct PalDFrontA {
    typedef DemodK<T, S> K;
    typedef VPolicy<CM_V_PALD> VP;
    HalfbandChain<T> up_x, dn_e, up_e;
    IirState<T, S::NE> bpf;
    T hold_b;

    CM_HD void reset() {
        up_x.reset(); dn_e.reset(); up_e.reset(); bpf.reset();
        hold_b = T(0);
    }
    CM_HD static int pair_offset(const K &k) { return 10 + k.q_e + 9 + 10; }   // nd = t - pair_offset

    // x_now = x[t] (0 beyond the row), x_d10 = x[t - 10], e_d10 = e[n3 - 10] (from the caller's
    // delay window).  Returns e[n3] through e_out (caller stores it in its window) and the pair U(n4).
    template <bool EDGE>
    CM_HD Mid<T> step(const K &k, FrontLatch<T> &la, int t, T x_now, T x_d10, T e_d10, T &e_out) {
        const int W = k.width;
        const int n1 = t - 10, n2 = n1 - k.q_e, n3 = n2 - 9;
        const bool ODD_E = S::RT ? k.odd_e != 0 : S::ODD_E;
        // --- up2(x)
        T a_odd = up_x.template push<VP::VT>(k.taps, x_now);
        T a_even = k.taps.c0 * x_d10;
        // --- band-pass at 2x rate with FilterFunction edge handling
        T b_even = T(0), b_odd = T(0);
        if (!EDGE || (n1 >= 0 && n1 < W + k.q_e)) {
            if (EDGE) {
                if (n1 == W - 1) la.a_last = a_odd;
                if (n1 >= W) a_even = a_odd = la.a_last;
            }
            T y0 = iir_bp<VP::VB>(bpf, k.ext, a_even);
            T y1 = iir_bp<VP::VB>(bpf, k.ext, a_odd);
            if (ODD_E) { b_even = hold_b; b_odd = y0; hold_b = y1; } else { b_even = y0; b_odd = y1; }
        }
        if (EDGE && (n2 < 0 || n2 >= W)) b_even = b_odd = T(0);
        // --- dn2 -> e[n3]
        T e = dn_e.template push_pair<VP::VT>(k.taps, b_even, b_odd);
        if (EDGE && (n3 < 0 || n3 >= W)) e = T(0);
        e_out = e;
        // --- up2(e)
        Mid<T> m;
        m.odd = up_e.template push<VP::VT>(k.taps, e);
        m.even = k.taps.c0 * e_d10;
        return m;
    }
};

template <typename T, class S>
struct PalDFront {
    typedef DemodK<T, S> K;
    typedef VPolicy<CM_V_PALD> VP;
    typedef PalDFrontA<T, S> StageA;
    typedef Detector<T, S, VP> StageB;
    StageA a;
    StageB b;

    CM_HD void reset() { a.reset(); b.reset(); }
    CM_HD static int latency(const K &k) { return 10 + k.q_e + 9 + 10 + k.q_l + 9; }

    // car = {C[2 n4], S[2 n4], C[2 n4 + 1], S[2 n4 + 1]} (cos/sin of m * cps)
    template <bool EDGE>
    CM_HD Pair<T> step(const K &k, FrontLatch<T> &la, int t, T x_now, T x_d10, T e_d10, const T car[4], T &e_out) {
        Mid<T> m = a.template step<EDGE>(k, la, t, x_now, x_d10, e_d10, e_out);
        return b.template step<EDGE>(k, la, t - StageA::pair_offset(k), m, car);
    }
};

// =============================================================================================
// QAM front end (ref qam.py:43-58): x -> up2 -> BPF -> * {2 sin, 2 cos}(theta + k cps) -> LPF
// -> dn2 = base pair (Bs, Bc)[n]; optionally the band-stop luma dn2(BSF(up2 x)).
// The factor 2 of the detector and all filter gains are folded into LaneK by the host.
//   n1 = t - 10 ; n2 = n1 - q_e (stage A ends) ; n5 = n2 - q_l ; n6 = n5 - 9 ; luma: nr = n1 - q_r, nl = nr - 9
// ODD_x: the corresponding FilterFunction shift is odd, i.e. output pairs straddle input pairs.
// =============================================================================================
template <typename T, class S, bool WITH_BSF>
struct QamFrontA {
    typedef DemodK<T, S> K;
    typedef VPolicy<CM_V_QAM> VP;
    HalfbandChain<T> up_x, dn_y;
    IirState<T, S::NE> bpf;
    IirState<T, S::NR> bsf;
    T hold_b, hold_y;  // previous odd outputs for odd shifts

    CM_HD void reset() {
        up_x.reset(); dn_y.reset(); bpf.reset(); bsf.reset();
        hold_b = hold_y = T(0);
    }
    CM_HD static int pair_offset(const K &k) { return 10 + k.q_e; }
    CM_HD static int luma_latency(const K &k) { return 10 + k.q_r + 9; }

    template <bool EDGE>
    CM_HD Mid<T> step(const K &k, FrontLatch<T> &la, int t, T x_now, T x_d10, T &luma_out) {
        const int W = k.width;
        const int n1 = t - 10;
        const bool ODD_E = S::RT ? k.odd_e != 0 : S::ODD_E, ODD_R = S::RT ? k.odd_r != 0 : S::ODD_R;
        T a_odd = up_x.template push<VP::VT>(k.taps, x_now);
        T a_even = k.taps.c0 * x_d10;
        if (EDGE) {
            if (n1 == W - 1) la.a_last = a_odd;
            if (n1 >= W) a_even = a_odd = la.a_last;
        }
        // --- chroma band-pass
        Mid<T> m;
        m.even = m.odd = T(0);
        if (!EDGE || (n1 >= 0 && n1 < W + k.q_e)) {
            T y0 = iir_bp<VP::VB>(bpf, k.ext, a_even);
            T y1 = iir_bp<VP::VB>(bpf, k.ext, a_odd);
            if (ODD_E) { m.even = hold_b; m.odd = y0; hold_b = y1; } else { m.even = y0; m.odd = y1; }
        }
        // (band-pass output outside [0, 2W) is never used: the detector is gated on n2)
        // --- luma band-stop (ref qam.py:57)
        if (WITH_BSF) {
            const int nr = n1 - k.q_r;
            T r_even = T(0), r_odd = T(0);
            if (!EDGE || (n1 >= 0 && n1 < W + k.q_r)) {
                T y0 = iir_sym<false>(bsf, k.rem, a_even);
                T y1 = iir_sym<false>(bsf, k.rem, a_odd);
                if (ODD_R) { r_even = hold_y; r_odd = y0; hold_y = y1; } else { r_even = y0; r_odd = y1; }
            }
            if (EDGE && (nr < 0 || nr >= W)) r_even = r_odd = T(0);
            luma_out = dn_y.template push_pair<VP::VT>(k.taps, r_even, r_odd) * k.luma_gain;
        }
        return m;
    }
};

template <typename T, class S, bool WITH_BSF>
struct QamFront {
    typedef DemodK<T, S> K;
    typedef VPolicy<CM_V_QAM> VP;
    typedef QamFrontA<T, S, WITH_BSF> StageA;
    typedef Detector<T, S, VP> StageB;
    StageA a;
    StageB b;

    CM_HD void reset() { a.reset(); b.reset(); }
    CM_HD static int latency(const K &k) { return 10 + k.q_e + k.q_l + 9; }
    CM_HD static int luma_latency(const K &k) { return StageA::luma_latency(k); }

    // car = {C[2 n2], S[2 n2], C[2 n2 + 1], S[2 n2 + 1]}
    template <bool EDGE>
    CM_HD Pair<T> step(const K &k, FrontLatch<T> &la, int t, T x_now, T x_d10, const T car[4], T &luma_out) {
        Mid<T> m = a.template step<EDGE>(k, la, t, x_now, x_d10, luma_out);
        return b.template step<EDGE>(k, la, t - StageA::pair_offset(k), m, car);
    }
};

// =============================================================================================
// Back end shared by every QAM-family decoder: comb combination of base pairs, chroma
// re-modulation for the luma (ref comb.py:51-53 -> qam.py:20-26), colour matrix.
//   n7 = n6 - s_p ; inputs: own/neighbour base pairs at n6, luma source sample x_l[n7] (or the
//   band-stop luma), carrier {C[2 n7], S[2 n7]}.
// =============================================================================================
template <typename T>
struct Rgb {
    T r, g, b;
};

template <typename T, class S, int DEPTH, bool NOTCH = false, bool MINAVG = false>
struct DemodBack {
    typedef DemodK<T, S> K;
    IirState<T, S::NP> pre_u, pre_v;
    IirState<T, 1> notch;   // touched by NOTCH instances only
    CM_HD void reset() {
        pre_u.reset(); pre_v.reset(); notch.reset();
    }
    // Combination only (the caller keeps the u/v delay windows).
    CM_HD void combine(const LaneK<T> &lk, const Pair<T> &b0, const Pair<T> &b1, const Pair<T> &b2, T &u, T &v) const {
        u = fmaf_(lk.cu[0][0], b0.s, lk.cu[0][1] * b0.c);
        v = fmaf_(lk.cv[0][0], b0.s, lk.cv[0][1] * b0.c);
        if (DEPTH >= 1) {
            u = fmaf_(lk.cu[1][0], b1.s, fmaf_(lk.cu[1][1], b1.c, u));
            v = fmaf_(lk.cv[1][0], b1.s, fmaf_(lk.cv[1][1], b1.c, v));
        }
        if (DEPTH >= 2) {
            u = fmaf_(lk.cu[2][0], b2.s, fmaf_(lk.cu[2][1], b2.c, u));
            v = fmaf_(lk.cv[2][0], b2.s, fmaf_(lk.cv[2][1], b2.c, v));
        }
        if (MINAVG) {
            T u2 = fmaf_(lk.cu2[0][0], b0.s, lk.cu2[0][1] * b0.c);
            T v2 = fmaf_(lk.cv2[0][0], b0.s, lk.cv2[0][1] * b0.c);
            if (DEPTH >= 1) {
                u2 = fmaf_(lk.cu2[1][0], b1.s, fmaf_(lk.cu2[1][1], b1.c, u2));
                v2 = fmaf_(lk.cv2[1][0], b1.s, fmaf_(lk.cv2[1][1], b1.c, v2));
            }
            if (DEPTH >= 2) {
                u2 = fmaf_(lk.cu2[2][0], b2.s, fmaf_(lk.cu2[2][1], b2.c, u2));
                v2 = fmaf_(lk.cv2[2][0], b2.s, fmaf_(lk.cv2[2][1], b2.c, v2));
            }
            u = minavg_(u, u2);
            v = minavg_(v, v2);
        }
    }
    // u, v are the combined chroma at n6; u_d, v_d the same signals at n7 = n6 - s_p;
    // y_src is the luma source at n7; car = {C[2 n7], S[2 n7]}.
    template <bool EDGE>
    CM_HD Rgb<T> step(const K &k, const LaneK<T> &lk, BackLatch<T> &la, int n6, T u, T v, T u_d, T v_d, T y_src, const T car[2]) {
        const int W = k.width;
        T wu = T(0), wv = T(0);
        if (!EDGE || (n6 >= 0 && n6 < W + k.s_p)) {
            if (EDGE) {
                if (n6 == W - 1) { la.u_last = u; la.v_last = v; }
                if (n6 >= W) { u = la.u_last; v = la.v_last; }
            }
            wu = iir_gen<false>(pre_u, k.pre, u);
            wv = iir_gen<false>(pre_v, k.pre, v);
        }
        T sn = fmaf_(lk.sph, car[0], lk.cph * car[1]);
        T cs = fmaf_(lk.vcph, car[0], -(lk.vsph * car[1]));  // +-cos(phi + 2 n7 cps)
        T y = y_src - fmaf_(sn, wu, cs * wv);
        if (NOTCH) {
            // comb.py:54-55, 109-110, pal.py:227-228: the notch follows the chroma strip, i.e. it acts on exactly the
            // lanes that re-modulate (sph, cph != 0); it sees luma[0 .. W) from a zero state (FilterFunction, shift 0)
            const int n7 = n6 - k.s_p;
            if (k.notch_gain != T(0) && (!EDGE || (n7 >= 0 && n7 < W))) {   // gain 0: the plan has no notch
                T yn = iir_sym<false>(notch, k.notch, y) * k.notch_gain;
                if (lk.sph != T(0) || lk.cph != T(0)) y = yn;
            }
        }
        Rgb<T> o;
        o.r = fmaf_(k.m[0][0], y, fmaf_(k.m[0][1], u_d, k.m[0][2] * v_d));
        o.g = fmaf_(k.m[1][0], y, fmaf_(k.m[1][1], u_d, k.m[1][2] * v_d));
        o.b = fmaf_(k.m[2][0], y, fmaf_(k.m[2][1], u_d, k.m[2][2] * v_d));
        return o;
    }
};

// =============================================================================================
// QAM modulator (ref qam.py:20-32 behind pal.py:48-52 / ntsc.py:43-45; encoder-side line averaging
// of comb.py:141-152 folded into the row weights):
//   composite[n] = y[n] + sin(phi + 2 n cps) * F_pre(u)[n] + (+-cos(phi + 2 n cps)) * F_pre(v)[n]
// F_pre is FilterFunction(pre-correction low-pass): the stream index of its output is n7 = n - s_p.
// =============================================================================================
template <typename T, int NP>
struct ModK {
    int32_t width, s_p;
    SosK<T, NP> pre;
    T e[3][3];  // (y, u, v) = e . (r, g, b)
};

template <typename T>
struct ModLaneK {
    T sph, cph;      // sin/cos of the start phase of the modulated line, times the pre-filter gain
    T vsph, vcph;    // the same two times the V-switch sign (pal.py:50-51)
    T wy0, wy1;      // luma  = wy0 * own row + wy1 * previous call's row   (comb.py:147)
    T wc0, wc1;      // chroma likewise                                     (comb.py:148-149)
};

template <typename T, int NP>
struct QamModCore {
    IirState<T, NP> pre_u, pre_v;
    T u_last, v_last;
    CM_HD void reset() {
        pre_u.reset(); pre_v.reset();
        u_last = v_last = T(0);
    }
    // n: index of (u, v); y_d = luma at n - s_p; car = {C[2 (n - s_p)], S[2 (n - s_p)]}
    // EDGE = false: the caller guarantees 0 <= n < W - 1 (no guard can fail, no latch fires): the interior of a row
    template <bool EDGE = true>
    CM_HD T step(const ModK<T, NP> &k, const ModLaneK<T> &lk, int n, T y_d, T u, T v, const T car[2]) {
        const int W = k.width;
        T wu = T(0), wv = T(0);
        if (!EDGE || (n >= 0 && n < W + k.s_p)) {
            if (EDGE) {
                if (n == W - 1) { u_last = u; v_last = v; }
                if (n >= W) { u = u_last; v = v_last; }
            }
            wu = iir_gen<false>(pre_u, k.pre, u);
            wv = iir_gen<false>(pre_v, k.pre, v);
        }
        T sn = fmaf_(lk.sph, car[0], lk.cph * car[1]);
        T cs = fmaf_(lk.vcph, car[0], -(lk.vsph * car[1]));
        return y_d + fmaf_(sn, wu, cs * wv);
    }
};

// =============================================================================================
// SECAM decoder (ref secam.py:278-304 with the FmDecoder of secam.py:127-149).
//
// Chroma stream index m runs over the pre-rolled row cc[m] (m < P: x[P - m], mirrored start;
// m >= P: x[m - P]; secam.py:283-284), Lc = W + P samples:
//   m1 = m - s_b          band-pass + bell output ch[m1]       (shift s_b, tail padded)
//   m2 = m1 - 10          pair A(m2) = up2(ch)
//   m3 = m2 - q_l         pair of low-passed I/Q products -> two phase steps -> frequencies_up
//   m4 = m3 - 9           decimated frequency; row sample n = m4 - P
// The discriminator is scale invariant, so no filter gain of this path is applied.  The phase
// step is taken as atan2(cross, dot) of consecutive I/Q samples, which is the wrapped
// difference numpy.unwrap + numpy.diff produce without the cancellation
// of subtracting two angles.
// =============================================================================================
template <typename T>
struct SecamDemodK {
    int32_t width, preroll;      // W, P
    int32_t s_b, q_l, s_y;       // chroma band-pass shift, low-pass pair delay ceil(shift / 2), luma band-stop shift
    int32_t odd_l;               // 1: the low-pass shift is odd (output pairs straddle the filter's pairs)
    int32_t has_bell;            // 0: the variant has no bell filter (secam.py:167-170)
    Taps<T> taps;
    SosK<T, 3> bpf;              // secam.py:183-184 (numerator 1 - z^-2 sections)
    SosK<T, 1> bell;             // secam.py:168-170
    SosK<T, 3> lpf;              // secam.py:131-132
    SosK<T, 3> ybs;              // secam.py:185-186
    SosK<T, 1> deemph;           // secam.py:175-177 (backward), first order
    T two_over_pi;               // frequencies_up - fc = 2 d / pi (secam.py:148)
    T luma_gain;
    T m[3][3];                   // (r, g, b) = m . (luma, dr, db), de-emphasis gain folded into columns 1, 2
};

// The decimated frequency is carried as its deviation from the discriminator centre: the float32 decimator runs on
// g = frequencies_up - fc (|g| < 0.1, so its rounding is an order of magnitude below that of values around fc = 0.64), the
// response to the constant fc - including the zero-padded row ends - comes from a float64 table of the plan:
//   2 resample_poly(frequencies_up, 1, 2)[n] = dn(g)[n] + 2 fc + dc[n],   dc[n] = fc (2 sum_k h[k] [inside] - 2)
template <typename T>
struct SecamDemodLaneK {
    T scale;          // 0.5 / fdev_x:  c = clip(dn(g) + dc + off2, lo, hi) * scale   (secam.py:290-296)
    T off2;           // 2 (fc - fsc_x)
    T lo, hi;         // 2 (flimit - fsc_x)
    T own_is_db;      // 1: this line carries Db (alternate line), 0: Dr
    T w_prev;         // 0 on the first call of a run (last_chroma = zeros), else 1
};

// Float32 is too thin for the band-pass + bell cascade near the row ends: the filter starts from zero state on a signal
// with a luma step (its states are large while the band-passed sub-carrier is still small) and the sub-carrier collapses
// again where the row ends; the rounding of the states then shows in the angle of (I, Q) - 1.04e-5 of full scale in a
// start-of-row sample of SECAM III at 640 (tests/golden/secam_iii_640_margin.npz), 9.7e-6 in the
// last column of SECAM II at 720.  So the float32 decoders carry these two filters (4 sections) in float64 over the first
// `head` samples of the chroma stream (m < head) and over its last `tail` (m >= W + P - tail) and hand the states to the
// float32 sections in between: the bodies there are the guarded ones anyway (cm_secam_kernels.h), which are extended to
// cover head and tail.  head = 4 tau, tail = 3 tau of the slowest pole of the two filters (host: cm_plan.h).
struct SecamBp64 {
    SosK<double, 3> bpf;
    SosK<double, 1> bell;
    int32_t head, tail;
};

// Main-loop samples [xb0, xb1) (multiples of 4) whose bodies need no guard and run the band-pass in float32: every stage
// index of the four steps m = P + xb .. + 3 lies strictly inside its stream - the first output sample n = m - 1 - lat is
// >= 0 (so is every earlier stage's index), the last input index m + 3 stays below the end-of-row latch at W + P - 1 (so
// does every later stage's, which run behind it) - and the float64 head / tail of SecamBp64 lie outside.
CM_HD void secam_mid_bounds(int W, int P, int lat, const SecamBp64 &e, int &xb0, int &xb1) {
    xb0 = (lat + 1 - P + 3) & ~3;
    xb1 = (W - 8) & ~3;
    const int h = (e.head - P + 3) & ~3, t = (W - e.tail) & ~3;
    if (h > xb0) xb0 = h;
    if (t < xb1) xb1 = t;
    if (xb0 < 0) xb0 = 0;
    if (xb1 <= xb0) xb0 = xb1 = 0;
}

template <typename A, typename B, int N>
CM_HD void convert_state(IirState<A, N> &d, const IirState<B, N> &s) {
#pragma unroll
    for (int j = 0; j < N; ++j) { d.s1[j] = (A)s.s1[j]; d.s2[j] = (A)s.s2[j]; }
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
#else
inline float atan2_(float y, float x) { return std::atan2(y, x); }
#endif
CM_HD double atan2_(double y, double x) { return std::atan2(y, x); }

template <typename T>
struct SecamDemod {
    typedef VPolicy<CM_V_SECAM> VP;
    IirState<T, 3> bpf, ybs;
    IirState<T, 1> bell;
    IirState<double, 3> bpf64;   // the band-pass + bell states while they run in float64 (SecamBp64)
    IirState<double, 1> bell64;
    int in64;
    IirState<T, 3> lp_i, lp_q;
    IirState<T, 1> deemph;
    HalfbandChain<T> up, dn;
    T cc_last, pi_last, pq_last, x_last;
    T i_prev, q_prev;
    T i_hold, q_hold;   // previous odd low-pass output (odd shifts)
    int have_prev;

    CM_HD void reset() {
        bpf.reset(); ybs.reset(); bell.reset(); lp_i.reset(); lp_q.reset(); deemph.reset();
        bpf64.reset(); bell64.reset();
        in64 = 0;
        up.reset(); dn.reset();
        cc_last = pi_last = pq_last = x_last = i_prev = q_prev = i_hold = q_hold = T(0);
        have_prev = 0;
    }
    // row sample n = m - latency(k)
    CM_HD static int latency(const SecamDemodK<T> &k) { return k.s_b + 10 + k.q_l + 9 + k.preroll; }

    CM_HD static T phase_step(T i0, T q0, T i1, T q1) {
        // angle of (i1 + j q1) * conj(i0 + j q0); the cross product with an error-free correction
        T t = q0 * i1;
        T e = fmaf_(-q0, i1, t);
        T cross = fmaf_(i0, q1, -t) + e;
        T dot = fmaf_(i0, i1, q0 * q1);
        return atan2_(cross, dot);
    }

    // cc_now = cc[m]; ch_d10 = ch[m1 - 10] (caller's delay window); car = {cos, sin} of the FM
    // reference at 2x samples 2 m2 and 2 m2 + 1.  Returns the de-emphasised colour-difference sample
    // c[n], n = m - latency (meaningful for 0 <= n < W), and ch[m1] through ch_out.
    // dc = the plan's table entry of row-stream sample m4 (see SecamDemodLaneK)
    // e64 != nullptr: this step runs the band-pass + bell in float64 (the guarded bodies of the float32 kernels, SecamBp64)
    CM_HD T chroma_step(const SecamDemodK<T> &k, const SecamDemodLaneK<T> &lk, int m, T cc_now, T ch_d10, const T car[4], T dc, T &ch_out,
                        const SecamBp64 *e64 = nullptr) {
        const int W = k.width, Lc = k.width + k.preroll;
        const int m1 = m - k.s_b, m2 = m1 - 10, m3 = m2 - k.q_l, m4 = m3 - 9, n = m4 - k.preroll;
        T ch = T(0);
        if ((e64 != nullptr) != (in64 != 0)) {   // hand the states over (float32 -> float64 is exact)
            if (e64) { convert_state(bpf64, bpf); convert_state(bell64, bell); }
            else { convert_state(bpf, bpf64); convert_state(bell, bell64); }
            in64 = e64 != nullptr;
        }
        if (m >= 0 && m < Lc + k.s_b) {
            if (m == Lc - 1) cc_last = cc_now;
            if (m >= Lc) cc_now = cc_last;
            if (e64) {
                double b = iir_bp<false>(bpf64, e64->bpf, (double)cc_now);
                if (m1 >= 0) ch = T(k.has_bell ? iir_bp<false>(bell64, e64->bell, b) : b);
            } else {
                T b = iir_bp<VP::VB>(bpf, k.bpf, cc_now);
                if (m1 >= 0) ch = k.has_bell ? iir_bp<false>(bell, k.bell, b) : b;   // the bell sees the band-pass output from its sample 0 on
            }
        }
        if (m1 < 0 || m1 >= Lc) ch = T(0);
        ch_out = ch;
        T a_odd = up.template push<VP::VT>(k.taps, ch);
        T a_even = k.taps.c0 * ch_d10;
        T pi_e = a_even * car[0], pq_e = -(a_even * car[1]);   // data_up = cos part - j sin part (secam.py:143)
        T pi_o = a_odd * car[2], pq_o = -(a_odd * car[3]);
        T f_e = T(0), f_o = T(0);
        if (m2 >= 0 && m2 < Lc + k.q_l) {
            if (m2 == Lc - 1) { pi_last = pi_o; pq_last = pq_o; }
            if (m2 >= Lc) { pi_e = pi_o = pi_last; pq_e = pq_o = pq_last; }
            T i0 = iir_sym<VP::VL>(lp_i, k.lpf, pi_e), q0 = iir_sym<VP::VL>(lp_q, k.lpf, pq_e);
            T i1 = iir_sym<VP::VL>(lp_i, k.lpf, pi_o), q1 = iir_sym<VP::VL>(lp_q, k.lpf, pq_o);
            if (k.odd_l) {   // pair m3 of the shifted stream = (odd output of the previous pair, even output of this one)
                const T ih = i_hold, qh = q_hold;
                i_hold = i1; q_hold = q1;
                i1 = i0; q1 = q0;
                i0 = ih; q0 = qh;
            }
            if (m3 >= 0 && m3 < Lc) {
                T d_e = have_prev ? phase_step(i_prev, q_prev, i0, q0) : T(0);  // secam.py:147: first step is 0
                T d_o = phase_step(i0, q0, i1, q1);
                have_prev = 1;
                i_prev = i1;
                q_prev = q1;
                f_e = d_e * k.two_over_pi;
                f_o = d_o * k.two_over_pi;
            }
        }
        T g2 = dn.template push_pair<VP::VT>(k.taps, f_e, f_o);   // 2 * resample_poly(frequencies_up - fc [inside], 1, 2)[m4]
        T c = T(0);
        if (n >= 0 && n < W) {
            T f2 = (g2 + dc) + lk.off2;                              // 2 (f - fsc)
            f2 = f2 < lk.lo ? lk.lo : (f2 > lk.hi ? lk.hi : f2);     // secam.py:290
            c = iir_gen<false>(deemph, k.deemph, f2 * lk.scale);     // secam.py:291-296
        }
        return c;
    }
    // luma[n] (secam.py:282): x_in = x[n + s_y] (anything beyond the row end: the filter is fed the last sample)
    CM_HD T luma_step(const SecamDemodK<T> &k, int n, T x_in) {
        const int W = k.width, j = n + k.s_y;
        T y = T(0);
        if (j >= 0 && j < W + k.s_y) {
            if (j == W - 1) x_last = x_in;
            if (j >= W) x_in = x_last;
            y = iir_sym<false>(ybs, k.ybs, x_in);
        }
        return y * k.luma_gain;
    }
    // own = this call's c[n], prev = the previous call's c[n]
    CM_HD Rgb<T> finish(const SecamDemodK<T> &k, const SecamDemodLaneK<T> &lk, T luma, T own, T prev) const {
        prev = prev * lk.w_prev;
        T dr = lk.own_is_db != T(0) ? prev : own;   // secam.py:297-300
        T db = lk.own_is_db != T(0) ? own : prev;
        Rgb<T> o;
        o.r = fmaf_(k.m[0][0], luma, fmaf_(k.m[0][1], dr, k.m[0][2] * db));
        o.g = fmaf_(k.m[1][0], luma, fmaf_(k.m[1][1], dr, k.m[1][2] * db));
        o.b = fmaf_(k.m[2][0], luma, fmaf_(k.m[2][1], dr, k.m[2][2] * db));
        return o;
    }
};

// =============================================================================================
// SECAM modulator (ref secam.py:240-276; encoder-side line averaging of comb.py:141-152 is applied
// by the caller on the components).  The colour-difference path - pre-correction low-pass, LF
// pre-emphasis, frequency and the running phase sum (numpy.cumsum, secam.py:245) - is carried in
// TD = double on the device: the phase is an integral over the whole line and float32 rounding of
// it shows at 1e-4.
//   n  : index of the component samples fed in;  n7 = n - s_p : index of the composite sample out
// =============================================================================================
template <typename T, typename TD>
struct SecamModK {
    int32_t width, s_p;
    SosK<TD, 2> pre_lp;     // secam.py:171-172 (order 3: a first- and a second-order section)
    SosK<TD, 1> lf_pre;     // secam.py:175-177 forward
    TD gain;                // product of the section gains of both filters
    int32_t pre_tail_first, lf_first;   // 1: pre_lp's second section / lf_pre is a FIRST-order section (b2 = a2 = 0; the host puts pre_lp's
                                        // second-order section in front): three float64 operations instead of five, the same results bit for bit
    TD f_min, f_max, f0, pi, two_pi;
    T m0, kn, kd;
    T e[3][3];              // (luma, dr, db) = e . (r, g, b)
};

template <typename T, typename TD>
struct SecamModLaneK {
    TD fsc, fdev;           // of the colour-difference signal this line carries
    T own_is_db;
    T start_phase;          // 0 or pi (secam.py:273)
    T wy0, wy1, wc0, wc1;   // row weights (comb.py:147-149)
};

#if defined(__HIP_DEVICE_COMPILE__)
// sin / cos of the SECAM encoder's running phase (round 5).  The phase is kept wrapped to [0, 2 pi) in float64, so the library sincosf's
// argument reduction for arbitrary floats - ~120 integer instructions per pixel, more than half of the encoder's instruction stream - is dead
// weight: one rounding to the nearest quarter turn, a two-term Cody-Waite subtraction, the two cephes minimax polynomials of |y| <= pi / 4
// (1 ulp each) and the quadrant's swap / signs.  Valid for |x| <= 8; within 1 - 2 ulp of sincosf there.
__device__ __forceinline__ void sincos_(float x, float &s, float &c) {
    const float k = __builtin_rintf(x * 0.636619772367581343f);               // quarter turns
    float y = __builtin_fmaf(-k, 1.5707963705062866f, x);                     // pi / 2 = hi + lo
    y = __builtin_fmaf(-k, -4.371139000186243e-8f, y);
    const float z = y * y;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    const float sy = __builtin_fmaf(y * z, ps, y);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    const float cy = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    const unsigned q = (unsigned)(int)k;
    const bool swap = (q & 1u) != 0u;
    const float s0 = swap ? cy : sy, c0 = swap ? sy : cy;
    s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) ^ ((q & 2u) << 30));           // quadrants 2, 3: sin < 0
    c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c0) ^ (((q + 1u) & 2u) << 30));     // quadrants 1, 2: cos < 0
}
#else
CM_HD void sincos_(float x, float &s, float &c) { s = std::sin(x); c = std::cos(x); }      // (host pass / the stage simulator)
#endif
CM_HD void sincos_(double x, double &s, double &c) { s = std::sin(x); c = std::cos(x); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double clamp_(double x, double lo, double hi) { return __builtin_fmin(__builtin_fmax(x, lo), hi); }
__device__ __forceinline__ float clamp_(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
#else
template <typename T> CM_HD T clamp_(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
#endif
// bell pre-emphasis G = m0 (1 + j kn F) / (1 + j kd F), F = f / f0 - f0 / f (secam.py:241-243) as (re, im); df = f - f0 (formed in the
// wide type by the caller), ff = f.  Round 5: the two quotients through reciprocals - on the device v_rcp_f32 (1 ulp) instead of three
// IEEE divisions (~10 instructions each); one definition for the streaming and the scan encoder, so that they stay bit-identical.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float recip_(float x) { return __builtin_amdgcn_rcpf(x); }
#else
CM_HD float recip_(float x) { return 1.0f / x; }
#endif
CM_HD double recip_(double x) { return 1.0 / x; }
template <typename T>
CM_HD void secam_bell_gain(T df, T ff, T f0, T m0, T kn, T kd, T &re, T &im) {
    const T F = df * (ff + f0) * recip_(ff * f0);
    const T F2 = F * F;
    const T rden = recip_(T(1) + kd * kd * F2);
    re = m0 * (T(1) + kn * kd * F2) * rden;
    im = m0 * F * (kn - kd) * rden;
}

template <typename T, typename TD>
struct SecamMod {
    IirState<TD, 2> pre_lp;
    IirState<TD, 1> lf_pre;
    TD d_last, acc;
    CM_HD void reset() {
        pre_lp.reset(); lf_pre.reset();
        d_last = acc = TD(0);
    }
    // d = colour-difference sample n of this line; luma_d = luma sample n7 = n - s_p.
    // EDGE = false: the caller guarantees s_p < n < W - 1 (inside the row, behind the phase start, before the end latch)
    template <bool EDGE = true>
    CM_HD T step(const SecamModK<T, TD> &k, const SecamModLaneK<T, TD> &lk, int n, T luma_d, T d) {
        const int W = k.width, n7 = n - k.s_p;
        TD dd = TD(d);
        TD w = TD(0);
        if (!EDGE || (n >= 0 && n < W + k.s_p)) {
            if (EDGE) {
                if (n == W - 1) d_last = dd;
                if (n >= W) dd = d_last;
            }
            // round 6: the order-3 pre-correction low-pass is a second-order and a FIRST-order section, the LF pre-emphasis a first-order
            // one (secam.py:171-177, 211-221); the general section form spent two of its five float64 operations per sample on their
            // b2 = a2 = 0 (12 of the encoder's ~20 float64 operations per pixel are these three sections' - profiles/r06_secam_mod_bound.txt)
#if defined(CM_EXPERIMENTS) && defined(CM_EXP_SECAM_MOD_F32)   /* VERDICT r05 item 6's float32 sections: every operation of the two filters rounded to float32
            (states kept in float64 variables, so this measures the ERROR of that arithmetic, not its speed) - profiles/r06_secam_mod_bound.txt */
            if (k.pre_tail_first) {
                const float xd = (float)dd, b10 = (float)k.pre_lp.b1[0], b20 = (float)k.pre_lp.b2[0], a10 = (float)k.pre_lp.na1[0], a20 = (float)k.pre_lp.na2[0];
                const float y0 = xd + (float)pre_lp.s1[0];
                const float t0 = __builtin_fmaf(b10, xd, (float)pre_lp.s2[0]);
                pre_lp.s1[0] = (TD)__builtin_fmaf(a10, y0, t0);
                pre_lp.s2[0] = (TD)__builtin_fmaf(a20, y0, b20 * xd);
                const float w1 = y0 + (float)pre_lp.s1[1];
                pre_lp.s1[1] = (TD)__builtin_fmaf((float)k.pre_lp.na1[1], w1, (float)k.pre_lp.b1[1] * y0);
                w = (TD)w1;
            } else
#endif
            if (k.pre_tail_first) {
                TD y0 = dd + pre_lp.s1[0];
                const TD t0 = fma3<false>(k.pre_lp.b1[0], dd, pre_lp.s2[0]);
                pre_lp.s1[0] = fmaf_(k.pre_lp.na1[0], y0, t0);
                pre_lp.s2[0] = fmaf_(k.pre_lp.na2[0], y0, k.pre_lp.b2[0] * dd);
                w = y0 + pre_lp.s1[1];
                pre_lp.s1[1] = fmaf_(k.pre_lp.na1[1], w, k.pre_lp.b1[1] * y0);      // (= the general form with s2 = 0: b1 y0 + 0)
            } else {
                w = iir_gen<false>(pre_lp, k.pre_lp, dd);
            }
        }
        if (EDGE && (n7 < 0 || n7 >= W)) return T(0);
        TD x;
#if defined(CM_EXPERIMENTS) && defined(CM_EXP_SECAM_MOD_F32)
        if (k.lf_first) {
            const float xf = (float)w + (float)lf_pre.s1[0];
            lf_pre.s1[0] = (TD)__builtin_fmaf((float)k.lf_pre.na1[0], xf, (float)k.lf_pre.b1[0] * (float)w);
            x = (TD)xf;
        } else
#endif
        if (k.lf_first) {
            x = w + lf_pre.s1[0];
            lf_pre.s1[0] = fmaf_(k.lf_pre.na1[0], x, k.lf_pre.b1[0] * w);
        } else {
            x = iir_gen<false>(lf_pre, k.lf_pre, w);
        }
        TD f = fmaf_(lk.fdev * k.gain, x, lk.fsc);                    // secam.py:266 / 271
        f = clamp_(f, k.f_min, k.f_max);                              // secam.py:272 (v_max_f64 + v_min_f64: two instructions, not six)
        // bell pre-emphasis G = m0 (1 + j kn F) / (1 + j kd F), F = f / f0 - f0 / f (secam.py:241-243)
        T re, im;
        secam_bell_gain<T>(T(f - k.f0), T(f), T(k.f0), k.m0, k.kn, k.kd, re, im);
        if (EDGE && n7 == 0) {
            acc = TD(lk.start_phase) - TD(atan2_(im, re));             // secam.py:244: start - pi f[0] - arg G[0] + pi f[0]
        } else {
            acc += k.pi * f;                                          // cumsum(pi f)
        }
        if (acc >= k.two_pi) acc -= k.two_pi;
        if (EDGE && acc < TD(0)) acc += k.two_pi;                      // (only the start phase of secam.py:244 can be negative: pi f > 0 afterwards)
        T sn, cs;
        sincos_(T(acc), sn, cs);
        return luma_d + (re * cs - im * sn);                           // secam.py:246, 276
    }
};

}  // namespace cm
#endif

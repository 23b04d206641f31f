// cm_stages_pk.h - stage B of the QAM-family demodulators in packed float32 (device only, gfx950).
//
// Everything behind the product detectors runs on PAIRS of signals that share their coefficients: the two detector
// channels (cos, sin) through the low-pass and the decimator, (u, v) through the comb combination, the
// pre-correction low-pass and the colour matrix.  v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process such a pair in
// one instruction: the arithmetic per pixel is unchanged, but a wavefront issues ~60 instructions per pixel less,
// and at the 3 waves per SIMD of the wave-pair kernels (cm_kernels.h) instruction issue, not the vector pipe, is what
// a wave runs out of (profiles/r01_pair_notes.md).
//
// Pair layout: lane .x = cos channel / u, lane .y = sin channel / v.  Coefficients are wave-uniform; two of them
// share a VGPR pair and the instruction's op_sel bits pick the half (no register is spent on duplicates).
// The scalar templates of cm_stages.h stay the definition of the arithmetic (tests/sim runs them on the host); the
// forms here perform the same operations in the same order, two at a time.
#ifndef CM_STAGES_PK_H
#define CM_STAGES_PK_H

#include "cm_stages.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
namespace cm {

typedef float pf2 __attribute__((ext_vector_type(2)));

// d = c[H] * x + acc
template <int H>
__device__ __forceinline__ pf2 pk_fma_c(pf2 c, pf2 x, pf2 acc) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(c), "v"(x), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(c), "v"(x), "v"(acc));
#else
    d = c[H] * x + acc;
#endif
    return d;
}
// d = c[H] * x
template <int H>
__device__ __forceinline__ pf2 pk_mul_c(pf2 c, pf2 x) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(c), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(c), "v"(x));
#else
    d = c[H] * x;
#endif
    return d;
}
// the same two with the broadcast operand in an SGPR pair (carriers, matrix columns)
template <int H>
__device__ __forceinline__ pf2 pk_fma_cs(pf2 c, pf2 x, pf2 acc) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(c), "v"(x), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(c), "v"(x), "v"(acc));
#else
    d = c[H] * x + acc;
#endif
    return d;
}
template <int H>
__device__ __forceinline__ pf2 pk_mul_cs(pf2 c, pf2 x) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "s"(c), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "s"(c), "v"(x));
#else
    d = c[H] * x;
#endif
    return d;
}
// d = x[H] * k (k: full pair in SGPRs)
template <int H>
__device__ __forceinline__ pf2 pk_mul_bs(pf2 x, pf2 k) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "s"(k));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "s"(k));
#else
    d = x[H] * k;
#endif
    return d;
}
template <int H>
__device__ __forceinline__ pf2 pk_fma_bs(pf2 x, pf2 k, pf2 acc) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(x), "s"(k), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "s"(k), "v"(acc));
#else
    d = x[H] * k + acc;
#endif
    return d;
}
// d = ksel(k) * x[XH] + acc: x broadcast from half XH of a VGPR pair; k a full coefficient pair - KSEL 0: (k.x, k.y),
// 1: swapped (k.y, k.x), 2: (k.y, k.y) - in SGPRs (KS) or VGPRs
template <int XH, int KSEL, bool KS>
__device__ __forceinline__ pf2 pk_fma_xk(pf2 x, pf2 k, pf2 acc) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
#define CM_PK_XK(OPS)                                                                                  \
    if (KS) asm("v_pk_fma_f32 %0, %1, %2, %3 " OPS : "=v"(d) : "v"(x), "s"(k), "v"(acc));              \
    else asm("v_pk_fma_f32 %0, %1, %2, %3 " OPS : "=v"(d) : "v"(x), "v"(k), "v"(acc));
    if (XH == 0 && KSEL == 0) { CM_PK_XK("op_sel:[0,0,0] op_sel_hi:[0,1,1]") }
    else if (XH == 0 && KSEL == 1) { CM_PK_XK("op_sel:[0,1,0] op_sel_hi:[0,0,1]") }
    else if (XH == 0 && KSEL == 2) { CM_PK_XK("op_sel:[0,1,0] op_sel_hi:[0,1,1]") }
    else if (XH == 1 && KSEL == 0) { CM_PK_XK("op_sel:[1,0,0] op_sel_hi:[1,1,1]") }
    else if (XH == 1 && KSEL == 1) { CM_PK_XK("op_sel:[1,1,0] op_sel_hi:[1,0,1]") }
    else { CM_PK_XK("op_sel:[1,1,0] op_sel_hi:[1,1,1]") }
#undef CM_PK_XK
#else
    const float kx = KSEL == 0 ? k.x : k.y, ky = KSEL == 1 ? k.x : k.y;
    d = pf2{kx * x[XH] + acc.x, ky * x[XH] + acc.y};
#endif
    return d;
}
template <int XH, bool KS>
__device__ __forceinline__ pf2 pk_mul_xk(pf2 x, pf2 k) {   // d = k * x[XH]
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (XH == 0) {
        if (KS) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "s"(k));
        else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(k));
    } else {
        if (KS) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "s"(k));
        else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(k));
    }
#else
    d = k * x[XH];
#endif
    return d;
}
// (v, -): a pair whose LOW half is v for the broadcasting forms above (their op_sel never reads the other half, which stays
// undefined: no move is spent on a duplicate)
__device__ __forceinline__ pf2 pk_lo(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    pf2 p = __builtin_nondeterministic_value(p);
    p.x = v;
    return p;
#else
    return pf2{v, v};
#endif
}
// d = (x[0] * k.x, -(x[0] * k.y)): the product of a real sample with the conjugate of a carrier pair in SGPRs (exact negation)
__device__ __forceinline__ pf2 pk_mul_bs_conj(pf2 x, pf2 k) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "s"(k));
#else
    d = pf2{x[0] * k.x, -(x[0] * k.y)};
#endif
    return d;
}
// (a.x, b.x) / (a.y, b.y): one v_pk_mov_b32 each (two complex samples -> their real parts / their imaginary parts)
__device__ __forceinline__ pf2 pk_lolo(pf2 a, pf2 b) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
#else
    d = pf2{a.x, b.x};
#endif
    return d;
}
__device__ __forceinline__ pf2 pk_hihi(pf2 a, pf2 b) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
#else
    d = pf2{a.y, b.y};
#endif
    return d;
}
__device__ __forceinline__ pf2 pk_add(pf2 a, pf2 b) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
#else
    d = a + b;
#endif
    return d;
}
__device__ __forceinline__ pf2 pk_mul(pf2 a, pf2 b) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
#else
    d = a * b;
#endif
    return d;
}
__device__ __forceinline__ void pin_pair(pf2 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
}

// One scalar of a uniform block, moved to a VGPR on its own.  Without the barrier the optimiser merges neighbouring
// elements into overlapping vector loads of the by-value kernel argument copy, which then stays in scratch memory.
__device__ __forceinline__ float take(const float &x) {
    float v = x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

__device__ __forceinline__ float take_s(const float &x) {   // the same, staying in an SGPR
    float v = x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(v));
#endif
    return v;
}

// ---- coefficient blocks, two per VGPR pair ------------------------------------------------------
struct TapsPk {
    pf2 c2[5];   // (c[0], c[1]) ... (c[8], c[9])
    pf2 c0;      // (c0, c0)
    template <int I>
    __device__ __forceinline__ void load_one(const Taps<float> &t) {
        c2[I] = pf2{take(t.c[2 * I]), take(t.c[2 * I + 1])};
        pin_pair(c2[I]);
        if constexpr (I + 1 < 5) load_one<I + 1>(t);
    }
    __device__ __forceinline__ void load(const Taps<float> &t) {
        load_one<0>(t);
        c0 = pf2{take(t.c0), take(t.c0)};
        pin_pair(c0);
    }
};
// d = tap(I) * x + acc with tap(I) = c[I < 10 ? I : 19 - I]
template <int I>
__device__ __forceinline__ pf2 tap_fma(const TapsPk &k, pf2 x, pf2 acc) {
    constexpr int idx = I < 10 ? I : 19 - I;
    return pk_fma_c<idx & 1>(k.c2[idx >> 1], x, acc);
}

template <int NSEC>
struct SosPk {
    static constexpr int NB = (NSEC + 1) / 2;
    pf2 a[NSEC];      // (na1, na2) of section j
    pf2 b[NB];        // b1 of sections 2i, 2i + 1
    pf2 b2[NB];       // b2 likewise (general numerators only)
    template <int J>
    __device__ __forceinline__ void load_a(const SosK<float, NSEC> &k) {
        a[J] = pf2{take(k.na1[J]), take(k.na2[J])};
        pin_pair(a[J]);
        if constexpr (J + 1 < NSEC) load_a<J + 1>(k);
    }
    template <int I>
    __device__ __forceinline__ void load_b(const SosK<float, NSEC> &k, bool with_b2) {
        constexpr int lo = 2 * I, hi = 2 * I + 1 < NSEC ? 2 * I + 1 : lo;
        constexpr float keep = 2 * I + 1 < NSEC ? 1.f : 0.f;
        b[I] = pf2{take(k.b1[lo]), take(k.b1[hi]) * keep};
        pin_pair(b[I]);
        if (with_b2) {
            b2[I] = pf2{take(k.b2[lo]), take(k.b2[hi]) * keep};
            pin_pair(b2[I]);
        }
        if constexpr (I + 1 < NB) load_b<I + 1>(k, with_b2);
    }
    __device__ __forceinline__ void load(const SosK<float, NSEC> &k, bool with_b2) {
        load_a<0>(k);
        load_b<0>(k, with_b2);
    }
};

// ---- transposed-form half-band decimator on a pair (cm_stages.h: HalfbandChain) ---------------------
struct HalfbandChainPk {
    pf2 s[19];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < 19; ++j) s[j] = pf2{0.f, 0.f};
    }
    template <int J>
    __device__ __forceinline__ void update(const TapsPk &k, pf2 x) {
        s[J] = tap_fma<J + 1>(k, x, s[J + 1]);
        if constexpr (J < 17) update<J + 1>(k, x);
    }
    __device__ __forceinline__ pf2 push_pair(const TapsPk &k, pf2 even, pf2 odd) {
#ifdef CM_EXP_NO_PKFIR   /* timing experiment: the packed decimator costs nothing (results are wrong) */
        return pk_add(even, odd);
#endif
        pf2 out = tap_fma<0>(k, odd, s[0]);
        update<0>(k, odd);
        s[18] = pk_mul_c<0>(k.c2[0], odd);
        s[8] = pk_fma_c<0>(k.c0, even, s[8]);   // centre tap lands on output m
        return out;
    }
};

// ---- transposed-form half-band interpolator, TWO pushes at a time, packed along the accumulator index -------------
// HalfbandChain::push updates a[j] <- c[j + 1] x + a[j + 1] (a[-1] = the output, a[19] = 0, c symmetric, c[20] = 0).  Two
// pushes x0, x1 move every accumulator by two:  a[j] <- c[j + 1] x1 + (c[j + 2] x0 + a[j + 2]).  With the accumulators in
// pairs q[i] = (a[2i - 1], a[2i]):
//     tmp   = (c[2i + 1], c[2i + 2]) x0 + q[i + 1]
//     q[i] <- (c[2i],     c[2i + 1]) x1 + tmp
// Each v_pk_fma_f32 performs exactly two of the scalar chain's FMAs, in the scalar chain's order: the results are
// bit-identical, no extra line state, and 40 scalar FMAs per two steps become 20 packed ones + 1.  The coefficient pairs are
// wave-uniform: the (even, odd) pairs are TapsPk's (the mirrored half of the symmetric taps by swapping the halves with
// op_sel), the (odd, even) pairs sit in SGPR pairs; x0 / x1 are broadcast from one VGPR pair by op_sel.
struct TapsPkOdd {      // SGPR pairs (c1, c2), (c3, c4), (c5, c6), (c7, c8) and (c0, 0)
    pf2 a[5];
    __device__ __forceinline__ void load(const Taps<float> &t) {
        a[0] = pf2{take_s(t.c[1]), take_s(t.c[2])};
        a[1] = pf2{take_s(t.c[3]), take_s(t.c[4])};
        a[2] = pf2{take_s(t.c[5]), take_s(t.c[6])};
        a[3] = pf2{take_s(t.c[7]), take_s(t.c[8])};
        const float zero = 0.f;
        a[4] = pf2{take_s(t.c[0]), take_s(zero)};
    }
};
struct HalfbandUp2Pk {
    pf2 q[10];          // q[i] = (s[2i - 1], s[2i]) of HalfbandChain; q[0].x is not state
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < 10; ++j) q[j] = pf2{0.f, 0.f};
    }
    // pushes x.x then x.y; returns (the output after the first push, the output after the second)
    __device__ __forceinline__ pf2 push2(const TapsPk &ke, const TapsPkOdd &ko, pf2 x) {
        const float out0 = __builtin_fmaf(ke.c2[0].x, x.x, q[0].y);
        pf2 t;
        t = pk_fma_xk<0, 0, true>(x, ko.a[0], q[1]);  q[0] = pk_fma_xk<1, 0, false>(x, ke.c2[0], t);
        t = pk_fma_xk<0, 0, true>(x, ko.a[1], q[2]);  q[1] = pk_fma_xk<1, 0, false>(x, ke.c2[1], t);
        t = pk_fma_xk<0, 0, true>(x, ko.a[2], q[3]);  q[2] = pk_fma_xk<1, 0, false>(x, ke.c2[2], t);
        t = pk_fma_xk<0, 0, true>(x, ko.a[3], q[4]);  q[3] = pk_fma_xk<1, 0, false>(x, ke.c2[3], t);
        t = pk_fma_xk<0, 2, false>(x, ke.c2[4], q[5]); q[4] = pk_fma_xk<1, 0, false>(x, ke.c2[4], t);   // (c9, c10 = c9)
        t = pk_fma_xk<0, 1, true>(x, ko.a[3], q[6]);  q[5] = pk_fma_xk<1, 1, false>(x, ke.c2[4], t);    // (c11, c12) = (c8, c7)
        t = pk_fma_xk<0, 1, true>(x, ko.a[2], q[7]);  q[6] = pk_fma_xk<1, 1, false>(x, ke.c2[3], t);
        t = pk_fma_xk<0, 1, true>(x, ko.a[1], q[8]);  q[7] = pk_fma_xk<1, 1, false>(x, ke.c2[2], t);
        t = pk_fma_xk<0, 1, true>(x, ko.a[0], q[9]);  q[8] = pk_fma_xk<1, 1, false>(x, ke.c2[1], t);
        t = pk_mul_xk<0, true>(x, ko.a[4]);           q[9] = pk_fma_xk<1, 1, false>(x, ke.c2[0], t);    // (c19, c20) = (c0, 0)
        return pf2{out0, q[0].x};
    }
};

// =============================================================================================
// Stage A of the PAL-D front end (cm_stages.h: PalDFrontA) with two of its three half-band chains in one packed chain:
// lane x = up2(x), lane y = dn2 -> e.  The interpolator runs ONE SAMPLE AHEAD (it is fed x[t + 1] in step t and its
// output is held for the next step), so that both lanes have their input at the same point of a step; each lane executes
// exactly the scalar chain's operations in the scalar chain's order - the results are bit-identical.  Why: v_pk_fma_f32
// does not care which register banks its sources sit in, a three-source v_fma_f32 does (3.0 instead of 2.1 cycles per
// instruction at three waves per SIMD when two sources share a bank, which hipcc's allocation makes the case for two of
// three: profiles/r02_ubench_bank.txt) - and 40 scalar FMAs per step become 20 packed ones.
// =============================================================================================
template <class S>
struct PalDFrontAPk {
    typedef DemodK<float, S> K;
    typedef VPolicy<CM_V_PALD> VP;
    HalfbandChainPk xe;               // (up2(x) one sample ahead, dn2 -> e)
    HalfbandChain<float> up_e;        // up2(e), one push per step (step) ...
    HalfbandUp2Pk up_e2;              // ... or two pushes per packed update (step2); a kernel uses one of the two
    IirState<float, S::NE> bpf;
    float hold_b, a_odd;              // a_odd: up2(x)'s odd output of the step to come

    __device__ __forceinline__ void reset() {
        xe.reset(); up_e.reset(); up_e2.reset(); bpf.reset();
        hold_b = a_odd = 0.f;
    }
    // before step 0: x0 = x[0]
    __device__ __forceinline__ void prime(const TapsPk &kp, float x0) {
        a_odd = xe.push_pair(kp, pf2{0.f, 0.f}, pf2{x0, 0.f}).x;
    }
    // x_next = x[t + 1] (0 beyond the row); the rest as PalDFrontA::step up to e[n3].  ts: the taps as scalars (halves of kp's pairs)
    template <bool EDGE>
    __device__ __forceinline__ float front(const K &k, const TapsPk &kp, const Taps<float> &ts, FrontLatch<float> &la, int t, float x_next,
                                           float x_d10) {
        const int W = k.width;
        const int n1 = t - 10, n2 = n1 - k.q_e, n3 = n2 - 9;
        const bool ODD_E = S::RT ? k.odd_e != 0 : S::ODD_E;
        float a_o = a_odd;
        float a_even = ts.c0 * x_d10;
        float b_even = 0.f, b_odd = 0.f;
        if (!EDGE || (n1 >= 0 && n1 < W + k.q_e)) {
            if (EDGE) {
                if (n1 == W - 1) la.a_last = a_o;
                if (n1 >= W) a_even = a_o = la.a_last;
            }
            const float y0 = iir_bp<VP::VB>(bpf, k.ext, a_even);
            const float y1 = iir_bp<VP::VB>(bpf, k.ext, a_o);
            if (ODD_E) { b_even = hold_b; b_odd = y0; hold_b = y1; } else { b_even = y0; b_odd = y1; }
        }
        if (EDGE && (n2 < 0 || n2 >= W)) b_even = b_odd = 0.f;
        const pf2 out = xe.push_pair(kp, pf2{0.f, b_even}, pf2{x_next, b_odd});
        a_odd = out.x;
        float e = out.y;
        if (EDGE && (n3 < 0 || n3 >= W)) e = 0.f;
        return e;
    }
    template <bool EDGE>
    __device__ __forceinline__ Mid<float> step(const K &k, const TapsPk &kp, const Taps<float> &ts, FrontLatch<float> &la, int t, float x_next,
                                               float x_d10, float e_d10, float &e_out) {
        const float e = front<EDGE>(k, kp, ts, la, t, x_next, x_d10);
        e_out = e;
        Mid<float> m;
        m.odd = up_e.template push<true>(ts, e);
        m.even = ts.c0 * e_d10;
        return m;
    }
    // steps t and t + 1 together: the interpolator of e takes both pushes as one packed update (HalfbandUp2Pk: bit-identical)
    template <bool EDGE>
    __device__ __forceinline__ void step2(const K &k, const TapsPk &kp, const TapsPkOdd &ko, const Taps<float> &ts, FrontLatch<float> &la, int t,
                                          float x_next0, float x_next1, float x_d10_0, float x_d10_1, float e_d10_0, float e_d10_1,
                                          float &e_out0, float &e_out1, Mid<float> &m0, Mid<float> &m1) {
        const float e0 = front<EDGE>(k, kp, ts, la, t, x_next0, x_d10_0);
        const float e1 = front<EDGE>(k, kp, ts, la, t + 1, x_next1, x_d10_1);
        e_out0 = e0;
        e_out1 = e1;
        const pf2 odd = up_e2.push2(kp, ko, pf2{e0, e1});
        m0.odd = odd.x;
        m1.odd = odd.y;
        m0.even = ts.c0 * e_d10_0;
        m1.even = ts.c0 * e_d10_1;
    }
};

// The QAM front end with the band-stop luma (cm_stages.h: QamFrontA<.., true>) the same way: lane x = up2(x) one sample ahead,
// lane y = the luma decimator.  Bit-identical to the scalar chains.
template <class S>
struct QamBsfFrontAPk {
    typedef DemodK<float, S> K;
    typedef VPolicy<CM_V_QAM> VP;
    HalfbandChainPk xy;
    IirState<float, S::NE> bpf;
    IirState<float, S::NR> bsf;
    float hold_b, hold_y, a_odd;

    __device__ __forceinline__ void reset() {
        xy.reset(); bpf.reset(); bsf.reset();
        hold_b = hold_y = a_odd = 0.f;
    }
    __device__ __forceinline__ void prime(const TapsPk &kp, float x0) {
        a_odd = xy.push_pair(kp, pf2{0.f, 0.f}, pf2{x0, 0.f}).x;
    }
    template <bool EDGE>
    __device__ __forceinline__ Mid<float> step(const K &k, const TapsPk &kp, FrontLatch<float> &la, int t, float x_next, float x_d10, float &luma_out) {
        const int W = k.width;
        const int n1 = t - 10;
        const bool ODD_E = S::RT ? k.odd_e != 0 : S::ODD_E, ODD_R = S::RT ? k.odd_r != 0 : S::ODD_R;
        float a_o = a_odd;
        float a_even = kp.c0.x * x_d10;
        if (EDGE) {
            if (n1 == W - 1) la.a_last = a_o;
            if (n1 >= W) a_even = a_o = la.a_last;
        }
        Mid<float> m;
        m.even = m.odd = 0.f;
        if (!EDGE || (n1 >= 0 && n1 < W + k.q_e)) {
            const float y0 = iir_bp<VP::VB>(bpf, k.ext, a_even);
            const float y1 = iir_bp<VP::VB>(bpf, k.ext, a_o);
            if (ODD_E) { m.even = hold_b; m.odd = y0; hold_b = y1; } else { m.even = y0; m.odd = y1; }
        }
        const int nr = n1 - k.q_r;
        float r_even = 0.f, r_odd = 0.f;
        if (!EDGE || (n1 >= 0 && n1 < W + k.q_r)) {
            const float y0 = iir_sym<false>(bsf, k.rem, a_even);
            const float y1 = iir_sym<false>(bsf, k.rem, a_o);
            if (ODD_R) { r_even = hold_y; r_odd = y0; hold_y = y1; } else { r_even = y0; r_odd = y1; }
        }
        if (EDGE && (nr < 0 || nr >= W)) r_even = r_odd = 0.f;
        const pf2 out = xy.push_pair(kp, pf2{0.f, r_even}, pf2{x_next, r_odd});
        a_odd = out.x;
        luma_out = out.y * k.luma_gain;
        return m;
    }
};

template <int NSEC>
struct IirStatePk {
    pf2 s1[NSEC], s2[NSEC];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < NSEC; ++j) s1[j] = s2[j] = pf2{0.f, 0.f};
    }
};
// numerator 1 + b1 z^-1 + z^-2 (cm_stages.h: iir_sym)
template <int J, int NSEC>
__device__ __forceinline__ pf2 iir_sym_pk(IirStatePk<NSEC> &st, const SosPk<NSEC> &k, pf2 x) {
    // order: no instruction reads the result of the one right before it (hipcc pads such pairs of packed
    // instructions with an s_nop)
    pf2 y = pk_add(x, st.s1[J]);
    pf2 t = pk_fma_c<J & 1>(k.b[J >> 1], x, st.s2[J]);
    st.s2[J] = pk_fma_c<1>(k.a[J], y, x);
    st.s1[J] = pk_fma_c<0>(k.a[J], y, t);
    if constexpr (J + 1 < NSEC) return iir_sym_pk<J + 1, NSEC>(st, k, y);
    else return y;
}
// general numerator 1 + b1 z^-1 + b2 z^-2 (cm_stages.h: iir_gen)
template <int J, int NSEC>
__device__ __forceinline__ pf2 iir_gen_pk(IirStatePk<NSEC> &st, const SosPk<NSEC> &k, pf2 x) {
    pf2 y = pk_add(x, st.s1[J]);
    pf2 t = pk_fma_c<J & 1>(k.b[J >> 1], x, st.s2[J]);
    pf2 z = pk_mul_c<J & 1>(k.b2[J >> 1], x);
    st.s1[J] = pk_fma_c<0>(k.a[J], y, t);
    st.s2[J] = pk_fma_c<1>(k.a[J], y, z);
    if constexpr (J + 1 < NSEC) return iir_gen_pk<J + 1, NSEC>(st, k, y);
    else return y;
}

// the same with the coefficients in SGPR pairs (light filters: saves the VGPR copies)
template <int NSEC>
struct SosPkS {
    pf2 a[NSEC], b[NSEC];     // (na1, na2), (b1, b2) of section j
    template <int J>
    __device__ __forceinline__ void load_j(const SosK<float, NSEC> &k) {
        a[J] = pf2{take_s(k.na1[J]), take_s(k.na2[J])};
        b[J] = pf2{take_s(k.b1[J]), take_s(k.b2[J])};
        if constexpr (J + 1 < NSEC) load_j<J + 1>(k);
    }
    __device__ __forceinline__ void load(const SosK<float, NSEC> &k) { load_j<0>(k); }
};
template <int J, int NSEC>
__device__ __forceinline__ pf2 iir_gen_pks(IirStatePk<NSEC> &st, const SosPkS<NSEC> &k, pf2 x) {
    pf2 y = pk_add(x, st.s1[J]);
    pf2 t = pk_fma_cs<0>(k.b[J], x, st.s2[J]);
    pf2 z = pk_mul_cs<1>(k.b[J], x);
    st.s1[J] = pk_fma_cs<0>(k.a[J], y, t);
    st.s2[J] = pk_fma_cs<1>(k.a[J], y, z);
    if constexpr (J + 1 < NSEC) return iir_gen_pks<J + 1, NSEC>(st, k, y);
    else return y;
}

// ---- uniform blocks of stage B in registers ----------------------------------------------------------
template <class S>
struct StageBK {
    TapsPk taps;
    SosPk<S::NL> lpf;
    SosPkS<S::NP> pre;
    pf2 m_u, m_v;                  // colour matrix columns of u and v, rows (r, g): SGPR pairs
    float m_yr, m_yg, m_b[3];      // luma column of r and g; row of b
    __device__ __forceinline__ void load(const DemodK<float, S> &k, bool with_lpf = true) {
        taps.load(k.taps);
        if (with_lpf) lpf.load(k.lpf, false);
        pre.load(k.pre);
        m_u = pf2{take_s(k.m[0][1]), take_s(k.m[1][1])};
        m_v = pf2{take_s(k.m[0][2]), take_s(k.m[1][2])};
        m_yr = take_s(k.m[0][0]);
        m_yg = take_s(k.m[1][0]);
        m_b[0] = take_s(k.m[2][0]); m_b[1] = take_s(k.m[2][1]); m_b[2] = take_s(k.m[2][2]);
    }
};

// =============================================================================================
// The low-pass half of DetectorPk on its own: for the instances whose stage A takes the detector products and the low-pass
// (cm_kernels.h: LCUT) and hands (q_e, q_o) over; stage B then only pushes them into its decimator.  Same operations, same
// order as DetectorPk::step.
// =============================================================================================
template <class S>
struct DetectorLpfPk {
    typedef DemodK<float, S> K;
    IirStatePk<S::NL> lpf;
    pf2 hold;
    __device__ __forceinline__ void reset() {
        lpf.reset();
        hold = pf2{0.f, 0.f};
    }
    template <bool EDGE>
    __device__ __forceinline__ void step(const K &k, const SosPk<S::NL> &lk, pf2 &p_last, int nd, pf2 p_e, pf2 p_o, pf2 &q_e, pf2 &q_o) {
        const int W = k.width;
        const int n5 = nd - k.q_l;
        const bool ODD_L = S::RT ? k.odd_l != 0 : S::ODD_L;
        q_e = q_o = pf2{0.f, 0.f};
        if (!EDGE || (nd >= 0 && nd < W + k.q_l)) {
            if (EDGE) {
                if (nd == W - 1) p_last = p_o;
                if (nd >= W) p_e = p_o = p_last;
            }
            const pf2 y0 = iir_sym_pk<0, S::NL>(lpf, lk, p_e);
            const pf2 y1 = iir_sym_pk<0, S::NL>(lpf, lk, p_o);
            if (ODD_L) { q_e = hold; q_o = y0; hold = y1; } else { q_e = y0; q_o = y1; }
        }
        if (EDGE && (n5 < 0 || n5 >= W)) q_e = q_o = pf2{0.f, 0.f};
    }
};

// =============================================================================================
// Detector (cm_stages.h) on the pair (cos channel, sin channel).
// =============================================================================================
template <class S>
struct DetectorPk {
    typedef DemodK<float, S> K;
    HalfbandChainPk dn;
    IirStatePk<S::NL> lpf;
    pf2 hold;

    __device__ __forceinline__ void reset() {
        dn.reset();
        lpf.reset();
        hold = pf2{0.f, 0.f};
    }
    // p_e, p_o: the incoming pair times {C, S}(2 nd) and {C, S}(2 nd + 1) (the caller forms them: pk_mul_bs)
    template <bool EDGE>
    __device__ __forceinline__ pf2 step(const K &k, const StageBK<S> &kb, pf2 &p_last, int nd, pf2 p_e, pf2 p_o) {
        const int W = k.width;
        const int n5 = nd - k.q_l;
        const bool ODD_L = S::RT ? k.odd_l != 0 : S::ODD_L;
        pf2 q_e = {0.f, 0.f}, q_o = {0.f, 0.f};
        if (!EDGE || (nd >= 0 && nd < W + k.q_l)) {
            if (EDGE) {
                if (nd == W - 1) p_last = p_o;
                if (nd >= W) p_e = p_o = p_last;
            }
#ifdef CM_EXP_NO_LPF   /* timing experiment: the packed detector low-pass costs nothing (results are wrong) */
            pf2 y0 = p_e, y1 = p_o;
#else
            pf2 y0 = iir_sym_pk<0, S::NL>(lpf, kb.lpf, p_e);
            pf2 y1 = iir_sym_pk<0, S::NL>(lpf, kb.lpf, p_o);
#endif
            if (ODD_L) { q_e = hold; q_o = y0; hold = y1; } else { q_e = y0; q_o = y1; }
        }
        if (EDGE && (n5 < 0 || n5 >= W)) q_e = q_o = pf2{0.f, 0.f};
        return dn.push_pair(kb.taps, q_e, q_o);
    }
};

// per-lane constants of the back end as pairs over (u, v)
struct LaneKPk {
    pf2 ks[3], kc[3];      // (u, v) = sum_j ks[j] * Rs[k-j] + kc[j] * Rc[k-j]
    pf2 ks2[3], kc2[3];    // second combination (MINAVG)
    pf2 ra, rb;            // (sn, cs) = ra * C + rb * S, ra = (sph, vcph), rb = (cph, -vsph)
    bool remod;            // sph or cph non-zero
    __device__ __forceinline__ void load(const LaneK<float> &lk, int depth, bool minavg) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            ks[j] = pf2{take(lk.cu[j][0]), take(lk.cv[j][0])};
            kc[j] = pf2{take(lk.cu[j][1]), take(lk.cv[j][1])};
            if (minavg) {
                ks2[j] = pf2{take(lk.cu2[j][0]), take(lk.cv2[j][0])};
                kc2[j] = pf2{take(lk.cu2[j][1]), take(lk.cv2[j][1])};
            }
        }
        (void)depth;
        ra = pf2{take(lk.sph), take(lk.vcph)};
        rb = pf2{take(lk.cph), -take(lk.vsph)};
        remod = lk.sph != 0.f || lk.cph != 0.f;
    }
};

// =============================================================================================
// DemodBack (cm_stages.h) on the pair (u, v).  Base pairs are (Rc, Rs).
// =============================================================================================
template <class S, int DEPTH, bool NOTCH, bool MINAVG>
struct DemodBackPk {
    typedef DemodK<float, S> K;
    IirStatePk<S::NP> pre;
    IirState<float, 1> notch;
    __device__ __forceinline__ void reset() {
        pre.reset();
        notch.reset();
    }
    __device__ __forceinline__ pf2 combine(const LaneKPk &lk, pf2 b0, pf2 b1, pf2 b2) const {
        pf2 uv = pk_mul_c<1>(b0, lk.ks[0]);
        uv = pk_fma_c<0>(b0, lk.kc[0], uv);
        if (DEPTH >= 1) {
            uv = pk_fma_c<0>(b1, lk.kc[1], uv);
            uv = pk_fma_c<1>(b1, lk.ks[1], uv);
        }
        if (DEPTH >= 2) {
            uv = pk_fma_c<0>(b2, lk.kc[2], uv);
            uv = pk_fma_c<1>(b2, lk.ks[2], uv);
        }
        if (MINAVG) {
            pf2 w = pk_mul_c<1>(b0, lk.ks2[0]);
            w = pk_fma_c<0>(b0, lk.kc2[0], w);
            if (DEPTH >= 1) {
                w = pk_fma_c<0>(b1, lk.kc2[1], w);
                w = pk_fma_c<1>(b1, lk.ks2[1], w);
            }
            if (DEPTH >= 2) {
                w = pk_fma_c<0>(b2, lk.kc2[2], w);
                w = pk_fma_c<1>(b2, lk.ks2[2], w);
            }
            uv = pf2{minavg_(uv.x, w.x), minavg_(uv.y, w.y)};
        }
        return uv;
    }
    // (sn, cs) = (sin, +-cos)(phi + 2 n7 cps) times the pre-filter gain; carb = {C[2 n7], S[2 n7]} (SGPRs)
    __device__ __forceinline__ pf2 remod(const LaneKPk &lk, pf2 carb) const {
        pf2 sc = pk_mul_cs<0>(carb, lk.ra);
        return pk_fma_cs<1>(carb, lk.rb, sc);
    }
    // uv: combined chroma at n6; uv_d: the same at n7 = n6 - s_p; y_src: luma source at n7; sc: remod() of sample n7
    template <bool EDGE>
    __device__ __forceinline__ Rgb<float> step(const K &k, const StageBK<S> &kb, const LaneKPk &lk, pf2 &uv_last, int n6, pf2 uv,
                                               pf2 uv_d, float y_src, pf2 sc) {
        const int W = k.width;
        pf2 w = {0.f, 0.f};
        if (!EDGE || (n6 >= 0 && n6 < W + k.s_p)) {
            if (EDGE) {
                if (n6 == W - 1) uv_last = uv;
                if (n6 >= W) uv = uv_last;
            }
            w = iir_gen_pks<0, S::NP>(pre, kb.pre, uv);
        }
        pf2 pr = pk_mul(sc, w);
        float y = (y_src - pr.x) - pr.y;
        if (NOTCH) {
            const int n7 = n6 - k.s_p;
            if (k.notch_gain != 0.f && (!EDGE || (n7 >= 0 && n7 < W))) {
                float yn = iir_sym<false>(notch, k.notch, y) * k.notch_gain;
                if (lk.remod) y = yn;
            }
        }
        // (r, g) as a pair, b alone; matrix columns from SGPRs
        pf2 rg = pk_mul_bs<0>(uv_d, kb.m_u);
        rg = pk_fma_bs<1>(uv_d, kb.m_v, rg);
        Rgb<float> o;
        o.r = fmaf_(kb.m_yr, y, rg.x);
        o.g = fmaf_(kb.m_yg, y, rg.y);
        o.b = fmaf_(kb.m_b[0], y, fmaf_(kb.m_b[1], uv_d.x, kb.m_b[2] * uv_d.y));
        return o;
    }
};

// =============================================================================================
// SECAM decoder with the two quadrature channels as one pair (cm_stages.h: SecamDemod::chroma_step is the
// definition; this is the same stream with (I, Q) = data_up * (cos, -sin) carried through the low-pass two at a time).
// The phase step between consecutive I/Q samples is small for an FM signal inside the channel (|d| < 0.15 rad at
// +-500 kHz): atan(r), r = cross / dot, from five odd terms when dot > 0 and |r| <= 1/4 (truncation 2e-8 rad,
// v_rcp_f32 1 ulp), the library atan2f otherwise (noise, unlocked input, the first samples of a row).
// =============================================================================================
__device__ __forceinline__ float phase_step_fast(float i0, float q0, float i1, float q1) {
    // angle of (i1 + j q1) * conj(i0 + j q0); the cross product with an error-free correction (SecamDemod::phase_step)
    const float t = q0 * i1;
    const float e = __builtin_fmaf(-q0, i1, t);
    const float cross = __builtin_fmaf(i0, q1, -t) + e;
    const float dot = __builtin_fmaf(i0, i1, q0 * q1);
#ifdef CM_EXP_SECAM_ALWAYS_FAST   /* timing experiment: never the library atan2f (results wrong for large steps) */
    if (true) {
#else
    if (dot > 0.f && __builtin_fabsf(cross) <= 0.25f * dot) {
#endif
        const float r = cross * __builtin_amdgcn_rcpf(dot);
        const float z = r * r;
        float p = __builtin_fmaf(z, 1.0f / 9.0f, -1.0f / 7.0f);
        p = __builtin_fmaf(z, p, 1.0f / 5.0f);
        p = __builtin_fmaf(z, p, -1.0f / 3.0f);
        return __builtin_fmaf(r * z, p, r);
    }
    return atan2f(cross, dot);
}

// The same step in two halves for a body that takes its eight phase steps together (the interior of a row): the small-angle
// series for all of them, one wave-uniform branch to the library function when any lane of any of them falls outside.
struct PhaseStep {
    float cross, dot;
    __device__ __forceinline__ void set(float i0, float q0, float i1, float q1) {
        const float t = q0 * i1;
        const float e = __builtin_fmaf(-q0, i1, t);
        cross = __builtin_fmaf(i0, q1, -t) + e;
        dot = __builtin_fmaf(i0, i1, q0 * q1);
    }
    __device__ __forceinline__ bool small() const { return dot > 0.f && __builtin_fabsf(cross) <= 0.25f * dot; }
    __device__ __forceinline__ float series() const {
        const float r = cross * __builtin_amdgcn_rcpf(dot);
        const float z = r * r;
        float p = __builtin_fmaf(z, 1.0f / 9.0f, -1.0f / 7.0f);
        p = __builtin_fmaf(z, p, 1.0f / 5.0f);
        p = __builtin_fmaf(z, p, -1.0f / 3.0f);
        return __builtin_fmaf(r * z, p, r);
    }
    __device__ __forceinline__ float full() const { return small() ? series() : atan2f(cross, dot); }
};

// ---- the phase steps two at a time (round 5; stage B of the wave pair, interior bodies) ----------------------------------------
// A pair holds the same quantity of two phase steps; every v_pk_* below performs exactly the two scalar operations of
// PhaseStep::set / series, in their order: bit-identical results, half the instructions.
__device__ __forceinline__ pf2 pk_fma(pf2 a, pf2 b, pf2 c) {           // a * b + c
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#else
    d = a * b + c;
#endif
    return d;
}
__device__ __forceinline__ pf2 pk_fma_nab(pf2 a, pf2 b, pf2 c) {       // -(a * b) + c, one rounding
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#else
    d = pf2{__builtin_fmaf(-a.x, b.x, c.x), __builtin_fmaf(-a.y, b.y, c.y)};
#endif
    return d;
}
__device__ __forceinline__ pf2 pk_fma_nc(pf2 a, pf2 b, pf2 c) {        // a * b - c, one rounding
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#else
    d = pf2{__builtin_fmaf(a.x, b.x, -c.x), __builtin_fmaf(a.y, b.y, -c.y)};
#endif
    return d;
}
// d = x * k[A] + k[B]: both constants broadcast from the halves of ONE VGPR pair
template <int A, int B>
__device__ __forceinline__ pf2 pk_fma_kk(pf2 x, pf2 k) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (A == 0 && B == 1) asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(k));
    else asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(x), "v"(k));
#else
    d = x * k[A] + k[B];
#endif
    return d;
}
// d = x * y + k[H]
template <int H>
__device__ __forceinline__ pf2 pk_fma_addk(pf2 x, pf2 y, pf2 k) {
    pf2 d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(x), "v"(y), "v"(k));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(y), "v"(k));
#else
    d = x * y + k[H];
#endif
    return d;
}

struct PhaseKPk {          // constants of the packed phase steps, pinned in three VGPR pairs
    pf2 a, b, c;           // (1/9, -1/7), (1/5, -1/3), (2 / pi, 2 / pi)
    __device__ __forceinline__ void load(float two_over_pi) {
        a = pf2{1.0f / 9.0f, -1.0f / 7.0f};
        b = pf2{1.0f / 5.0f, -1.0f / 3.0f};
        c = pf2{two_over_pi, two_over_pi};
        pin_pair(a); pin_pair(b); pin_pair(c);
    }
};

struct PhaseStepPk {
    pf2 cross, dot;
    // (pi, pq) = I / Q of the two earlier samples, (ci, cq) = of the two later ones
    __device__ __forceinline__ void set(pf2 pi, pf2 pq, pf2 ci, pf2 cq) {
        const pf2 t = pk_mul(pq, ci);
        const pf2 e = pk_fma_nab(pq, ci, t);
        cross = pk_add(pk_fma_nc(pi, cq, t), e);
        dot = pk_fma(pi, ci, pk_mul(pq, cq));
    }
    // > 0 in both halves <=> both steps qualify for the series (dot > 0 and |cross| < dot / 4; PhaseStep::small has <=: a tie
    // takes the library function here, whose result agrees to the last bit or two)
    __device__ __forceinline__ float margin() const {
        return __builtin_fminf(__builtin_fmaf(-4.0f, __builtin_fabsf(cross.x), dot.x), __builtin_fmaf(-4.0f, __builtin_fabsf(cross.y), dot.y));
    }
    // the two phase steps times 2 / pi
    __device__ __forceinline__ pf2 series(const PhaseKPk &k) const {
        pf2 rc;
        rc.x = __builtin_amdgcn_rcpf(dot.x);
        rc.y = __builtin_amdgcn_rcpf(dot.y);
        const pf2 r = pk_mul(cross, rc);
        const pf2 z = pk_mul(r, r);
        pf2 p = pk_fma_kk<0, 1>(z, k.a);
        p = pk_fma_addk<0>(z, p, k.b);
        p = pk_fma_addk<1>(z, p, k.b);
        return pk_mul(pk_fma(pk_mul(r, z), p, r), k.c);
    }
    // four packed steps stage by stage: a v_pk_* result feeding the next instruction costs a wait state (s_nop), four
    // independent chains side by side have none
    static __device__ __forceinline__ void set4(PhaseStepPk ph[4], const pf2 pi[4], const pf2 pq[4], const pf2 ci[4], const pf2 cq[4]) {
        pf2 t[4], e[4], x[4], dd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = pk_mul(pq[j], ci[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[j] = pk_mul(pq[j], cq[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = pk_fma_nab(pq[j], ci[j], t[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = pk_fma_nc(pi[j], cq[j], t[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) ph[j].dot = pk_fma(pi[j], ci[j], dd[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) ph[j].cross = pk_add(x[j], e[j]);
    }
    static __device__ __forceinline__ void series4(const PhaseStepPk ph[4], const PhaseKPk &k, pf2 f[4]) {
        pf2 rc[4], r[4], z[4], p[4], rz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rc[j].x = __builtin_amdgcn_rcpf(ph[j].dot.x);
            rc[j].y = __builtin_amdgcn_rcpf(ph[j].dot.y);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = pk_mul(ph[j].cross, rc[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = pk_mul(r[j], r[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = pk_fma_kk<0, 1>(z[j], k.a);
#pragma unroll
        for (int j = 0; j < 4; ++j) rz[j] = pk_mul(r[j], z[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = pk_fma_addk<0>(z[j], p[j], k.b);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = pk_fma_addk<1>(z[j], p[j], k.b);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = pk_fma(rz[j], p[j], r[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = pk_mul(r[j], k.c);
    }
    __device__ __forceinline__ pf2 full(const PhaseKPk &k) const {
        PhaseStep a, b;
        a.cross = cross.x; a.dot = dot.x;
        b.cross = cross.y; b.dot = dot.y;
        return pf2{a.full() * k.c.x, b.full() * k.c.x};
    }
};

// ---- transposed-form half-band decimator with its accumulators in pairs: TWO push_pair steps per packed update ------------------
// HalfbandChain::push_pair (cm_stages.h) pushes the odd sample through the chain and adds the centre tap's product of the even
// sample to s[8].  Two consecutive steps (e0, o0), (e1, o1) are HalfbandUp2Pk::push2's update with o0 / o1 for x0 / x1 - here
// taken from halves of DIFFERENT register pairs, the packed phase steps deliver them that way - plus the two centre-tap
// products where the scalar chain adds them: on s'[8] between the pushes (the low half of the temporary of pair 4) and on
// s''[8] after the second.  Bit-identical to two scalar steps; 20 packed + 3 scalar FMAs instead of 42.
// The guarded row ends push one step at a time on the same registers (push_pair: the scalar chain on the halves).
struct HalfbandDn2Pk {
    pf2 q[10];          // q[i] = (s[2i - 1], s[2i]) of HalfbandChain; q[0].x is not state
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < 10; ++j) q[j] = pf2{0.f, 0.f};
    }
    template <int J> __device__ __forceinline__ float get() const {
        if constexpr ((J & 1) != 0) return q[(J + 1) >> 1].x; else return q[J >> 1].y;
    }
    template <int J> __device__ __forceinline__ void put(float v) {
        if constexpr ((J & 1) != 0) q[(J + 1) >> 1].x = v; else q[J >> 1].y = v;
    }
    template <int J> __device__ __forceinline__ void update(const TapsPk &k, float x) {
        constexpr int t = (J + 1) < 10 ? (J + 1) : 18 - J;
        put<J>(__builtin_fmaf(k.c2[t >> 1][t & 1], x, get<J + 1>()));
        if constexpr (J < 17) update<J + 1>(k, x);
    }
    __device__ __forceinline__ float push_pair(const TapsPk &k, float even, float odd) {
        const float out = __builtin_fmaf(k.c2[0].x, odd, get<0>());
        update<0>(k, odd);
        put<18>(k.c2[0].x * odd);
        put<8>(__builtin_fmaf(k.c0.x, even, get<8>()));
        return out;
    }
    // steps s, s + 1 with (even, odd) = (f0[H], f1[H]) and (f2[H], f3[H]); returns their two outputs
    template <int H>
    __device__ __forceinline__ pf2 push2(const TapsPk &ke, const TapsPkOdd &ko, pf2 f0, pf2 f1, pf2 f2, pf2 f3) {
        const float out0 = __builtin_fmaf(ke.c2[0].x, f1[H], q[0].y);
        pf2 t;
        t = pk_fma_xk<H, 0, true>(f1, ko.a[0], q[1]);  q[0] = pk_fma_xk<H, 0, false>(f3, ke.c2[0], t);
        t = pk_fma_xk<H, 0, true>(f1, ko.a[1], q[2]);  q[1] = pk_fma_xk<H, 0, false>(f3, ke.c2[1], t);
        t = pk_fma_xk<H, 0, true>(f1, ko.a[2], q[3]);  q[2] = pk_fma_xk<H, 0, false>(f3, ke.c2[2], t);
        t = pk_fma_xk<H, 0, true>(f1, ko.a[3], q[4]);  q[3] = pk_fma_xk<H, 0, false>(f3, ke.c2[3], t);
        t = pk_fma_xk<H, 2, false>(f1, ke.c2[4], q[5]);
        t.x = __builtin_fmaf(ke.c0.x, f0[H], t.x);                                                      // centre tap of the first step on s'[8]
        q[4] = pk_fma_xk<H, 0, false>(f3, ke.c2[4], t);
        t = pk_fma_xk<H, 1, true>(f1, ko.a[3], q[6]);  q[5] = pk_fma_xk<H, 1, false>(f3, ke.c2[4], t);
        t = pk_fma_xk<H, 1, true>(f1, ko.a[2], q[7]);  q[6] = pk_fma_xk<H, 1, false>(f3, ke.c2[3], t);
        t = pk_fma_xk<H, 1, true>(f1, ko.a[1], q[8]);  q[7] = pk_fma_xk<H, 1, false>(f3, ke.c2[2], t);
        t = pk_fma_xk<H, 1, true>(f1, ko.a[0], q[9]);  q[8] = pk_fma_xk<H, 1, false>(f3, ke.c2[1], t);
        t = pk_mul_xk<H, true>(f1, ko.a[4]);           q[9] = pk_fma_xk<H, 1, false>(f3, ke.c2[0], t);
        q[4].y = __builtin_fmaf(ke.c0.x, f2[H], q[4].y);                                                // ... of the second on s''[8]
        return pf2{out0, q[0].x};
    }
};

struct SecamDemodKPk {   // uniform blocks of the chroma path in VGPR pairs
    TapsPk taps;
    SosPk<3> lpf;
    __device__ __forceinline__ void load(const SecamDemodK<float> &k) {
        taps.load(k.taps);
        lpf.load(k.lpf, false);
    }
};

struct SecamDemodPk {
    typedef VPolicy<CM_V_SECAM> VP;
    IirState<float, 3> bpf, ybs;
    IirState<float, 1> bell;
    IirState<double, 3> bpf64;      // band-pass + bell of the guarded bodies (cm_stages.h: SecamBp64)
    IirState<double, 1> bell64;
    IirStatePk<3> lp;               // (I, Q)
    IirState<float, 1> deemph;
    HalfbandChain<float> up, dn;
    float cc_last, x_last;
    pf2 p_last, iq_prev, iq_hold;
    int have_prev;

    __device__ __forceinline__ void to32() { convert_state(bpf, bpf64); convert_state(bell, bell64); }
    __device__ __forceinline__ void to64() { convert_state(bpf64, bpf); convert_state(bell64, bell); }
    __device__ __forceinline__ void reset() {
        bpf.reset(); ybs.reset(); bell.reset(); lp.reset(); deemph.reset();
        bpf64.reset(); bell64.reset();
        up.reset(); dn.reset();
        cc_last = x_last = 0.f;
        p_last = iq_prev = iq_hold = pf2{0.f, 0.f};
        have_prev = 0;
    }
    // car_e / car_o: {cos, sin} of the FM reference at 2x samples 2 m2 and 2 m2 + 1 (SGPR pairs).
    // EDGE = false: the caller guarantees that no stage index of this step touches a row boundary (every guard below holds
    // and no latch fires): the interior of a row runs without the wave-uniform branches.  The guarded steps run the
    // band-pass + bell in float64 (e64; the caller converts the states where the kind of body changes: to32 / to64).
    template <bool EDGE = true>
    __device__ __forceinline__ float chroma_step(const SecamDemodK<float> &k, const SecamDemodKPk &kp, const SecamDemodLaneK<float> &lk, int m,
                                                 float cc_now, float ch_d10, pf2 car_e, pf2 car_o, float dc, float &ch_out, const SecamBp64 &e64) {
        const int W = k.width, Lc = k.width + k.preroll;
        const int m1 = m - k.s_b, m2 = m1 - 10, m3 = m2 - k.q_l, m4 = m3 - 9, n = m4 - k.preroll;
        float ch = 0.f;
        if (EDGE) {
            if (m >= 0 && m < Lc + k.s_b) {
                if (m == Lc - 1) cc_last = cc_now;
                if (m >= Lc) cc_now = cc_last;
                const double b = iir_bp<false>(bpf64, e64.bpf, (double)cc_now);
                if (m1 >= 0) ch = (float)(k.has_bell ? iir_bp<false>(bell64, e64.bell, b) : b);
            }
        } else {
            float b = iir_bp<VP::VB>(bpf, k.bpf, cc_now);
            ch = k.has_bell ? iir_bp<false>(bell, k.bell, b) : b;
        }
        if (EDGE && (m1 < 0 || m1 >= Lc)) ch = 0.f;
        ch_out = ch;
        const float a_odd = up.template push<VP::VT>(k.taps, ch);
        const float a_even = k.taps.c0 * ch_d10;
        // data_up = cos part - j sin part (secam.py:143): the sign of Q rides on the pair (1, -1)
        const pf2 sgn = {1.f, -1.f};
        pf2 p_e = pk_mul(pk_mul_bs<0>(pf2{a_even, a_even}, car_e), sgn);
        pf2 p_o = pk_mul(pk_mul_bs<0>(pf2{a_odd, a_odd}, car_o), sgn);
        float f_e = 0.f, f_o = 0.f;
        if (!EDGE || (m2 >= 0 && m2 < Lc + k.q_l)) {
            if (EDGE) {
                if (m2 == Lc - 1) p_last = p_o;
                if (m2 >= Lc) p_e = p_o = p_last;
            }
            pf2 y0 = iir_sym_pk<0, 3>(lp, kp.lpf, p_e);
            pf2 y1 = iir_sym_pk<0, 3>(lp, kp.lpf, p_o);
            if (k.odd_l) {   // odd shift: pair m3 = (odd output of the previous pair, even output of this one)
                const pf2 h = iq_hold;
                iq_hold = y1;
                y1 = y0;
                y0 = h;
            }
            if (!EDGE || (m3 >= 0 && m3 < Lc)) {
                const float d_e = (!EDGE || have_prev) ? phase_step_fast(iq_prev.x, iq_prev.y, y0.x, y0.y) : 0.f;  // secam.py:147: first step is 0
                const float d_o = phase_step_fast(y0.x, y0.y, y1.x, y1.y);
                have_prev = 1;
                iq_prev = y1;
                f_e = d_e * k.two_over_pi;
                f_o = d_o * k.two_over_pi;
            }
        }
        const float g2 = dn.template push_pair<VP::VT>(k.taps, f_e, f_o);   // decimated deviation from fc (cm_stages.h: SecamDemodLaneK)
        float c = 0.f;
        if (!EDGE || (n >= 0 && n < W)) {
            float f2 = (g2 + dc) + lk.off2;                              // 2 (f - fsc)
            f2 = f2 < lk.lo ? lk.lo : (f2 > lk.hi ? lk.hi : f2);         // secam.py:290
            c = iir_gen<false>(deemph, k.deemph, f2 * lk.scale);         // secam.py:291-296
        }
        return c;
    }
    // chroma_step<false> in two halves (exactly its operations, in its order, per step): up to the low-passed (I, Q) of the
    // pair, and from the two phase steps on.  Between them the caller turns y0 / y1 of several steps into phase steps.
    __device__ __forceinline__ void chroma_front_mid(const SecamDemodK<float> &k, const SecamDemodKPk &kp, float cc_now, float ch_d10, pf2 car_e,
                                                     pf2 car_o, float &ch_out, pf2 &y0, pf2 &y1) {
        const float b = iir_bp<VP::VB>(bpf, k.bpf, cc_now);
        const float ch = k.has_bell ? iir_bp<false>(bell, k.bell, b) : b;
        ch_out = ch;
        const float a_odd = up.template push<VP::VT>(k.taps, ch);
        const float a_even = k.taps.c0 * ch_d10;
        const pf2 sgn = {1.f, -1.f};
        const pf2 p_e = pk_mul(pk_mul_bs<0>(pf2{a_even, a_even}, car_e), sgn);
        const pf2 p_o = pk_mul(pk_mul_bs<0>(pf2{a_odd, a_odd}, car_o), sgn);
        y0 = iir_sym_pk<0, 3>(lp, kp.lpf, p_e);
        y1 = iir_sym_pk<0, 3>(lp, kp.lpf, p_o);
        if (k.odd_l) {
            const pf2 h = iq_hold;
            iq_hold = y1;
            y1 = y0;
            y0 = h;
        }
    }
    __device__ __forceinline__ float chroma_back_mid(const SecamDemodK<float> &k, const SecamDemodLaneK<float> &lk, float d_e, float d_o, float dc) {
        const float g2 = dn.template push_pair<VP::VT>(k.taps, d_e * k.two_over_pi, d_o * k.two_over_pi);
        float f2 = (g2 + dc) + lk.off2;
        f2 = f2 < lk.lo ? lk.lo : (f2 > lk.hi ? lk.hi : f2);
        return iir_gen<false>(deemph, k.deemph, f2 * lk.scale);
    }
    template <bool EDGE = true>
    __device__ __forceinline__ float luma_step(const SecamDemodK<float> &k, int n, float x_in) {
        const int W = k.width, j = n + k.s_y;
        float y = 0.f;
        if (!EDGE || (j >= 0 && j < W + k.s_y)) {
            if (EDGE) {
                if (j == W - 1) x_last = x_in;
                if (j >= W) x_in = x_last;
            }
            y = iir_sym<false>(ybs, k.ybs, x_in);
        }
        return y * k.luma_gain;
    }
    __device__ __forceinline__ Rgb<float> finish(const SecamDemodK<float> &k, const SecamDemodLaneK<float> &lk, float luma, float own, float prev) const {
        prev = prev * lk.w_prev;
        const float dr = lk.own_is_db != 0.f ? prev : own;   // secam.py:297-300
        const float db = lk.own_is_db != 0.f ? own : prev;
        Rgb<float> o;
        o.r = fmaf_(k.m[0][0], luma, fmaf_(k.m[0][1], dr, k.m[0][2] * db));
        o.g = fmaf_(k.m[1][0], luma, fmaf_(k.m[1][1], dr, k.m[1][2] * db));
        o.b = fmaf_(k.m[2][0], luma, fmaf_(k.m[2][1], dr, k.m[2][2] * db));
        return o;
    }
};

// =============================================================================================
// The same decoder cut in two for the wave pair (cm_secam_kernels.h: secam_demod_pair_kernel): stage A runs the chroma
// path up to the low-passed (I, Q) pairs, stage B turns them into frequencies and finishes the line.  The two halves
// execute exactly the operations of SecamDemodPk::chroma_step, in the same order.
// =============================================================================================
struct SecamDemodPkA {
    typedef VPolicy<CM_V_SECAM_A> VP;
    IirState<float, 3> bpf;
    IirState<float, 1> bell;
    IirState<double, 3> bpf64;      // band-pass + bell of the guarded bodies (cm_stages.h: SecamBp64)
    IirState<double, 1> bell64;
    IirStatePk<3> lp;               // (I, Q)
    HalfbandChain<float> up;
    float cc_last;
    pf2 p_last, iq_hold;

    __device__ __forceinline__ void to32() { convert_state(bpf, bpf64); convert_state(bell, bell64); }
    __device__ __forceinline__ void to64() { convert_state(bpf64, bpf); convert_state(bell64, bell); }
    __device__ __forceinline__ void reset() {
        bpf.reset(); bell.reset(); lp.reset(); up.reset();
        bpf64.reset(); bell64.reset();
        cc_last = 0.f;
        p_last = iq_hold = pf2{0.f, 0.f};
    }
    // y0, y1: the low-passed (I, Q) of the pair m3 (meaningful where stage B's guards hold).  The guarded step: band-pass
    // + bell in float64 (the caller converts the states where the kind of body changes: to32 / to64)
    __device__ __forceinline__ void step(const SecamDemodK<float> &k, const SecamDemodKPk &kp, int m, float cc_now, float ch_d10, pf2 car_e,
                                         pf2 car_o, float &ch_out, pf2 &y0, pf2 &y1, const SecamBp64 &e64) {
        const int Lc = k.width + k.preroll;
        const int m1 = m - k.s_b, m2 = m1 - 10;
        float ch = 0.f;
        if (m >= 0 && m < Lc + k.s_b) {
            if (m == Lc - 1) cc_last = cc_now;
            if (m >= Lc) cc_now = cc_last;
            const double b = iir_bp<false>(bpf64, e64.bpf, (double)cc_now);
            if (m1 >= 0) ch = (float)(k.has_bell ? iir_bp<false>(bell64, e64.bell, b) : b);
        }
        if (m1 < 0 || m1 >= Lc) ch = 0.f;
        ch_out = ch;
        const float a_odd = up.template push<VP::VT>(k.taps, ch);
        const float a_even = k.taps.c0 * ch_d10;
        const pf2 sgn = {1.f, -1.f};
        pf2 p_e = pk_mul(pk_mul_bs<0>(pf2{a_even, a_even}, car_e), sgn);
        pf2 p_o = pk_mul(pk_mul_bs<0>(pf2{a_odd, a_odd}, car_o), sgn);
        y0 = y1 = pf2{0.f, 0.f};
        if (m2 >= 0 && m2 < Lc + k.q_l) {
            if (m2 == Lc - 1) p_last = p_o;
            if (m2 >= Lc) p_e = p_o = p_last;
#ifdef CM_EXP_SECAM_NO_LPF   /* timing experiment (results wrong) */
            y0 = p_e; y1 = p_o;
#else
            y0 = iir_sym_pk<0, 3>(lp, kp.lpf, p_e);
            y1 = iir_sym_pk<0, 3>(lp, kp.lpf, p_o);
#endif
            if (k.odd_l) {
                const pf2 h = iq_hold;
                iq_hold = y1;
                y1 = y0;
                y0 = h;
            }
        }
    }
    // the same step where no stage index touches a row boundary (the interior bodies)
    __device__ __forceinline__ void step_mid(const SecamDemodK<float> &k, const SecamDemodKPk &kp, float cc_now, float ch_d10, pf2 car_e, pf2 car_o,
                                             float &ch_out, pf2 &y0, pf2 &y1) {
        const float b = iir_bp<VP::VB>(bpf, k.bpf, cc_now);
        // the bell section always runs and a variant without one (secam.py:167-170) takes its input instead: a select on a wave-uniform
        // flag (a branch per step cost more, two copies of the body cost stage A its registers)
        const float bl = iir_bp<false>(bell, k.bell, b);
        const float ch = k.has_bell ? bl : b;
        ch_out = ch;
        const float a_odd = up.template push<VP::VT>(k.taps, ch);
        const float a_even = k.taps.c0 * ch_d10;
        // data_up = cos part - j sin part (secam.py:143): the sign of Q as the instruction's neg_hi
        const pf2 p_e = pk_mul_bs_conj(pk_lo(a_even), car_e);
        const pf2 p_o = pk_mul_bs_conj(pk_lo(a_odd), car_o);
        y0 = iir_sym_pk<0, 3>(lp, kp.lpf, p_e);       // the filter's own pair: an odd shift (k.odd_l) is the caller's, once per body of
        y1 = iir_sym_pk<0, 3>(lp, kp.lpf, p_o);       // four steps, through hold() / set_hold()
    }
    __device__ __forceinline__ pf2 hold() const { return iq_hold; }
    __device__ __forceinline__ void set_hold(pf2 h) { iq_hold = h; }
};

// Stage A with the whole chroma front end in float64 (scalar): for the SECAM shapes whose float32 margin is thin - the
// variants without de-emphasis and sampling rates from about 24 MHz on, where isolated samples near a row end (small
// band-passed sub-carrier, ill-conditioned angle) miss 1e-5 in float32.  (I, Q) leave as float32: rounding
// them keeps their RELATIVE precision, which is what the angle needs.
struct SecamDemodA64 {
    IirState<double, 3> bpf;
    IirState<double, 1> bell;
    IirState<double, 3> lp_i, lp_q;
    HalfbandChain<double> up;
    double cc_last, pi_last, pq_last, i_hold, q_hold;

    __device__ __forceinline__ void to32() {}
    __device__ __forceinline__ void to64() {}

    __device__ __forceinline__ void reset() {
        bpf.reset(); bell.reset(); lp_i.reset(); lp_q.reset(); up.reset();
        cc_last = pi_last = pq_last = i_hold = q_hold = 0.0;
    }
    // car = {cos, sin} of the FM reference at 2x samples 2 m2 and 2 m2 + 1 (float64 table)
    __device__ __forceinline__ void step(const SecamDemodK<double> &k, int m, double cc_now, double ch_d10, const double car[4], double &ch_out,
                                         pf2 &y0, pf2 &y1) {
        const int Lc = k.width + k.preroll;
        const int m1 = m - k.s_b, m2 = m1 - 10;
        double ch = 0.0;
        if (m >= 0 && m < Lc + k.s_b) {
            if (m == Lc - 1) cc_last = cc_now;
            if (m >= Lc) cc_now = cc_last;
            double b = iir_bp<false>(bpf, k.bpf, cc_now);
            if (m1 >= 0) ch = k.has_bell ? iir_bp<false>(bell, k.bell, b) : b;
        }
        if (m1 < 0 || m1 >= Lc) ch = 0.0;
        ch_out = ch;
        const double a_odd = up.template push<false>(k.taps, ch);
        const double a_even = k.taps.c0 * ch_d10;
        double pi_e = a_even * car[0], pq_e = -(a_even * car[1]);
        double pi_o = a_odd * car[2], pq_o = -(a_odd * car[3]);
        y0 = y1 = pf2{0.f, 0.f};
        if (m2 >= 0 && m2 < Lc + k.q_l) {
            if (m2 == Lc - 1) { pi_last = pi_o; pq_last = pq_o; }
            if (m2 >= Lc) { pi_e = pi_o = pi_last; pq_e = pq_o = pq_last; }
            double i0 = iir_sym<false>(lp_i, k.lpf, pi_e), q0 = iir_sym<false>(lp_q, k.lpf, pq_e);
            double i1 = iir_sym<false>(lp_i, k.lpf, pi_o), q1 = iir_sym<false>(lp_q, k.lpf, pq_o);
            if (k.odd_l) {
                const double ih = i_hold, qh = q_hold;
                i_hold = i1; q_hold = q1;
                i1 = i0; q1 = q0;
                i0 = ih; q0 = qh;
            }
            y0 = pf2{(float)i0, (float)q0};
            y1 = pf2{(float)i1, (float)q1};
        }
    }
    // the same step where no stage index touches a row boundary (the interior bodies)
    __device__ __forceinline__ void step_mid(const SecamDemodK<double> &k, double cc_now, double ch_d10, const double car[4], double &ch_out, pf2 &y0,
                                             pf2 &y1) {
        const double b = iir_bp<false>(bpf, k.bpf, cc_now);
        const double ch = k.has_bell ? iir_bp<false>(bell, k.bell, b) : b;
        ch_out = ch;
        const double a_odd = up.template push<false>(k.taps, ch);
        const double a_even = k.taps.c0 * ch_d10;
        const double pi_e = a_even * car[0], pq_e = -(a_even * car[1]);
        const double pi_o = a_odd * car[2], pq_o = -(a_odd * car[3]);
        double i0 = iir_sym<false>(lp_i, k.lpf, pi_e), q0 = iir_sym<false>(lp_q, k.lpf, pq_e);
        double i1 = iir_sym<false>(lp_i, k.lpf, pi_o), q1 = iir_sym<false>(lp_q, k.lpf, pq_o);
        y0 = pf2{(float)i0, (float)q0};               // the filter's own pair (an odd shift is the caller's: hold() / set_hold())
        y1 = pf2{(float)i1, (float)q1};
    }
    // the held sample crosses to stage B as float32 either way, so the float value is all the interior bodies keep of it
    __device__ __forceinline__ pf2 hold() const { return pf2{(float)i_hold, (float)q_hold}; }
    __device__ __forceinline__ void set_hold(pf2 h) { i_hold = h.x; q_hold = h.y; }
};

struct SecamFinishK {
    pf2 mo_rg, mp_rg;       // per lane: (m[0][c], m[1][c]) of this call's colour-difference signal and of the previous call's (times w_prev)
    float mo_b, mp_b;
    pf2 my_rg;              // (m[0][0], m[1][0]), wave-uniform (an SGPR pair)
    float my_b;
    __device__ __forceinline__ void load(const SecamDemodK<float> &k, const SecamDemodLaneK<float> &lk) {
        const bool db = lk.own_is_db != 0.f;
        mo_rg = pf2{db ? k.m[0][2] : k.m[0][1], db ? k.m[1][2] : k.m[1][1]};
        mp_rg = pf2{(db ? k.m[0][1] : k.m[0][2]) * lk.w_prev, (db ? k.m[1][1] : k.m[1][2]) * lk.w_prev};
        mo_b = db ? k.m[2][2] : k.m[2][1];
        mp_b = (db ? k.m[2][1] : k.m[2][2]) * lk.w_prev;
        my_rg = pf2{take_s(k.m[0][0]), take_s(k.m[1][0])};
        my_b = k.m[2][0];
        pin_pair(mo_rg); pin_pair(mp_rg);
    }
};

// Stage B.  Round 5: the decimator's accumulators live in register pairs (HalfbandDn2Pk) and the last (I, Q) sample in the high
// halves of two pairs, so that the interior bodies of the row can take their eight phase steps as four packed ones and two
// decimator steps per packed update (cm_secam_kernels.h); the guarded bodies at the row ends step through the same registers
// one sample at a time (chroma_step: the operations of SecamDemodPk::chroma_step, in its order).
struct SecamDemodPkB {
    IirState<float, 3> ybs;
    IirState<float, 1> deemph;
    HalfbandDn2Pk dn;
    float x_last;
    pf2 last_i, last_q;     // .y: I / Q of the newest 2x-rate sample (the packed bodies keep their last pair here)
    int have_prev;

    __device__ __forceinline__ void reset() {
        ybs.reset(); deemph.reset(); dn.reset();
        x_last = 0.f;
        last_i = last_q = pf2{0.f, 0.f};
        have_prev = 0;
    }
    // de-emphasis: a first-order section (secam.py:175-177; SosK's b2 = a2 = 0, so iir_gen's second state stays 0)
    __device__ __forceinline__ float deemph_step(const SecamDemodK<float> &k, float x) {
        const float y = x + deemph.s1[0];
        deemph.s1[0] = __builtin_fmaf(k.deemph.na1[0], y, k.deemph.b1[0] * x);
        return y;
    }
    __device__ __forceinline__ float clip_scale(const SecamDemodLaneK<float> &lk, float g2, float dc) {
        const float f2 = (g2 + dc) + lk.off2;                                    // 2 (f - fsc)
        return __builtin_amdgcn_fmed3f(f2, lk.lo, lk.hi) * lk.scale;              // secam.py:290 (lo < hi)
    }
    __device__ __forceinline__ float chroma_step(const SecamDemodK<float> &k, const TapsPk &kp, const SecamDemodLaneK<float> &lk, int m, pf2 y0, pf2 y1,
                                                 float dc) {
        const int W = k.width, Lc = k.width + k.preroll;
        const int m2 = m - k.s_b - 10, m3 = m2 - k.q_l, m4 = m3 - 9, n = m4 - k.preroll;
        float f_e = 0.f, f_o = 0.f;
        if (m2 >= 0 && m2 < Lc + k.q_l && m3 >= 0 && m3 < Lc) {
#ifdef CM_EXP_SECAM_NO_PHASE   /* timing experiment (results wrong) */
            const float d_e = y0.x, d_o = y1.y;
#else
            const float d_e = have_prev ? phase_step_fast(last_i.y, last_q.y, y0.x, y0.y) : 0.f;  // secam.py:147: first step is 0
            const float d_o = phase_step_fast(y0.x, y0.y, y1.x, y1.y);
#endif
            have_prev = 1;
            last_i.y = y1.x;
            last_q.y = y1.y;
            f_e = d_e * k.two_over_pi;
            f_o = d_o * k.two_over_pi;
        }
        const float g2 = dn.push_pair(kp, f_e, f_o);
        float c = 0.f;
        if (n >= 0 && n < W) c = deemph_step(k, clip_scale(lk, g2, dc));         // secam.py:290-296
        return c;
    }
    // interior bodies: from the decimated deviation g2 (the caller takes phase steps and decimator several steps at a time) on
    __device__ __forceinline__ float chroma_back_mid(const SecamDemodK<float> &k, const SecamDemodLaneK<float> &lk, float g2, float dc) {
        return deemph_step(k, clip_scale(lk, g2, dc));
    }
    __device__ __forceinline__ float luma_step_mid(const SecamDemodK<float> &k, float x_in) { return iir_sym<false>(ybs, k.ybs, x_in) * k.luma_gain; }
    __device__ __forceinline__ float luma_step(const SecamDemodK<float> &k, int n, float x_in) {
        const int W = k.width, j = n + k.s_y;
        float y = 0.f;
        if (j >= 0 && j < W + k.s_y) {
            if (j == W - 1) x_last = x_in;
            if (j >= W) x_in = x_last;
            y = iir_sym<false>(ybs, k.ybs, x_in);
        }
        return y * k.luma_gain;
    }
    // (r, g, b) = m . (luma, dr, db) with (dr, db) = (own, previous call's) or the other way round (secam.py:297-300): the matrix
    // columns of "own" and "previous" are picked per lane once (SecamFinishK; w_prev folded into the latter), (r, g) run as a pair
    __device__ __forceinline__ Rgb<float> finish(const SecamFinishK &fk, float luma, float own, float prev) const {
        pf2 rg = pk_mul_xk<0, false>(pk_lo(prev), fk.mp_rg);
        rg = pk_fma_xk<0, 0, false>(pk_lo(own), fk.mo_rg, rg);
        rg = pk_fma_xk<0, 0, true>(pk_lo(luma), fk.my_rg, rg);
        Rgb<float> o;
        o.r = rg.x;
        o.g = rg.y;
        o.b = fmaf_(fk.my_b, luma, fmaf_(fk.mo_b, own, fk.mp_b * prev));
        return o;
    }
};

}  // namespace cm
#endif
#endif

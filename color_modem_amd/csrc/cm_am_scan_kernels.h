// cm_am_scan_kernels.h - Proto-SECAM (ref protosecam.py:74-112) and NIIR / SECAM-IV (ref niir.py:78-164) in small batches: one WAVEFRONT per scan line, as cm_scan_kernels.h does
// for the QAM family and SECAM.  The streaming kernels of cm_am_kernels.h walk a row with one lane (~790 dependent steps: 0.3 - 0.4 ms
// however few rows there are); here lane l owns samples [l C1, (l + 1) C1) of the row - 3 C1 samples of the 3x-rate signals - and
//   * resample_poly(x, 3, 1) / (., 1, 3) (scipy.signal, 61 taps) are polyphase sums over 1x-rate windows of LDS rows: a 3x-rate
//     signal lives in LDS as its three phases (rows R0, R1, R2: z[3 i + j] = Rj[i]), with zero margins;
//   * the recursive filters at the 3x rate are chunked scans over the lanes (scan_iir, cm_scan_kernels.h) with FilterFunction's
//     padding (the last sample repeated) in the lanes beyond the row and its shift as the offset the chunk goes back to LDS at
//     (scan_put3: shift = 3 q - r moves a sample by q places and r phases);
//   * the other colour-difference signal is the neighbouring wave's (workgroup = NW - 1 calls behind one halo wave).
// Same constants, same arithmetic type as the streaming kernels; results differ by the operation order (float32 resolution).
#ifndef CM_AM_SCAN_KERNELS_H
#define CM_AM_SCAN_KERNELS_H

#include "cm_am_kernels.h"
#include "cm_scan_kernels.h"

namespace cm {

struct ScanProtoK {                // decoder constants (device memory, one per plan)
    int32_t width, c1, sparse_taps, pad0;
    float h[kAmTaps + 3];          // 3 h (Taps3); sparse_taps: h[3 q] == 0 except h[30] (firwin at 1 / 3: a third-band filter)
    ScanFilter ext, rem, post;     // 3x rate: chunk = 3 c1
    float chroma_gain, luma_gain;
    float m[9];
};
struct ScanProtoModK {             // encoder constants
    int32_t width, c1, sparse_taps, luma_filter, averaging, pad0, pad1, pad2;
    float h[kAmTaps + 3];
    ScanFilter pre;                // 1x rate: chunk = c1
    ScanFilter rem;                // 3x rate: chunk = 3 c1
    float pre_gain, luma_gain;
    float e[9];
};
typedef const __attribute__((address_space(4))) ScanProtoK const_ScanProtoK;
typedef const __attribute__((address_space(4))) ScanProtoModK const_ScanProtoModK;
typedef const __attribute__((address_space(4))) float const_float;

template <int C1> constexpr int scan_proto_wave_floats() { return 5 * (64 * C1 + 2 * kScanMargin); }     // x, R0, R1, R2, own chroma

// u[3 i + j] = sum_q h[3 q + j] x[n0 + i + 10 - q]: resample_poly(x, 3, 1) of this lane's chunk (Up3, cm_am_stages.h).
// X: the 1x-rate row with zero margins.  SPARSE: phase 0 is the centre tap alone.
template <int C1, bool SPARSE>
__device__ __forceinline__ void scan_up3(const lds_float *X, int n0, const_float *h, float (&u)[3 * C1]) {
    float w[C1 + 24];              // w[k] = x[n0 - 12 + k]
#pragma unroll
    for (int q = 0; q < (C1 + 24) / 4; ++q) {
        const f4 t = *(const lds_f4 *)(X + n0 - 12 + 4 * q);
        w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (SPARSE && j == 0) {
#pragma unroll
            for (int i = 0; i < C1; ++i) u[3 * i] = h[30] * w[i + 12];
            continue;
        }
        float acc[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; 3 * q + j < kAmTaps; ++q) {
            const float t = h[3 * q + j];
#pragma unroll
            for (int i = 0; i < C1; ++i) acc[i] = fmaf_(t, w[i + 22 - q], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < C1; ++i) u[3 * i + j] = acc[i];
    }
}
// u[3 (W - 1) + 2]: the last sample of the interpolated row (FilterFunction's padding value), by every lane
__device__ __forceinline__ float scan_up3_last(const lds_float *X, int W, const_float *h) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; 3 * q + 2 < kAmTaps; ++q) acc = fmaf_(h[3 * q + 2], X[W - 1 + 10 - q], acc);
    return acc;
}
// y[n0 + i] = sum_k h[k] z[3 (n0 + i) + 30 - k], z in its three phase rows: resample_poly(z, 1, 3) (Dn3; the caller's gain holds
// the 1 / 3 that turns the interpolator's taps into the decimator's)
template <int C1, bool SPARSE>
__device__ __forceinline__ void scan_dn3(const lds_float *R0, const lds_float *R1, const lds_float *R2, int n0, const_float *h, float (&y)[C1]) {
    float w[C1 + 24];
    auto window = [&](const lds_float *R) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < (C1 + 24) / 4; ++q) {
            const f4 t = *(const lds_f4 *)(R + n0 - 12 + 4 * q);
            w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
        }
    };
    // k = 3 q: z index 3 (n + 10 - q): row 0;  k = 3 q + 1: 3 (n + 9 - q) + 2: row 2;  k = 3 q + 2: 3 (n + 9 - q) + 1: row 1
    window(R0);
    if (SPARSE) {
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = h[30] * w[i + 12];
    } else {
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = 0.f;
#pragma unroll
        for (int q = 0; 3 * q < kAmTaps; ++q) {
            const float t = h[3 * q];
#pragma unroll
            for (int i = 0; i < C1; ++i) y[i] = fmaf_(t, w[i + 22 - q], y[i]);
        }
    }
    window(R2);
#pragma unroll
    for (int q = 0; 3 * q + 1 < kAmTaps; ++q) {
        const float t = h[3 * q + 1];
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = fmaf_(t, w[i + 21 - q], y[i]);
    }
    window(R1);
#pragma unroll
    for (int q = 0; 3 * q + 2 < kAmTaps; ++q) {
        const float t = h[3 * q + 2];
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = fmaf_(t, w[i + 21 - q], y[i]);
    }
}
// This lane's chunk of a 3x-rate sequence (v[k] = a[3 n0 + k]) back to the phase rows `shift` samples earlier: sample m lands at
// m - shift = 3 (n0 - q) + k + r with q = ceil(shift / 3), r = 3 q - shift.  Then what fell before sample 0 and from sample 3 len on
// is zeroed again (the decimator reads zeros there: resample_poly zero-extends).
template <int C1, int R>
__device__ __forceinline__ void scan_put3_r(lds_float *R0, lds_float *R1, lds_float *R2, const float (&v)[3 * C1], int base) {
#pragma unroll
    for (int k = 0; k < 3 * C1; ++k) {
        lds_float *row = (k + R) % 3 == 0 ? R0 : ((k + R) % 3 == 1 ? R1 : R2);
        row[base + (k + R) / 3] = v[k];
    }
}
template <int C1>
__device__ __forceinline__ void scan_put3(lds_float *R0, lds_float *R1, lds_float *R2, const float (&v)[3 * C1], int n0, int shift, int len, int lane) {
    const int q = (shift + 2) / 3, r = 3 * q - shift;
    if (r == 0) scan_put3_r<C1, 0>(R0, R1, R2, v, n0 - q);
    else if (r == 1) scan_put3_r<C1, 1>(R0, R1, R2, v, n0 - q);
    else scan_put3_r<C1, 2>(R0, R1, R2, v, n0 - q);
    R0[lane - kScanMargin] = 0.f; R1[lane - kScanMargin] = 0.f; R2[lane - kScanMargin] = 0.f;
    R0[len + lane] = 0.f; R1[len + lane] = 0.f; R2[len + lane] = 0.f;
    scan_fence();
}
// this lane's chunk of a 3x-rate sequence out of the phase rows
template <int C1>
__device__ __forceinline__ void scan_get3(const lds_float *R0, const lds_float *R1, const lds_float *R2, int n0, float (&v)[3 * C1]) {
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const f4 a = *(const lds_f4 *)(R0 + n0 + 4 * q), b = *(const lds_f4 *)(R1 + n0 + 4 * q), c = *(const lds_f4 *)(R2 + n0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[3 * (4 * q + e)] = a[e];
            v[3 * (4 * q + e) + 1] = b[e];
            v[3 * (4 * q + e) + 2] = c[e];
        }
    }
}

// =============================================================================================================================
// Proto-SECAM decoder (ProtoDemod::step + proto_demod_kernel's finish): workgroup = NW - 1 calls behind one halo wave.
// =============================================================================================================================
template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void proto_demod_scan_kernel(const Geom g, const AmGeom am, const ScanProtoK *km) {
    constexpr int C3 = 3 * C1, N1 = 64 * C1, MG = kScanMargin, kRow = N1 + 2 * MG;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanProtoK &k = *(const_ScanProtoK *)km;
    const long long c = (long long)blockIdx.x * (NW - 1) - 1 + w;
    const LaneCall lc = locate_call_at(g, c, w >= 1);
    const bool alive = c >= 0 && c < g.total_calls;
    lds_float *wave = (lds_float *)scan_lds + w * scan_proto_wave_floats<C1>();
    lds_float *X = wave + MG, *R0 = X + kRow, *R1 = R0 + kRow, *R2 = R1 + kRow, *COWN = R2 + kRow;
    const int W = g.W, L = 3 * W;
    const int n0 = lane * C1, m0 = 3 * n0;
    const bool sparse = k.sparse_taps != 0;
    // ---- the row ------------------------------------------------------------------------------------------------------------
    {
        const float *xp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, g.Wp);
        X[lane - MG] = 0.f;
        X[N1 + lane] = 0.f;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            const int n = n0 + 4 * q;
            f4 t = {0.f, 0.f, 0.f, 0.f};
            if (alive && n < g.Wp) t = scan_load4<U8>(xp, n);
            if (n + 3 >= W) {
                if (n >= W) t.x = 0.f;
                if (n + 1 >= W) t.y = 0.f;
                if (n + 2 >= W) t.z = 0.f;
                if (n + 3 >= W) t.w = 0.f;
            }
            *(lds_f4 *)(X + n) = t;
        }
        scan_fence();
    }
    float u[C3], v[C3];
    if (sparse) scan_up3<C1, true>(X, n0, k.h, u);
    else scan_up3<C1, false>(X, n0, k.h, u);
    const float u_last = scan_up3_last(X, W, k.h);
#pragma unroll
    for (int i = 0; i < C3; ++i) u[i] = m0 + i >= L ? u_last : u[i];              // FilterFunction's padding (utils.py:31-33)
    // ---- chroma: band-pass -> |.| -> low-pass at the 3x rate -> decimator (protosecam.py:96-103) -------------------------------
#pragma unroll
    for (int i = 0; i < C3; ++i) v[i] = u[i];
    scan_iir<C3>(v, k.ext, lane);
    scan_put3<C1>(R0, R1, R2, v, n0, k.ext.shift, W, lane);
    {
        const float c_last = __builtin_fabsf(R2[W - 1]);
        scan_get3<C1>(R0, R1, R2, n0, v);
#pragma unroll
        for (int i = 0; i < C3; ++i) v[i] = m0 + i >= L ? c_last : __builtin_fabsf(v[i]);
    }
    scan_iir<C3>(v, k.post, lane);
    scan_put3<C1>(R0, R1, R2, v, n0, k.post.shift, W, lane);
    float chroma[C1];
    if (sparse) scan_dn3<C1, true>(R0, R1, R2, n0, k.h, chroma);
    else scan_dn3<C1, false>(R0, R1, R2, n0, k.h, chroma);
#pragma unroll
    for (int i = 0; i < C1; ++i) chroma[i] = fmaf_(k.chroma_gain, chroma[i], -1.f);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) *(lds_f4 *)(COWN + n0 + 4 * q) = f4{chroma[4 * q], chroma[4 * q + 1], chroma[4 * q + 2], chroma[4 * q + 3]};
    scan_fence();
    // ---- luma: band-stop at the 3x rate -> decimator (protosecam.py:110-111) ---------------------------------------------------
    scan_iir<C3>(u, k.rem, lane);
    scan_put3<C1>(R0, R1, R2, u, n0, k.rem.shift, W, lane);
    float luma[C1];
    if (sparse) scan_dn3<C1, true>(R0, R1, R2, n0, k.h, luma);
    else scan_dn3<C1, false>(R0, R1, R2, n0, k.h, luma);
    __syncthreads();
    if (w < 1 || !alive || !lc.store_ok) return;
    // ---- finish: this call's and the previous call's colour difference, matrix -------------------------------------------------
    const lds_float *CPREV = COWN - scan_proto_wave_floats<C1>();
    const bool alt = am.line.alternate((long long)am.frame_base + lc.frame, lc.line);
    const float w_prev = lc.kk > 0 ? 1.f : 0.f;          // protosecam.py:93-94: the first line of a run has no previous chroma
    float *op = U8 ? (float *)scan_row<true>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride)
                   : g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride + n0;
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const f4 pv = *(const lds_f4 *)(CPREV + n0 + 4 * q);
        f4 o[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            const float y = luma[i] * k.luma_gain, prev = pv[e] * w_prev;
            const float dr = alt ? prev : chroma[i], db = alt ? chroma[i] : prev;       // protosecam.py:105-108
#pragma unroll
            for (int p = 0; p < 3; ++p) o[p][e] = fmaf_(k.m[3 * p], y, fmaf_(k.m[3 * p + 1], dr, k.m[3 * p + 2] * db));
        }
        if (n0 + 4 * q < g.Wp) {
            if (U8) scan_store_rgb4_u8(op, n0 + 4 * q, o[0], o[1], o[2]);
            else {
#pragma unroll
                for (int p = 0; p < 3; ++p) *(f4 *)(op + p * g.out_plane_stride + 4 * q) = o[p];
            }
        }
    }
}

// =============================================================================================================================
// Proto-SECAM encoder (ProtoMod::step + proto_mod_kernel's caller side): one wavefront per call, NW independent calls per workgroup,
// no barrier - inside ColorAveragingModem (comb.py:141-152) the previous call's row comes straight from memory.
// =============================================================================================================================
template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void proto_mod_scan_kernel(const Geom g, const AmGeom am, const ScanProtoModK *km) {
    constexpr int C3 = 3 * C1, N1 = 64 * C1, MG = kScanMargin, kRow = N1 + 2 * MG;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanProtoModK &k = *(const_ScanProtoModK *)km;
    const long long c = (long long)blockIdx.x * NW + w;
    if (c >= g.total_calls) return;                       // (no barrier in this kernel)
    const LaneCall lc = locate_call_at(g, c, true);
    const LaneCall lp = locate_call_at(g, c > 0 ? c - 1 : 0, true);      // the previous call of the list (line averaging)
    lds_float *wave = (lds_float *)scan_lds + w * scan_proto_wave_floats<C1>();
    lds_float *X = wave + MG, *R0 = X + kRow, *R1 = R0 + kRow, *R2 = R1 + kRow, *PC = R2 + kRow;
    const int W = g.W, L = 3 * W;
    const int n0 = lane * C1, m0 = 3 * n0;
    const int depth = k.averaging;
    const long long frame = (long long)am.frame_base + lc.frame;
    const int line = depth ? lc.line - 2 : lc.line;       // the line that is modulated
    const bool alt = am.line.alternate(frame, line);
    float cph, sph;
    {
        const double phi = am.line.start_phase(frame, line);
        cph = (float)cos(phi);
        sph = (float)sin(phi);
    }
    const bool have_prev = depth != 0 && lc.kk > 0;
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, row_stride);
    const float *rq = scan_row<U8>(g.in, lp.frame, g.in_frame_stride, lp.src_row, row_stride);
    // (luma, d) of one sample: proto_mod_kernel's body, the same operation order
    auto yd_of = [&](float r, float gg, float b, float rr, float gr, float br, float &y, float &d) {
        y = fmaf_(k.e[0], r, fmaf_(k.e[1], gg, k.e[2] * b));
        float dr = fmaf_(k.e[3], r, fmaf_(k.e[4], gg, k.e[5] * b));
        float db = fmaf_(k.e[6], r, fmaf_(k.e[7], gg, k.e[8] * b));
        if (have_prev) {
            const float yp = fmaf_(k.e[0], rr, fmaf_(k.e[1], gr, k.e[2] * br));
            const float drp = fmaf_(k.e[3], rr, fmaf_(k.e[4], gr, k.e[5] * br));
            const float dbp = fmaf_(k.e[6], rr, fmaf_(k.e[7], gr, k.e[8] * br));
            y = yp;                                  // comb.py:147
            dr = 0.5f * (dr + drp);                  // comb.py:148-149
            db = 0.5f * (db + dbp);
        }
        d = alt ? db : dr;                           // protosecam.py:75-78
    };
    float y[C1], d[C1];
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const int n = n0 + 4 * q;
        f4 a[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = b[p] = f4{0.f, 0.f, 0.f, 0.f};
        if (n < g.Wp) {
            scan_load_rgb4<U8>(rp, g.in_plane_stride, n, a);
            if (have_prev) scan_load_rgb4<U8>(rq, g.in_plane_stride, n, b);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float yy, dd;
            yd_of(a[0][e], a[1][e], a[2][e], b[0][e], b[1][e], b[2][e], yy, dd);
            y[4 * q + e] = n + e < W ? yy : 0.f;
            d[4 * q + e] = dd;
        }
    }
    {   // FilterFunction pads with the last sample (utils.py:31-33): d[W - 1], by every lane
        float ar, ag, ab, br = 0.f, bg = 0.f, bb = 0.f, yl, dl;
        scan_load_rgb1<U8>(rp, g.in_plane_stride, W - 1, ar, ag, ab);
        if (have_prev) scan_load_rgb1<U8>(rq, g.in_plane_stride, W - 1, br, bg, bb);
        yd_of(ar, ag, ab, br, bg, bb, yl, dl);
#pragma unroll
        for (int i = 0; i < C1; ++i) d[i] = n0 + i >= W ? dl : d[i];
    }
    // ---- chroma: pre-correction low-pass at the 1x rate, output s_c samples earlier (protosecam.py:80-83) ----------------------
    scan_iir<C1>(d, k.pre, lane);
    PC[lane - MG] = 0.f;
    scan_put<C1>(PC, d, n0, k.pre.shift);
    // ---- luma: resample_poly(., 3, 1) -> band-stop -> resample_poly(., 1, 3) (protosecam.py:84-86) -------------------------------
    if (k.luma_filter) {
        X[lane - MG] = 0.f;
        X[N1 + lane] = 0.f;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) *(lds_f4 *)(X + n0 + 4 * q) = f4{y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
        scan_fence();
        float u[C3];
        const bool sparse = k.sparse_taps != 0;
        if (sparse) scan_up3<C1, true>(X, n0, k.h, u);
        else scan_up3<C1, false>(X, n0, k.h, u);
        const float u_last = scan_up3_last(X, W, k.h);
#pragma unroll
        for (int i = 0; i < C3; ++i) u[i] = m0 + i >= L ? u_last : u[i];
        scan_iir<C3>(u, k.rem, lane);
        scan_put3<C1>(R0, R1, R2, u, n0, k.rem.shift, W, lane);
        if (sparse) scan_dn3<C1, true>(R0, R1, R2, n0, k.h, y);
        else scan_dn3<C1, false>(R0, R1, R2, n0, k.h, y);
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] *= k.luma_gain;
    }
    if (!lc.store_ok) return;
    float *op = (float *)scan_row<U8>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        if (n0 + 4 * q >= g.Wp) continue;
        const f4 tc = *(const lds_f4 *)(PC + n0 + 4 * q);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * q + e;
            const f2 cs = ((const_f2 *)am.carrier)[n < W ? n : W - 1];
            const float cosp = fmaf_(cph, cs.x, -(sph * cs.y));                  // cos(phi + n step)
            const float chroma = fmaf_(0.125f * k.pre_gain, tc[e], 0.125f);
            o[e] = fmaf_(cosp, chroma, y[4 * q + e]);                            // protosecam.py:87-90
        }
        scan_store4<U8>(op, n0 + 4 * q, o);
    }
}

// =============================================================================================================================
// NIIR / SECAM-IV decoder (ref niir.py:106-164; NiirFront / NiirSyn / NiirBack / niir_finish, cm_am_stages.h) with one wavefront per
// call, NW - 1 calls behind one halo wave.  3x-rate signals of a call in LDS (three phase rows each):
//   P  band-pass output M, then phasemod_up = c_pm M / S       S  low-pass of |M|       T  products on their way to the decimators;
//                                                                                          on the first line of a run first the band-passed
//                                                                                          synthetic reference (niir.py:107-110)
// The previous call's phasemod_up is the neighbouring wave's P (one barrier).  The carrier's derivative (niir.py:127-129) reads its
// two neighbours out of the rows, so the one-triple delay of the streaming form is not needed.
// =============================================================================================================================
struct ScanNiirK {
    int32_t width, c1, sparse_taps, pad0;
    float h[kAmTaps + 3];
    ScanFilter bp, lp;             // 3x rate: chunk = 3 c1
    float c_pm, g_b, sat_gain, alt_scale, third, pad1, pad2, pad3;
    float m[9];
};
typedef const __attribute__((address_space(4))) ScanNiirK const_ScanNiirK;
// The decoder's HUE PATH IS FLOAT64 (round 4; cm_am_stages.h: NiirHue has the reasons): interpolator, band-pass, low-pass, the quotient
// M / S, the hue products and the decimators that read them.  The saturation and niir_finish stay float32.
struct ScanNiirK64 {
    double h[kAmTaps + 3];
    ScanFilterD bp, lp;
    double c_pm, alt_scale;
};
typedef const __attribute__((address_space(4))) ScanNiirK64 const_ScanNiirK64;
typedef const __attribute__((address_space(4))) double const_double;
typedef __attribute__((address_space(3))) double lds_double;
// rows of a wave, in floats: the row x | M, then P (phasemod_up) as three phase rows of doubles | S, then the hue products as three rows of doubles
template <int C1> constexpr int scan_niir_wave_floats() { return 13 * (64 * C1 + 2 * kScanMargin); }

template <int C1>
__device__ __forceinline__ void scan_up3_d(const lds_float *X, int n0, const_double *h, double (&u)[3 * C1]) {
    double w[C1 + 24];             // w[k] = x[n0 - 12 + k]
#pragma unroll
    for (int q = 0; q < (C1 + 24) / 4; ++q) {
        const f4 t = *(const lds_f4 *)(X + n0 - 12 + 4 * q);
        w[4 * q] = (double)t.x; w[4 * q + 1] = (double)t.y; w[4 * q + 2] = (double)t.z; w[4 * q + 3] = (double)t.w;
    }
#pragma unroll
    for (int i = 0; i < C1; ++i) u[3 * i] = h[30] * w[i + 12];          // third-band taps: phase 0 is the centre tap alone
#pragma unroll
    for (int j = 1; j < 3; ++j) {
        double acc[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) acc[i] = 0.0;
#pragma unroll
        for (int q = 0; 3 * q + j < kAmTaps; ++q) {
            const double t = h[3 * q + j];
#pragma unroll
            for (int i = 0; i < C1; ++i) acc[i] = fmaf_(t, w[i + 22 - q], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < C1; ++i) u[3 * i + j] = acc[i];
    }
}
__device__ __forceinline__ double scan_up3_last_d(const lds_float *X, int W, const_double *h) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; 3 * q + 2 < kAmTaps; ++q) acc = fmaf_(h[3 * q + 2], (double)X[W - 1 + 10 - q], acc);
    return acc;
}
// scan_dn3 on phase rows of doubles (third-band taps: of row 0 the centre tap alone)
template <int C1>
__device__ __forceinline__ void scan_dn3_d(const lds_double *R0, const lds_double *R1, const lds_double *R2, int n0, const_double *h, double (&y)[C1]) {
    double w[C1 + 22];             // w[k] = R[n0 - 11 + k]
    auto window = [&](const lds_double *R) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < C1 + 22; ++q) w[q] = R[n0 - 11 + q];
    };
#pragma unroll
    for (int i = 0; i < C1; ++i) y[i] = h[30] * R0[n0 + i];
    window(R2);
#pragma unroll
    for (int q = 0; 3 * q + 1 < kAmTaps; ++q) {
        const double t = h[3 * q + 1];
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = fmaf_(t, w[i + 20 - q], y[i]);
    }
    window(R1);
#pragma unroll
    for (int q = 0; 3 * q + 2 < kAmTaps; ++q) {
        const double t = h[3 * q + 2];
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = fmaf_(t, w[i + 20 - q], y[i]);
    }
}
template <int C1, int R>
__device__ __forceinline__ void scan_put3_d_r(lds_double *R0, lds_double *R1, lds_double *R2, const double (&v)[3 * C1], int base) {
#pragma unroll
    for (int k = 0; k < 3 * C1; ++k) {
        lds_double *row = (k + R) % 3 == 0 ? R0 : ((k + R) % 3 == 1 ? R1 : R2);
        row[base + (k + R) / 3] = v[k];
    }
}
template <int C1>
__device__ __forceinline__ void scan_margins_d(lds_double *R0, lds_double *R1, lds_double *R2, int len, int lane) {
    R0[lane - kScanMargin] = 0.0; R1[lane - kScanMargin] = 0.0; R2[lane - kScanMargin] = 0.0;
    R0[len + lane] = 0.0; R1[len + lane] = 0.0; R2[len + lane] = 0.0;
    scan_fence();
}
template <int C1>
__device__ __forceinline__ void scan_put3_d(lds_double *R0, lds_double *R1, lds_double *R2, const double (&v)[3 * C1], int n0, int shift, int len, int lane) {
    const int q = (shift + 2) / 3, r = 3 * q - shift;
    if (r == 0) scan_put3_d_r<C1, 0>(R0, R1, R2, v, n0 - q);
    else if (r == 1) scan_put3_d_r<C1, 1>(R0, R1, R2, v, n0 - q);
    else scan_put3_d_r<C1, 2>(R0, R1, R2, v, n0 - q);
    scan_margins_d<C1>(R0, R1, R2, len, lane);
}
template <int C1>
__device__ __forceinline__ void scan_get3_d(const lds_double *R0, const lds_double *R1, const lds_double *R2, int n0, double (&v)[3 * C1]) {
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        v[3 * i] = R0[n0 + i];
        v[3 * i + 1] = R1[n0 + i];
        v[3 * i + 2] = R2[n0 + i];
    }
}

// syn: the plan's float64 tables of the first lines' phase reference (cm_am_plan.h: build_niir_syn): [R_c | R_s] 3 W each, then
// [D_c | D_s | A_c | A_s] W each - the reference, its decimation and the decimation of its derivative for cos / sin(n step); the reference of a
// line is linear in (sin, cos) of its start phase, so a first line needs no second front end and no rows of its own.
template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void niir_demod_scan_kernel(const Geom g, const AmGeom am, const ScanNiirK *km,
                                                                                                            const ScanNiirK64 *km64, const double *syn,
                                                                                                            double line_phase_shift, double bandpass_phase_shift,
                                                                                                            int strip_i) {
    constexpr int C3 = 3 * C1, N1 = 64 * C1, MG = kScanMargin, kRow = N1 + 2 * MG;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanNiirK &k = *(const_ScanNiirK *)km;
    const_ScanNiirK64 &k64 = *(const_ScanNiirK64 *)km64;
    const long long c = (long long)blockIdx.x * (NW - 1) - 1 + w;
    const LaneCall lc = locate_call_at(g, c, w >= 1);
    const bool alive = c >= 0 && c < g.total_calls;
    lds_float *wave = (lds_float *)scan_lds + w * scan_niir_wave_floats<C1>();
    lds_float *X = wave + MG;
    lds_double *PD0 = (lds_double *)(wave + kRow) + MG, *PD1 = PD0 + kRow, *PD2 = PD1 + kRow;        // M, then P = phasemod_up
    lds_double *QD0 = (lds_double *)(wave + 7 * kRow) + MG, *QD1 = QD0 + kRow, *QD2 = QD1 + kRow;    // S, later the hue products
    const int W = g.W, L = 3 * W;
    const int n0 = lane * C1, m0 = 3 * n0;
    const bool first = __builtin_amdgcn_readfirstlane(lc.kk) == 0;     // the first line of a run: its phase reference is synthetic
    // ---- the row ------------------------------------------------------------------------------------------------------------
    float xr[C1];
    {
        const float *xp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, g.Wp);
        X[lane - MG] = 0.f;
        X[N1 + lane] = 0.f;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            const int n = n0 + 4 * q;
            f4 t = {0.f, 0.f, 0.f, 0.f};
            if (alive && n < g.Wp) t = scan_load4<U8>(xp, n);
            if (n + 3 >= W) {
                if (n >= W) t.x = 0.f;
                if (n + 1 >= W) t.y = 0.f;
                if (n + 2 >= W) t.z = 0.f;
                if (n + 3 >= W) t.w = 0.f;
            }
            *(lds_f4 *)(X + n) = t;
            xr[4 * q] = t.x; xr[4 * q + 1] = t.y; xr[4 * q + 2] = t.z; xr[4 * q + 3] = t.w;
        }
        scan_fence();
    }
    {
        // ---- M = band-pass of the interpolated row; S = low-pass of |M| (niir.py:111-114); phasemod_up = c_pm M / S inside the row --------
        double d[C3];
        scan_up3_d<C1>(X, n0, k64.h, d);
        const double u_last = scan_up3_last_d(X, W, k64.h);
#pragma unroll
        for (int i = 0; i < C3; ++i) d[i] = m0 + i >= L ? u_last : d[i];
        scan_iir_d<C3>(d, k64.bp, lane);
        scan_put3_d<C1>(PD0, PD1, PD2, d, n0, k64.bp.shift, W, lane);
        scan_get3_d<C1>(PD0, PD1, PD2, n0, d);
        {
            const double a_last = __builtin_fabs(PD2[W - 1]);
#pragma unroll
            for (int i = 0; i < C3; ++i) d[i] = m0 + i >= L ? a_last : __builtin_fabs(d[i]);
        }
        scan_iir_d<C3>(d, k64.lp, lane);
        scan_put3_d<C1>(QD0, QD1, QD2, d, n0, k64.lp.shift, W, lane);
        // P over M, a lane's own samples in place (M and S come back out of the rows: nothing of the chunk stays in registers)
#pragma unroll
        for (int i = 0; i < C3; ++i) {
            lds_double *pr = (i % 3 == 0 ? PD0 : (i % 3 == 1 ? PD1 : PD2)) + n0 + i / 3;
            const lds_double *sr = (i % 3 == 0 ? QD0 : (i % 3 == 1 ? QD1 : QD2)) + n0 + i / 3;
            *pr = m0 + i < L ? am_div(k64.c_pm * *pr, *sr) : 0.0;
        }
        scan_fence();
    }
    __syncthreads();
    if (w < 1 || !alive || !lc.store_ok) return;
    // ---- back end (NiirHue::step + the re-modulation decimators) ------------------------------------------------------------------
    const long long frame = (long long)am.frame_base + lc.frame;
    const bool alt = am.line.alternate(frame, lc.line);
    float sin_shift, cos_shift, sin_ps, cos_ps;
    {   // niir.py:117-124, 148-157
        const double shift = alt ? -line_phase_shift : line_phase_shift;
        const double ps = (alt ? 0.0 : line_phase_shift) + 3.14159265358979323846 - bandpass_phase_shift;
        sin_shift = (float)sin(shift); cos_shift = (float)cos(shift);
        sin_ps = (float)sin(ps); cos_ps = (float)cos(ps);
    }
    double syn_s = 0.0, syn_c = 0.0;          // the reference of a first line: syn_s R_c + syn_c R_s
    if (first) {
        const double phi = am.line.start_phase(frame, lc.line - 2);
        const double sg = am.line.alternate(frame, lc.line - 2) ? -1.0 : 1.0;
        syn_s = sg * sin(phi);
        syn_c = sg * cos(phi);
    }
    const lds_double *VD0 = PD0 - scan_niir_wave_floats<C1>() / 2, *VD1 = VD0 + kRow, *VD2 = VD1 + kRow;      // the previous call's phasemod_up
    // sample m of the previous call's phasor / of this call's
    auto prev_at = [&](int m) __attribute__((always_inline)) -> double {
        if (first) return (m >= 0 && m < L) ? fmaf_(syn_s, syn[m], syn_c * syn[L + m]) : 0.0;
        // (the three rows of a signal lie kRow apart: an address, not a choice among three pointers - with a run-time m that choice
        // went through a pointer array in scratch memory and a flat load per sample)
        const int i = m >= 0 ? m / 3 : -1, j = m - 3 * i;
        return VD0[j * kRow + i];
    };
    auto own_at = [&](int m) __attribute__((always_inline)) -> double {
        const int i = m >= 0 ? m / 3 : -1, j = m - 3 * i;
        return PD0[j * kRow + i];
    };
    auto car_at = [&](int m) __attribute__((always_inline)) -> double { return alt ? own_at(m) : prev_at(m); };      // carrier_up (niir.py:117-124)
    auto hue_at = [&](int m) __attribute__((always_inline)) -> double { return alt ? prev_at(m) : own_at(m); };      // the hue-modulated signal
    float sinc[C1], sat[C1], sinphi[C1], cosphi[C1], cosc[C1];
    const bool car_syn = first && !alt;        // the carrier is the synthetic reference: its decimations are tables of the plan
    if (car_syn) {
        const double *Dc = syn + 2 * (size_t)L, *Ds = Dc + W, *Ac = Ds + W, *As = Ac + W;
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const int n = n0 + i < W ? n0 + i : W - 1;
            sinc[i] = (float)fmaf_(syn_s, Dc[n], syn_c * Ds[n]);
            cosc[i] = (float)fmaf_(syn_s, Ac[n], syn_c * As[n]);
        }
    }
    // Five decimations - saturation (S), sincarrier (the carrier's rows), then three signals that are formed into the rows S leaves free:
    // sinphi (hue x carrier), cosphi (hue x the carrier's derivative), coscarrier (the derivative; niir.py:126-146) - as ONE loop around one
    // decimator (inlined five times the kernel outgrows the instruction cache: 215 instead of ~100 us per frame)
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
        if (car_syn && (r == 1 || r == 4)) continue;
        const lds_double *R0 = r == 1 ? (alt ? PD0 : VD0) : QD0;
        if (r >= 2) {
            scan_fence();                            // the rows' last readers are done
#pragma unroll 1
            for (int ii = 0; ii < C1; ++ii) {
                double cb = car_at(m0 + 3 * ii - 1), cm = car_at(m0 + 3 * ii);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int p = m0 + 3 * ii + j;
                    const double ca = car_at(p + 1);
                    const double ac = (p >= 1 && p <= L - 2) ? k64.alt_scale * (ca - cb) : 0.0;       // the carrier's derivative needs its neighbours
                    const double hh = r == 4 ? 1.0 : hue_at(p);
                    (j == 0 ? QD0 : (j == 1 ? QD1 : QD2))[n0 + ii] = hh * (r == 2 ? cm : ac);
                    cb = cm;
                    cm = ca;
                }
            }
            scan_margins_d<C1>(QD0, QD1, QD2, W, lane);
        }
        double y[C1];
        scan_dn3_d<C1>(R0, R0 + kRow, R0 + 2 * kRow, n0, k64.h, y);
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < C1; ++i) sat[i] = (float)y[i];
        } else if (r == 1) {
#pragma unroll
            for (int i = 0; i < C1; ++i) sinc[i] = (float)y[i];
        } else if (r == 2) {
#pragma unroll
            for (int i = 0; i < C1; ++i) sinphi[i] = (float)y[i];
        } else if (r == 3) {
#pragma unroll
            for (int i = 0; i < C1; ++i) cosphi[i] = (float)y[i];
        } else {
#pragma unroll
            for (int i = 0; i < C1; ++i) cosc[i] = (float)y[i];
        }
    }
    // ---- niir_finish: niir.py:134-163, 63-67, 52-61 -------------------------------------------------------------------------------
    const bool strip = strip_i != 0;
    float *op = U8 ? (float *)scan_row<true>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride)
                   : g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride + n0;
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        f4 o[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            const float o_sat = k.sat_gain * sat[i], o_sinc = k.third * sinc[i], o_cosc = k.third * cosc[i];
            const float inv = am_rsqrt(cosphi[i] * cosphi[i] + sinphi[i] * sinphi[i]);
            const float c1 = cosphi[i] * inv, s1 = sinphi[i] * inv;
            const float s2 = -c1 * sin_shift - s1 * cos_shift;
            const float c2 = s1 * sin_shift - c1 * cos_shift;
            float db = o_sat * s2, dr = o_sat * c2;
            const float r = am_sqrt(db * db + dr * dr);
            float luma = xr[i];
            if (strip) {
                const float u = alt ? -r : db, v = alt ? 0.f : dr;
                const float u2 = u * cos_ps - v * sin_ps, v2 = u * sin_ps + v * cos_ps;
                luma = xr[i] - (u2 * o_sinc + v2 * o_cosc);
            }
            const float keep = r > 0.f ? (r - 0.1f > 0.f ? am_div(r - 0.1f, r) : 0.f) : 0.f;
            db *= keep;
            dr *= keep;
#pragma unroll
            for (int p = 0; p < 3; ++p) o[p][e] = row3(k.m[3 * p], k.m[3 * p + 1], k.m[3 * p + 2], luma, db, dr);
        }
        if (n0 + 4 * q < g.Wp) {
            if (U8) scan_store_rgb4_u8(op, n0 + 4 * q, o[0], o[1], o[2]);
            else {
#pragma unroll
                for (int p = 0; p < 3; ++p) *(f4 *)(op + p * g.out_plane_stride + 4 * q) = o[p];
            }
        }
    }
}

// =============================================================================================================================
// NIIR encoder (ref niir.py:78-90, 42-49, 69-76; HueCorrectingNiirModem :181-202; niir_mod_kernel): one wavefront per call, NW
// independent calls per workgroup; the pre-correction low-pass of (db, dr) as one packed scan.
// =============================================================================================================================
struct ScanNiirModK {
    int32_t width, c1, averaging, pad0;
    ScanFilter pre;                // 1x rate: chunk = c1
    float pre_gain, pad1, pad2, pad3;
    float e[9], pad4;
    double ed[6];                  // the db and dr rows of the matrix in float64 (niir_chroma_f64, cm_am_stages.h)
};
typedef const __attribute__((address_space(4))) ScanNiirModK const_ScanNiirModK;

template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void niir_mod_scan_kernel(const Geom g, const AmGeom am, const ScanNiirModK *km, const float *noise) {
    constexpr int N1 = 64 * C1, MG = kScanMargin;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanNiirModK &k = *(const_ScanNiirModK *)km;
    const long long c = (long long)blockIdx.x * NW + w;
    if (c >= g.total_calls) return;                       // (no barrier in this kernel)
    const LaneCall lc = locate_call_at(g, c, true);
    const LaneCall lp = locate_call_at(g, c > 0 ? c - 1 : 0, true);
    lds_float *PB = (lds_float *)scan_lds + w * scan_mod_wave_floats<C1>() + MG, *PR = PB + N1 + 2 * MG;
    const int W = g.W, n0 = lane * C1;
    const int depth = k.averaging;
    const long long frame = (long long)am.frame_base + lc.frame;
    const int line = depth ? lc.line - 2 : lc.line;       // the line that is modulated (niir.py:202)
    const bool alt = am.line.alternate(frame, line);
    float cph, sph;
    {
        const double phi = am.line.start_phase(frame, line);
        cph = (float)cos(phi);
        sph = (float)sin(phi);
    }
    const bool have_prev = lc.kk > 0;
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, row_stride);
    const float *rq = have_prev ? scan_row<U8>(g.in, lp.frame, g.in_frame_stride, lp.src_row, row_stride) : rp;      // niir.py:182-186
    const float *np = noise ? noise + 2LL * lc.call * W : nullptr;
    // (luma, db, dr) with the pedestal of one sample: niir_mod_kernel's body, the same operation order
    auto ybr_of = [&](float r, float gg, float b, float rr, float gr, float br, float nb, float nr, float &y, float &db, float &dr) {
        y = fmaf_(k.e[0], r, fmaf_(k.e[1], gg, k.e[2] * b));
        db = row3(k.e[3], k.e[4], k.e[5], r, gg, b);       // (unit rows keep the sign of a zero: cm_am_stages.h)
        dr = row3(k.e[6], k.e[7], k.e[8], r, gg, b);
        if (depth) {
            const float py = fmaf_(k.e[0], rr, fmaf_(k.e[1], gr, k.e[2] * br));
            const float pdb = row3(k.e[3], k.e[4], k.e[5], rr, gr, br);
            const float pdr = row3(k.e[6], k.e[7], k.e[8], rr, gr, br);
            float odb, odr;
            double ed[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) ed[i] = k.ed[i];
            niir_hue_pixel<U8>(ed, r, gg, b, rr, gr, br, db, dr, pdb, pdr, nb, nr, odb, odr);
            y = py;
            db = odb;
            dr = odr;
        } else {
            double ed[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) ed[i] = k.ed[i];
            niir_offset_pixel<U8>(ed, r, gg, b, db, dr, nb, nr, np != nullptr);
        }
    };
    float y[C1];
    f2 br2[C1];
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const int n = n0 + 4 * q;
        f4 a[3], b[3], nz[2];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = b[p] = f4{0.f, 0.f, 0.f, 0.f};
        nz[0] = nz[1] = f4{0.f, 0.f, 0.f, 0.f};
        if (n < g.Wp) {
            scan_load_rgb4<U8>(rp, g.in_plane_stride, n, a);
            if (depth) scan_load_rgb4<U8>(rq, g.in_plane_stride, n, b);
        }
        if (np) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < W) { nz[0][e] = np[n + e]; nz[1][e] = np[W + n + e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float yy, db, dr;
            ybr_of(a[0][e], a[1][e], a[2][e], b[0][e], b[1][e], b[2][e], nz[0][e], nz[1][e], yy, db, dr);
            y[4 * q + e] = yy;
            br2[4 * q + e] = f2{db, dr};
        }
    }
    {   // FilterFunction pads with the last sample (utils.py:31-33): (db, dr)[W - 1], by every lane
        float ar, ag, ab, pr = 0.f, pg = 0.f, pb = 0.f, yl, dbl, drl;
        scan_load_rgb1<U8>(rp, g.in_plane_stride, W - 1, ar, ag, ab);
        if (depth) scan_load_rgb1<U8>(rq, g.in_plane_stride, W - 1, pr, pg, pb);
        ybr_of(ar, ag, ab, pr, pg, pb, np ? np[W - 1] : 0.f, np ? np[2 * W - 1] : 0.f, yl, dbl, drl);
        if (n0 + C1 > W) {
#pragma unroll
            for (int i = 0; i < C1; ++i) br2[i] = n0 + i >= W ? f2{dbl, drl} : br2[i];
        }
    }
    scan_iir2<C1>(br2, k.pre, lane);
    {
        float s[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) s[i] = br2[i].x;
        scan_put<C1>(PB, s, n0, k.pre.shift);
#pragma unroll
        for (int i = 0; i < C1; ++i) s[i] = br2[i].y;
        scan_put<C1>(PR, s, n0, k.pre.shift);
    }
    if (!lc.store_ok) return;
    float *op = (float *)scan_row<U8>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        if (n0 + 4 * q >= g.Wp) continue;
        const f4 tb = *(const lds_f4 *)(PB + n0 + 4 * q), tr = *(const lds_f4 *)(PR + n0 + 4 * q);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * q + e;
            const f2 cs = ((const_f2 *)am.carrier)[n < W ? n : W - 1];
            const float sn = fmaf_(sph, cs.x, cph * cs.y), cn = fmaf_(cph, cs.x, -(sph * cs.y));
            const float b = k.pre_gain * tb[e], r = k.pre_gain * tr[e];
            const float chroma = alt ? -am_sqrt(b * b + r * r) * sn : fmaf_(b, sn, r * cn);      // niir.py:73-76
            o[e] = y[4 * q + e] + chroma;
        }
        scan_store4<U8>(op, n0 + 4 * q, o);
    }
}

}  // namespace cm
#endif

// Micro-benchmark + numerics prototype (tool only, not product code): the 20-tap half-band FIR chains of the per-scan-line
// kernels as split-float16 Toeplitz products on the matrix pipe ("next round" 2d).
//
// One lane = one scan line, as in the product kernels.  Time is blocked: a lane holds 32 consecutive samples of its line
// in registers.  y[t] = sum_j g[j] x[t - j], j = 0..19, for the 32 outputs of a block is  Y[32 x lines] = A[32 x 48] . X[48 x lines]
// over a window of 48 inputs (the block and the 16 samples before it; the 6 products that reach further back are added with
// v_fma_f32).  v_mfma_f32_32x32x16_f16 takes time on M, lines on N: B[k][n] wants lane l to hold 8 consecutive samples
// (k = 8 (l >> 5) ..) of line l & 31, so each k-step of 16 samples (8 packed registers per lane) goes through 4
// v_permlane32_swap_b32 (lanes 32-63 of the first half <-> lanes 0-31 of the second) and yields the operands of the two
// MFMAs (lines 0-31, lines 32-63); 16 more swaps bring the two 32x32 results back to "all 32 outputs of my own line".
// float32 accuracy comes from splitting data and taps into two float16 pieces each: hi.hi + hi.lo + lo.hi (3 MFMAs,
// float32 accumulation; the dropped lo.lo term is 2^-22).
//
//   check   one FIR through this path against a float64 sum on the host (max error relative to max |y|)
//   mode 0  the wave pair's arithmetic on the vector pipe: FIR chains as v_fma_f32 (20 per step) + IIR sections (4 per step)
//   mode 1  the same FIR chains on the matrix pipe (per 32 steps and chain: 18 MFMA, 32 swaps, ~70 conversions) + the IIRs
//   mode 2  the matrix FIRs alone      mode 3  the IIRs alone      mode 4  the vector FIRs alone
// Model of the pair as in ubench_mfma_fir.hip: stage A = 3 chains + 4 sections, stage B = 2 chains + 27 sections per step.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int kTaps = 20;
constexpr float kTapScale = 64.f;   // taps enter the matrix pipe times 64 (keeps their low pieces out of the float16 subnormals)

__device__ __forceinline__ float fma3(float a, float b, float c) {
    float d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ h8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    u4 v = {a, b, c, d};
    return __builtin_bit_cast(h8, v);
}

// split two samples into packed float16 high and low pieces
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    h2 ph = {h0, h1}, pl = {l0, l1};
    hi = __builtin_bit_cast(uint32_t, ph);
    lo = __builtin_bit_cast(uint32_t, pl);
}

// State of one FIR chain between blocks: the last k-step of the previous block in operand form, and its last 3 samples.
struct FirState {
    uint32_t hi[8], lo[8];
    float p13, p14, p15;
};

__device__ __forceinline__ void fir_state_zero(FirState &s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s.hi[i] = s.lo[i] = 0u;
    s.p13 = s.p14 = s.p15 = 0.f;
}

// One block on pre-split input: nhi / nlo = the block's 32 samples as 16 packed float16 pairs each (high and low pieces),
// t29..t31 = its samples 13..15.  a_tiles: LDS image [3 k-steps][hi, lo][64 lanes] of h8; g17..g19: the last three taps.
__device__ __forceinline__ void fir_block_packed(uint32_t (&nhi)[16], uint32_t (&nlo)[16], float t29, float t30, float t31,
                                                 float (&out)[32], FirState &st, const h8 *a_tiles, float g17, float g18, float g19,
                                                 int lane) {
    // operand form of the two new k-steps: registers [8c .. 8c+3] <- lines 0-31, [8c+4 .. 8c+7] <- lines 32-63
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            auto r = __builtin_amdgcn_permlane32_swap(nhi[8 * c + i], nhi[8 * c + 4 + i], false, false);
            nhi[8 * c + i] = r[0]; nhi[8 * c + 4 + i] = r[1];
            auto q = __builtin_amdgcn_permlane32_swap(nlo[8 * c + i], nlo[8 * c + 4 + i], false, false);
            nlo[8 * c + i] = q[0]; nlo[8 * c + 4 + i] = q[1];
        }
    f16v dl, dh;
#pragma unroll
    for (int i = 0; i < 16; ++i) dl[i] = dh[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const h8 a_hi = a_tiles[(2 * c) * 64 + lane], a_lo = a_tiles[(2 * c + 1) * 64 + lane];
        const uint32_t *wh = c == 0 ? st.hi : nhi + 8 * (c - 1), *wl = c == 0 ? st.lo : nlo + 8 * (c - 1);
        const h8 l_hi = frag(wh[0], wh[1], wh[2], wh[3]), l_lo = frag(wl[0], wl[1], wl[2], wl[3]);
        const h8 u_hi = frag(wh[4], wh[5], wh[6], wh[7]), u_lo = frag(wl[4], wl[5], wl[6], wl[7]);
        dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, l_hi, dl, 0, 0, 0);
        dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, u_hi, dh, 0, 0, 0);
        dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, l_lo, dl, 0, 0, 0);
        dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, u_lo, dh, 0, 0, 0);
        dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, l_hi, dl, 0, 0, 0);
        dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, u_hi, dh, 0, 0, 0);
    }
    const float inv = 1.f / kTapScale;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // (clang folds __builtin_bit_cast of a vector ELEMENT expression to element 0: go through scalars)
        const float lo_f = dl[r], hi_f = dh[r];
        auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_f), __float_as_uint(hi_f), false, false);
        const int t = (r & 3) + 8 * (r >> 2);
        out[t] = __uint_as_float(s[0]) * inv;
        out[t + 4] = __uint_as_float(s[1]) * inv;
    }
    // the 6 products that reach past the 48-sample window (samples 13..15 of the previous block)
    out[0] = fma3(g17, st.p15, fma3(g18, st.p14, fma3(g19, st.p13, out[0])));
    out[1] = fma3(g18, st.p15, fma3(g19, st.p14, out[1]));
    out[2] = fma3(g19, st.p15, out[2]);
#pragma unroll
    for (int i = 0; i < 8; ++i) { st.hi[i] = nhi[8 + i]; st.lo[i] = nlo[8 + i]; }
    st.p13 = t29; st.p14 = t30; st.p15 = t31;
}

__device__ __forceinline__ void fir_block_mfma(const float (&in)[32], float (&out)[32], FirState &st, const h8 *a_tiles,
                                               float g17, float g18, float g19, int lane) {
    uint32_t nhi[16], nlo[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) split2(in[2 * i], in[2 * i + 1], nhi[i], nlo[i]);
    const float t29 = in[13], t30 = in[14], t31 = in[15];   // samples 13..15: what the next block needs beyond its window
    fir_block_packed(nhi, nlo, t29, t30, t31, out, st, a_tiles, g17, g18, g19, lane);
}

// ---- numerics check: 64 lines x n samples, one wave ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_check(const float *x, float *y, const h8 *tiles, const float *g, int n) {
    __shared__ h8 a_tiles[6 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 6 * 64; i += 64) a_tiles[i] = tiles[i];
    __syncthreads();
    FirState st;
    fir_state_zero(st);
    const float g17 = g[17], g18 = g[18], g19 = g[19];
    for (int b = 0; b < n / 32; ++b) {
        float in[32], out[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) in[i] = x[(size_t)lane * n + 32 * b + i];
        fir_block_mfma(in, out, st, a_tiles, g17, g18, g19, lane);
#pragma unroll
        for (int i = 0; i < 32; ++i) y[(size_t)lane * n + 32 * b + i] = out[i];
    }
}

__global__ void k_swapdiag(uint32_t *o) {
    const uint32_t lane = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    o[lane] = r[0];
    o[64 + lane] = r[1];
}

// raw layout probe: A[i][k] = (k == i % 16), B[k][j] = k + 100 j  ->  D[i][j] = i % 16 + 100 j if the documented maps hold
__global__ void k_rawdiag(float *o) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        a[j] = (_Float16)(k == (r % 16) ? 1.f : 0.f);
        b[j] = (_Float16)(float)(k + 100 * r);
    }
    f16v d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[lane * 16 + i] = d[i];
}

// ---- timing model -----------------------------------------------------------------------------------------------------------
template <int N>
struct Sections {
    float s1[N > 0 ? N : 1], s2[N > 0 ? N : 1];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int j = 0; j < N; ++j) s1[j] = s2[j] = 0.f;
    }
    // a cascade of N second-order sections (4 ops each) over one sample
    __device__ __forceinline__ float step(float in, const float (&ic)[6]) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float y = in + s1[j];
            s1[j] = __builtin_fmaf(ic[(j % 3) * 2], in, __builtin_fmaf(-ic[(j % 3) * 2 + 1], y, s2[j]));
            s2[j] = __builtin_fmaf(-ic[(j % 3)], y, in);
            in = y;
        }
        return in;
    }
};

struct VFir {
    float acc[kTaps];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < kTaps; ++i) acc[i] = 0.f;
    }
    __device__ __forceinline__ float step(float xin, const float (&taps)[kTaps]) {
        const float y = fma3(taps[0], xin, acc[0]);
#pragma unroll
        for (int i = 0; i + 2 < kTaps; ++i) acc[i] = fma3(taps[i + 1], xin, acc[i + 1]);
        acc[kTaps - 2] = taps[kTaps - 1] * xin;
        return y;
    }
};

// Stage A of the pair: chain -> 4 sections -> chain -> chain.  Stage B: 12 + 12 sections on two products of the input ->
// a chain each -> 3 sections on their sum.  (FIR chains: 20 taps; sections: 4 ops.)
template <int MODE, bool STAGE_B>
__device__ __forceinline__ void wave_work(float *out_g, const float *coef, const h8 *a_tiles, int blocks) {
    constexpr bool VFIR = MODE == 0 || MODE == 4, MFIR = MODE == 1 || MODE == 2, IIR = MODE == 0 || MODE == 1 || MODE == 3;
    constexpr int kFir = STAGE_B ? 2 : 3;
    const int lane = threadIdx.x & 63;
    float taps[kTaps];
#pragma unroll
    for (int i = 0; i < kTaps; ++i) { taps[i] = coef[i]; asm volatile("" : "+v"(taps[i])); }
    VFir vf[VFIR ? kFir : 1];
    FirState st[MFIR ? kFir : 1];
    if (VFIR) {
#pragma unroll
        for (int f = 0; f < kFir; ++f) vf[f].zero();
    }
    if (MFIR) {
#pragma unroll
        for (int f = 0; f < kFir; ++f) fir_state_zero(st[f]);
    }
    const float g17 = taps[17], g18 = taps[18], g19 = taps[19];
    float ic[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { ic[i] = coef[64 + i]; asm volatile("" : "+v"(ic[i])); }
    Sections<IIR ? (STAGE_B ? 12 : 4) : 0> sa;
    Sections<IIR && STAGE_B ? 12 : 0> sb;
    Sections<IIR && STAGE_B ? 3 : 0> sc;
    sa.zero(); sb.zero(); sc.zero();
    float x = out_g[blockIdx.x * 128 + threadIdx.x];
    float sink = 0.f;
    for (int b = 0; b < blocks; ++b) {
        if (!MFIR) {
            // stepwise, as the product kernels run today
#pragma unroll 1
            for (int t = 0; t < 32; ++t) {
                x = __builtin_amdgcn_fractf(__builtin_fmaf(x, 1.37f, 0.11f));
                if (!STAGE_B) {
                    float v = VFIR ? vf[0].step(x, taps) : x;
                    v = sa.step(v, ic);
                    if (VFIR) { v = vf[1].step(v, taps); v = vf[2].step(v, taps); }
                    sink += v;
                } else {
                    float p = sa.step(x * ic[0], ic), q = sb.step(x * ic[1], ic);
                    if (VFIR) { p = vf[0].step(p, taps); q = vf[1].step(q, taps); }
                    sink += sc.step(p + q, ic);
                }
            }
        } else {
            float cur[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) { x = __builtin_amdgcn_fractf(__builtin_fmaf(x, 1.37f, 0.11f)); cur[t] = x; }
            if (!STAGE_B) {
                float a[32], c[32];
                fir_block_mfma(cur, a, st[0], a_tiles, g17, g18, g19, lane);
#pragma unroll
                for (int t = 0; t < 32; ++t) a[t] = sa.step(a[t], ic);
                fir_block_mfma(a, c, st[1], a_tiles, g17, g18, g19, lane);
                fir_block_mfma(c, a, st[2], a_tiles, g17, g18, g19, lane);
#pragma unroll
                for (int t = 0; t < 32; ++t) sink += a[t];
            } else {
                float q[32], r[32];
                uint32_t nhi[16], nlo[16];
                float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
                for (int t = 0; t < 32; t += 2) {
                    const float p0 = sa.step(cur[t] * ic[0], ic), p1 = sa.step(cur[t + 1] * ic[0], ic);
                    split2(p0, p1, nhi[t / 2], nlo[t / 2]);
                    if (t == 12) e0 = p1;
                    if (t == 14) { e1 = p0; e2 = p1; }
                }
                fir_block_packed(nhi, nlo, e0, e1, e2, q, st[0], a_tiles, g17, g18, g19, lane);
#pragma unroll
                for (int t = 0; t < 32; t += 2) {
                    const float p0 = sb.step(cur[t] * ic[1], ic), p1 = sb.step(cur[t + 1] * ic[1], ic);
                    split2(p0, p1, nhi[t / 2], nlo[t / 2]);
                    if (t == 12) e0 = p1;
                    if (t == 14) { e1 = p0; e2 = p1; }
                }
                fir_block_packed(nhi, nlo, e0, e1, e2, r, st[1], a_tiles, g17, g18, g19, lane);
#pragma unroll
                for (int t = 0; t < 32; ++t) sink += sc.step(r[t] + q[t], ic);
            }
        }
    }
    out_g[blockIdx.x * 128 + threadIdx.x] = sink + x;
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(128, WAVES) void k_mix(float *out, const float *coef, const h8 *tiles, int blocks) {
    __shared__ h8 a_tiles[6 * 64];
    for (int i = threadIdx.x; i < 6 * 64; i += 128) a_tiles[i] = tiles[i];
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0) wave_work<MODE, false>(out, coef, a_tiles, blocks);
    else wave_work<MODE, true>(out, coef, a_tiles, blocks);
}

template <int MODE, int WAVES>
static void run(const char *name, float *d_out, const float *d_coef, const h8 *d_tiles, int blocks) {
    const int grid = 256 * 2 * WAVES * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<MODE, WAVES>), dim3(grid), dim3(128), 0, 0, d_out, d_coef, d_tiles, blocks);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mix<MODE, WAVES>), dim3(grid), dim3(128), 0, 0, d_out, d_coef, d_tiles, blocks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double lane_steps = (double)grid * 64 * blocks * 32;
    const double gsteps = lane_steps / (best * 1e-3) / 1e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f Gpx-steps/s  => 1000 PAL frames of steps in %.2f ms\n", name, WAVES, best, gsteps,
           414.72e6 / (gsteps * 1e9) * 1e3);
    fflush(stdout);
}

static uint16_t f2h_bits(float f) {   // round to nearest even, via the compiler's _Float16 on the host
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

int main(int argc, char **argv) {
    // the 20 non-zero odd taps of 2 * firwin(41, 0.5, ('kaiser', 5.0)), symmetric
    const double half[10] = {0.3167003457, -0.1009301974, 0.0552882839, -0.0343401799, 0.0220231206,
                             -0.0139899732, 0.008556559, -0.0048948343, 0.0025089668, -0.0010514588};
    double g[20];
    for (int i = 0; i < 10; ++i) { g[10 + i] = 2 * half[i]; g[9 - i] = 2 * half[i]; }
    // A tiles: k-step c, lane l (t = l & 31, h = l >> 5), element j: window sample s = 16 c + 8 h + j, tap index t + 16 - s
    std::vector<uint16_t> tiles(6 * 64 * 8);
    for (int c = 0; c < 3; ++c)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int t = l & 31, s = 16 * c + 8 * (l >> 5) + j, k = t + 16 - s;
                const float v = (k >= 0 && k < 20) ? (float)(g[k] * kTapScale) : 0.f;
                const uint16_t hi = f2h_bits(v);
                const uint16_t lo = f2h_bits(v - h2f(hi));
                tiles[((2 * c) * 64 + l) * 8 + j] = hi;
                tiles[((2 * c + 1) * 64 + l) * 8 + j] = lo;
            }
    h8 *d_tiles;
    CK(hipMalloc(&d_tiles, tiles.size() * 2));
    CK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> gf(128, 0.f);
    for (int i = 0; i < 20; ++i) gf[i] = (float)g[i];
    gf[64] = 0.31f; gf[65] = 0.5f; gf[66] = 0.27f; gf[67] = 0.4f; gf[68] = 0.22f; gf[69] = 0.3f;
    float *d_coef;
    CK(hipMalloc(&d_coef, 128 * sizeof(float)));
    CK(hipMemcpy(d_coef, gf.data(), 128 * sizeof(float), hipMemcpyHostToDevice));

    {   // numerics
        const int n = 736;
        std::vector<float> x(64 * n), y(64 * n);
        srand(7);
        for (int l = 0; l < 64; ++l) {
            double w = 0;
            for (int i = 0; i < n; ++i) {
                w = 0.7 * w + 0.3 * ((rand() % 20001) * 1e-4 - 1.0);
                x[l * n + i] = (float)(0.5 + 0.8 * w + 0.1 * std::sin(1.03 * i + l));   // a video-like level with a sub-carrier
            }
        }
        float *dx, *dy;
        CK(hipMalloc(&dx, x.size() * 4));
        CK(hipMalloc(&dy, y.size() * 4));
        CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dx, dy, d_tiles, d_coef, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst32 = 0, ymax = 0;
        for (int l = 0; l < 64; ++l)
            for (int t = 0; t < n; ++t) {
                double ref = 0;
                float ref32 = 0.f;
                for (int j = 0; j < 20; ++j)
                    if (t - j >= 0) { ref += g[j] * (double)x[l * n + t - j]; ref32 = fmaf((float)g[j], x[l * n + t - j], ref32); }
                worst = fmax(worst, fabs(ref - (double)y[l * n + t]));
                worst32 = fmax(worst32, fabs(ref - (double)ref32));
                ymax = fmax(ymax, fabs(ref));
            }
        printf("check: split-f16 MFMA FIR max |err| %.3e (rel. to max |y| = %.3f: %.3e); float32 fmaf chain: %.3e\n", worst, ymax, worst / ymax,
               worst32 / ymax);
        fflush(stdout);
        CK(hipFree(dx));
        CK(hipFree(dy));
    }
    if (argc > 1 && !strcmp(argv[1], "diag")) {
        uint32_t *d_o, h_o[128];
        CK(hipMalloc(&d_o, 512));
        hipLaunchKernelGGL(k_swapdiag, dim3(1), dim3(64), 0, 0, d_o);
        CK(hipMemcpy(h_o, d_o, 512, hipMemcpyDeviceToHost));
        printf("swap(vdst = lane, src = 100 + lane):\n new vdst:");
        for (int i = 0; i < 64; ++i) printf(" %u", h_o[i]);
        printf("\n new src: ");
        for (int i = 0; i < 64; ++i) printf(" %u", h_o[64 + i]);
        printf("\n");
        {
            float *d_r, h_r[1024];
            CK(hipMalloc(&d_r, 4096));
            hipLaunchKernelGGL(k_rawdiag, dim3(1), dim3(64), 0, 0, d_r);
            CK(hipMemcpy(h_r, d_r, 4096, hipMemcpyDeviceToHost));
            const int ls[6] = {0, 1, 5, 32, 33, 37};
            for (int li = 0; li < 6; ++li) {
                printf("raw D lane %d:", ls[li]);
                for (int i = 0; i < 16; ++i) printf(" %g", h_r[ls[li] * 16 + i]);
                printf("\n");
            }
        }
        // delta taps: y = x delayed by d
        for (int d = 0; d < 20; d += 7) {
            std::vector<uint16_t> tl(6 * 64 * 8, 0);
            for (int c = 0; c < 3; ++c)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int t = l & 31, sidx = 16 * c + 8 * (l >> 5) + j, k = t + 16 - sidx;
                        tl[((2 * c) * 64 + l) * 8 + j] = f2h_bits(k == d ? kTapScale : 0.f);
                    }
            CK(hipMemcpy(d_tiles, tl.data(), tl.size() * 2, hipMemcpyHostToDevice));
            std::vector<float> gd(128, 0.f);
            gd[d] = 1.f;
            CK(hipMemcpy(d_coef, gd.data(), 512, hipMemcpyHostToDevice));
            const int n = 64;
            std::vector<float> x(64 * n), y(64 * n);
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < n; ++i) x[l * n + i] = (float)(l * 100 + i + 1);   // exact in float16 up to 2048: lines 0..19
            float *dx, *dy;
            CK(hipMalloc(&dx, x.size() * 4));
            CK(hipMalloc(&dy, y.size() * 4));
            CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dx, dy, d_tiles, d_coef, n);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
            const int lines[4] = {0, 3, 33, 40};
            for (int li = 0; li < 4; ++li) {
                printf("delay %d line %d:", d, lines[li]);
                for (int t = 0; t < n; ++t) printf(" %g", y[lines[li] * n + t]);
                printf("\n");
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "check")) return 0;

    float *d_out;
    const int n = 256 * 4 * 4 * 4 * 64;
    CK(hipMalloc(&d_out, n * sizeof(float)));
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(rand() % 1000) * 1e-3f;
    CK(hipMemcpy(d_out, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    const int blocks = 50;
    run<0, 2>("vector FIR + IIR", d_out, d_coef, d_tiles, blocks);
    run<1, 2>("matrix-f16 FIR + IIR", d_out, d_coef, d_tiles, blocks);
    run<2, 2>("matrix-f16 FIR alone", d_out, d_coef, d_tiles, blocks);
    run<3, 2>("IIR alone", d_out, d_coef, d_tiles, blocks);
    run<4, 2>("vector FIR alone", d_out, d_coef, d_tiles, blocks);
    run<0, 3>("vector FIR + IIR", d_out, d_coef, d_tiles, blocks);
    run<1, 3>("matrix-f16 FIR + IIR", d_out, d_coef, d_tiles, blocks);
    run<2, 3>("matrix-f16 FIR alone", d_out, d_coef, d_tiles, blocks);
    run<3, 3>("IIR alone", d_out, d_coef, d_tiles, blocks);
    run<4, 3>("vector FIR alone", d_out, d_coef, d_tiles, blocks);
    return 0;
}

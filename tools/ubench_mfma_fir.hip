// Micro-benchmark (tool only, not product code): would the half-band FIR chains of the per-scan-line kernels run better
// as float32 outer products on the matrix pipe?
//
// One lane = one scan line, as in the product kernels.  Per step (one pixel per lane) the PAL-D decoder executes about
// 5 half-band FIR updates (20 taps each, transposed form: acc[i] = c[i] * x + acc[i + 1]) and about 31 second-order IIR
// sections (4 FMAs each).  v_mfma_f32_4x4x1_16B_f32 computes, for 16 blocks of 4 lanes, D[i][lane] += A[i] * B[lane] with
// i = 0..3: with B = the lane's input sample and A = four taps this is the transposed-FIR update of four outputs of every
// lane's own row at once, bit-identical to the fmaf chain, with no data movement between lanes.
//
//   mode 0  FIRs on the vector pipe (100 v_fma_f32 per step) + IIRs (124 v_fma_f32 per step)
//   mode 1  FIRs on the matrix pipe (5 x 6 MFMA 4x4x1 per step: 24 output slots per input, 20 useful) + the same IIRs
//   mode 2  the MFMAs alone
//   mode 3  the IIRs alone
//   mode 4  the vector FIRs alone
// Reports ns per step per resident workgroup slot and the implied time for 1000 PAL frames (414.72 M pixel-steps + 7 %
// of edge steps are ignored here) on 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fma3(float a, float b, float c) {
    float d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

constexpr int kTaps = 20, kGroups = 6;

// one wave's share: kFir FIR chains + kSections IIR sections per step
template <int MODE, int kFir, int kSections>
__device__ __forceinline__ void wave_work(float *out, const float *coef, int bodies) {
    constexpr bool VFIR = MODE == 0 || MODE == 4, MFIR = MODE == 1 || MODE == 2, IIR = MODE == 0 || MODE == 1 || MODE == 3;
    const int lane = threadIdx.x & 63;
    float taps[kTaps];
    if (MODE == 0 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < kTaps; ++i) { taps[i] = coef[i]; asm volatile("" : "+v"(taps[i])); }
    }
    float acc[VFIR ? kFir : 1][kTaps];
    v4 grp[MFIR ? kFir : 1][kGroups];
    float ca[MFIR ? kGroups : 1][4];      // lane l holds the tap of output (l % 4) of group g at phase p
    if (VFIR) {
#pragma unroll
        for (int f = 0; f < kFir; ++f)
#pragma unroll
            for (int i = 0; i < kTaps; ++i) acc[f][i] = 0.f;
    }
    if (MFIR) {
#pragma unroll
        for (int f = 0; f < kFir; ++f)
#pragma unroll
            for (int g = 0; g < kGroups; ++g) grp[f][g] = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < kGroups; ++g)
#pragma unroll
            for (int p = 0; p < 4; ++p) { ca[g][p] = coef[32 + (4 * g + (lane & 3) + p) % 24]; asm volatile("" : "+v"(ca[g][p])); }
    }
    float s1[IIR ? kSections : 1], s2[IIR ? kSections : 1];
    float ic[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { ic[i] = coef[64 + i]; asm volatile("" : "+v"(ic[i])); }
    if (IIR) {
#pragma unroll
        for (int j = 0; j < kSections; ++j) s1[j] = s2[j] = 0.f;
    }
    float x = out[blockIdx.x * 128 + threadIdx.x];
    float sink = 0.f, obs = 0.f;
    for (int b = 0; b < bodies; ++b) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float fir_out[kFir];
            if (VFIR) {
#pragma unroll
                for (int f = 0; f < kFir; ++f) {
                    const float xin = f == 0 ? x : fir_out[f - 1];
                    fir_out[f] = fma3(taps[0], xin, acc[f][0]);
#pragma unroll
                    for (int i = 0; i + 2 < kTaps; ++i) acc[f][i] = fma3(taps[i + 1], xin, acc[f][i + 1]);
                    acc[f][kTaps - 2] = taps[kTaps - 1] * xin;
                }
            }
            if (MFIR) {
#pragma unroll
                for (int f = 0; f < kFir; ++f) {
                    // the chain of the product kernel feeds each FIR from an earlier stage of the same step; here the input
                    // is the previous body's output of the FIR before it (available for a whole body)
                    const float xin = f == 0 ? x : grp[f - 1][0][p];
#pragma unroll
                    for (int g = 0; g < kGroups; ++g) {
                        // first step of a body: the groups move up by one through the accumulator operand (C = group g + 1,
                        // D = group g), so the rotation costs no moves
                        const v4 c_in = p != 0 ? grp[f][g] : (g + 1 < kGroups ? grp[f][g + 1] : v4{0.f, 0.f, 0.f, 0.f});
                        grp[f][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ca[g][p], xin, c_in, 0, 0, 0);
                    }
                    fir_out[f] = grp[f][0][p];
                }
            }
            if (!VFIR && !MFIR) {
#pragma unroll
                for (int f = 0; f < kFir; ++f) fir_out[f] = x;
            }
            if (IIR) {
                // chains of three sections (1 + b1 z^-1 + z^-2 numerators: 4 ops per section), inputs from the FIR outputs
#pragma unroll
                for (int j = 0; j < kSections; ++j) {
                    const float in = (j % 3 == 0) ? fir_out[(j / 3) % kFir] : sink;
                    const float y = in + s1[j];
                    s1[j] = __builtin_fmaf(ic[(j % 3) * 2], in, __builtin_fmaf(-ic[(j % 3) * 2 + 1], y, s2[j]));
                    s2[j] = __builtin_fmaf(-ic[(j % 3)], y, in);
                    sink = y;
                    if (j % 3 == 2 || j == kSections - 1) obs += y;   // keeps every chain alive
                }
            } else {
#pragma unroll
                for (int f = 0; f < kFir; ++f) sink += fir_out[f];
            }
            x = __builtin_amdgcn_fractf(__builtin_fmaf(x, 1.37f, 0.11f));
        }
        if (MFIR) {   // a group of four outputs per FIR is complete: rotate the groups (register renaming in the product)
#pragma unroll
            for (int f = 0; f < kFir; ++f) {
                sink += grp[f][0].x + grp[f][0].w;
            }
        }
    }
    out[blockIdx.x * 128 + threadIdx.x] = sink + x + obs;
}

// the product's wave pair: stage A = 3 FIR chains + 4 sections, stage B = 2 FIR chains + 27 sections
template <int MODE, int WAVES>
__global__ __launch_bounds__(128, WAVES) void k_mix(float *out, const float *coef, int bodies) {
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0) wave_work<MODE, 3, 4>(out, coef, bodies);
    else wave_work<MODE, 2, 27>(out, coef, bodies);
}

template <int MODE, int WAVES>
static void run(const char *name, float *d_out, const float *d_coef, int bodies) {
    const int grid = 256 * 2 * WAVES * 4;     // 128-thread workgroups, four rounds of every resident slot
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<MODE, WAVES>), dim3(grid), dim3(128), 0, 0, d_out, d_coef, bodies);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mix<MODE, WAVES>), dim3(grid), dim3(128), 0, 0, d_out, d_coef, bodies);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double lane_steps = (double)grid * 64 * bodies * 4;   // a workgroup (two waves) advances 64 rows
    const double gsteps = lane_steps / (best * 1e-3) / 1e9;      // 1e9 lane-steps (= pixels) per second
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f Gpx-steps/s  => 1000 PAL frames of steps in %.2f ms\n", name, WAVES, best, gsteps,
           414.72e6 / (gsteps * 1e9) * 1e3);
}

int main() {
    float *d_out, *d_coef;
    const int n = 256 * 4 * 4 * 4 * 64;
    CK(hipMalloc(&d_out, n * sizeof(float)));
    CK(hipMalloc(&d_coef, 128 * sizeof(float)));
    std::vector<float> h(n), c(128);
    for (int i = 0; i < n; ++i) h[i] = (float)(rand() % 1000) * 1e-3f;
    for (int i = 0; i < 128; ++i) c[i] = 0.02f + 0.3f * (float)(rand() % 1000) * 1e-3f;
    c[65] = 0.5f; c[67] = 0.4f; c[69] = 0.3f;
    CK(hipMemcpy(d_out, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_coef, c.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
    const int bodies = 400;
    run<0, 2>("vector FIR + IIR", d_out, d_coef, bodies);
    run<1, 2>("matrix FIR + IIR", d_out, d_coef, bodies);
    run<2, 2>("matrix FIR alone", d_out, d_coef, bodies);
    run<3, 2>("IIR alone", d_out, d_coef, bodies);
    run<4, 2>("vector FIR alone", d_out, d_coef, bodies);
    run<1, 3>("matrix FIR + IIR", d_out, d_coef, bodies);
    run<0, 3>("vector FIR + IIR", d_out, d_coef, bodies);
    run<2, 3>("matrix FIR alone", d_out, d_coef, bodies);
    run<3, 3>("IIR alone", d_out, d_coef, bodies);
    run<4, 3>("vector FIR alone", d_out, d_coef, bodies);
    return 0;
}

// Micro-benchmark (tool, not product code): issue rate of v_fma_f64 against v_fma_f32 / v_pk_fma_f32 on gfx950, 1 - 4 waves per SIMD.
// Sizes the float64 front end of the NIIR decoder (cm_am_kernels.h).  hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_f64 tools/ubench_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, const float *coef, int iters) {
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2];
    const double d0 = c0, d1 = c1, d2 = c2;
    float a[24];
    double b[16];
    f2 p[12];
#pragma unroll
    for (int i = 0; i < 24; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = threadIdx.x * 0.001 + i;
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) a[i] = __builtin_fmaf((i & 1) ? c0 : c1, a[i], c2);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = __builtin_fma((i & 1) ? d0 : d1, b[i], d2);
        } else if (MODE == 2) {
            const f2 cc = f2{c0, c1}, dd = f2{c2, c2};
#pragma unroll
            for (int i = 0; i < 12; ++i) p[i] = __builtin_elementwise_fma(cc, p[i], dd);
        } else if (MODE == 3) {      // transposed chain in float64: b[i] = fma(tap, x, b[i + 1])
            const double x = b[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) b[i] = __builtin_fma((i & 1) ? d0 : d1, x, b[i + 1]);
            b[15] = x * d2;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (float)b[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int n_inst, int flop_per_inst, float *out, float *coef) {
    const int iters = 20000;
    for (int w = 1; w <= 4; ++w) {
        const int blocks = 256 * 4 * w;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, coef, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, coef, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double inst_per_simd = (double)w * iters * n_inst;
        printf("%-10s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD  %.1f TFLOP/s\n", name, w, ms, ms * 1e6 / inst_per_simd,
               (double)blocks * 64 * iters * n_inst * flop_per_inst / (ms * 1e-3) * 1e-12);
    }
}

int main() {
    float *out, *coef;
    CK(hipMalloc(&out, 256 * 4 * 4 * 64 * 4));
    CK(hipMemset(out, 0, 256 * 4 * 4 * 64 * 4));
    float hc[4] = {0.99f, 1.01f, 0.001f, 0.5f};
    CK(hipMalloc(&coef, 16)); CK(hipMemcpy(coef, hc, 16, hipMemcpyHostToDevice));
    run<0>("fma_f32", 24, 2, out, coef);
    run<1>("fma_f64", 16, 2, out, coef);
    run<2>("pk_fma_f32", 12, 4, out, coef);
    run<3>("chain_f64", 16, 2, out, coef);
    return 0;
}

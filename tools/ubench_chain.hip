// Issue rate of the kernel's own building blocks (transposed FIR chain via 3-address v_fma_f32 with an SGPR
// coefficient; SOS cascade) at 1-2 waves per SIMD, 64-thread workgroups.  tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__device__ __forceinline__ float fma3(float c, float x, float acc) {
    float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(acc)); return d;
}
__device__ __forceinline__ float fma3v(float c, float x, float acc) {
    float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(c), "v"(x), "v"(acc)); return d;
}
struct K { float c[10]; float c0; float a[12]; };
template <int MODE>
__global__ __launch_bounds__(64, 2) void k(float *out, const K kk, int iters) {
    float s[5][19];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int j = 0; j < 19; ++j) s[q][j] = 0.f;
    float z1[6] = {0,0,0,0,0,0}, z2[6] = {0,0,0,0,0,0};
    float x = threadIdx.x * 0.01f;
    float cv[10], av[12];
#pragma unroll
    for (int i = 0; i < 10; ++i) { cv[i] = kk.c[i]; asm volatile("" : "+v"(cv[i])); }
#pragma unroll
    for (int i = 0; i < 12; ++i) { av[i] = kk.a[i]; asm volatile("" : "+v"(av[i])); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float o = fma3v(cv[0], x, s[q][0]);
#pragma unroll
                for (int j = 0; j < 18; ++j) s[q][j] = fma3v(cv[(j + 1) < 10 ? (j + 1) : 18 - j], x, s[q][j + 1]);
                s[q][18] = cv[0] * x;
                x = o * 0.5f + 0.001f;
            }
        }
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = x;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float y = v + z1[j];
                    float t = fma3v(av[j], v, z2[j]);
                    z1[j] = __builtin_fmaf(av[j + 6], y, t);
                    z2[j] = __builtin_fmaf(av[(j + 3) % 12], y, v);
                    v = y;
                }
                x = v * 0.25f;
            }
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float o = fma3(kk.c[0], x, s[q][0]);
#pragma unroll
                for (int j = 0; j < 18; ++j) s[q][j] = fma3(kk.c[(j + 1) < 10 ? (j + 1) : 18 - j], x, s[q][j + 1]);
                s[q][18] = kk.c[0] * x;
                x = o * 0.5f + 0.001f;
            }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = x;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float y = v + z1[j];
                    float t = fma3(kk.a[j], v, z2[j]);
                    z1[j] = __builtin_fmaf(kk.a[j + 6], y, t);
                    z2[j] = __builtin_fmaf(kk.a[(j + 3) % 12], y, v);
                    v = y;
                }
                x = v * 0.25f;
            }
        }
    }
    float acc = x;
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += s[q][j];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += z1[j] + z2[j];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE> int run(const char *name, int n_inst, float *out, K kk) {
    const int iters = 4000;
    for (int w = 1; w <= 2; ++w) {
        int blocks = 256 * 4 * w;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, kk, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, kk, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-18s waves/SIMD=%d  %.3f ms  %.3fe12 wave-instr/s\n", name, w, ms, (double)blocks * iters * n_inst / ms * 1e-9);
    }
    return 0;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 4 * 4 * 64 * 4));
    K kk; for (int i = 0; i < 10; ++i) kk.c[i] = 0.01f * (i + 1); kk.c0 = 1.f; for (int i = 0; i < 12; ++i) kk.a[i] = 0.05f * (i - 6);
    run<0>("fir5x20", 5 * 21, out, kk);
    run<1>("sos6x4", 4 * (6 * 4 + 1) , out, kk);
    run<2>("fir+sos", 5 * 21 + 4 * 25, out, kk);
    run<3>("fir5x20 vgpr-coef", 5 * 21, out, kk);
    run<4>("sos6x4 vgpr-coef", 4 * 25, out, kk);
    run<5>("fir+sos vgpr-coef", 5 * 21 + 4 * 25, out, kk);
    return 0;
}

// Which property of the transposed-FIR update costs issue rate?  tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__device__ __forceinline__ float fma3(float c, float x, float acc) {
    float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(acc)); return d;
}
__device__ __forceinline__ float fma3v(float c, float x, float acc) {   // coefficient in a VGPR
    float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(c), "v"(x), "v"(acc)); return d;
}
struct K { float c[20]; };
template <int MODE>
__global__ __launch_bounds__(64, 2) void k(float *out, const K kk, int iters) {
    float a[40];
#pragma unroll
    for (int j = 0; j < 40; ++j) a[j] = 0.001f * j;
    float x = threadIdx.x * 0.01f, x2 = x + 1.f;
    float cv[4] = {kk.c[0], kk.c[1], kk.c[2], kk.c[3]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // in-place accumulate, compiler codegen (v_fmac)
#pragma unroll
            for (int j = 0; j < 38; ++j) a[j] = __builtin_fmaf(kk.c[j % 20], x, a[j]);
        } else if (MODE == 1) {   // shift by one register, 3-address asm (the kernel's form)
#pragma unroll
            for (int j = 0; j < 38; ++j) a[j] = fma3(kk.c[j % 20], x, a[j + 1]);
        } else if (MODE == 2) {   // shift by two registers
#pragma unroll
            for (int j = 0; j < 38; ++j) a[j] = fma3(kk.c[j % 20], x, a[j + 2]);
        } else if (MODE == 3) {   // in-place, 3-address asm
#pragma unroll
            for (int j = 0; j < 38; ++j) a[j] = fma3(kk.c[j % 20], x, a[j]);
        } else if (MODE == 4) {   // shift by one, coefficient in VGPR
#pragma unroll
            for (int j = 0; j < 38; ++j) a[j] = fma3v(cv[j % 4], x, a[j + 1]);
        } else if (MODE == 5) {   // shift by one, two alternating multiplicands
#pragma unroll
            for (int j = 0; j < 38; ++j) a[j] = fma3(kk.c[j % 20], (j & 1) ? x : x2, a[j + 1]);
        }
        x = a[0] * 0.5f; x2 = a[1] * 0.5f;
    }
    float acc = x;
#pragma unroll
    for (int j = 0; j < 40; ++j) acc += a[j];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE> int run(const char *name, float *out, K kk) {
    const int iters = 10000;
    for (int w = 1; w <= 2; ++w) {
        int blocks = 256 * 4 * w;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, kk, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, kk, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s waves/SIMD=%d  %.3f ms  %.3fe12 wave-instr/s\n", name, w, ms, (double)blocks * iters * 40 / ms * 1e-9);
    }
    return 0;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 4 * 4 * 64 * 4));
    K kk; for (int i = 0; i < 20; ++i) kk.c[i] = 0.01f * (i + 1);
    run<0>("in-place fmac (compiler)", out, kk);
    run<3>("in-place v_fma 3-addr", out, kk);
    run<1>("shift-1 v_fma 3-addr", out, kk);
    run<2>("shift-2 v_fma 3-addr", out, kk);
    run<4>("shift-1, coef in VGPR", out, kk);
    run<5>("shift-1, 2 multiplicands", out, kk);
    return 0;
}

// Micro-benchmarks that size the per-scanline streaming kernels (tools only, not product code):
//   fma     : independent v_fma_f32 with a scalar coefficient (the transposed-FIR update)
//   pkfma   : v_pk_fma_f32 on float2
//   chain   : transposed FIR chain  acc[i] = fma(c[i], x, acc[i+1])  (dst != accumulate src)
//   mov     : v_mov_b32 rotation
//   gather  : one dwordx4 per lane from 64 different rows (row stride 5760 B)
// Reports shader cycles per wave-instruction for W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k_valu(float *out, unsigned long long *cyc, const float *coef, int iters) {
    float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];
    float a[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) a[i] = threadIdx.x * 0.001f + i;
    f2 p[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    float x = out[threadIdx.x];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 24 independent fma, scalar coefficient
#pragma unroll
            for (int i = 0; i < 24; ++i) a[i] = __builtin_fmaf((i & 1) ? c0 : c1, a[i], c2);
        } else if (MODE == 1) {  // 12 pk_fma = 24 flop-pairs
            f2 cc = f2{c0, c1};
            f2 dd = f2{c2, c3};
#pragma unroll
            for (int i = 0; i < 12; ++i) p[i] = __builtin_elementwise_fma(cc, p[i], dd);
        } else if (MODE == 2) {  // transposed FIR chain, 24 taps
#pragma unroll
            for (int i = 0; i < 23; ++i) a[i] = __builtin_fmaf((i & 1) ? c0 : c1, x, a[i + 1]);
            a[23] = c2 * x;
            x = a[0] * c3;
        } else if (MODE == 3) {  // register rotation: 23 moves + 1
            float t = a[0];
#pragma unroll
            for (int i = 0; i < 23; ++i) a[i] = a[i + 1];
            a[23] = t;
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
            asm volatile("" : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
            asm volatile("" : "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]));
        } else if (MODE == 4) {  // 24 fma with a wave shuffle every 12
#pragma unroll
            for (int i = 0; i < 24; ++i) a[i] = __builtin_fmaf((i & 1) ? c0 : c1, a[i], c2);
            a[0] += __shfl_up(a[5], 1);
            a[1] += __shfl_up(a[6], 1);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = x;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// gather: lane i walks row i with dwordx4 loads; nfma independent fmas per 4 samples hide it or not
template <int NFMA>
__global__ __launch_bounds__(64) void k_gather(const float *in, float *out, unsigned long long *cyc, int width, int rows_per_wave_stride) {
    const float4 *row = (const float4 *)(in + ((size_t)blockIdx.x * 64 + threadIdx.x) * rows_per_wave_stride);
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = i;
    unsigned long long t0 = __builtin_readcyclecounter();
    float4 nxt = row[0];
    for (int j = 0; j < width / 4 - 1; ++j) {
        float4 cur = nxt;
        nxt = row[j + 1];
#pragma unroll
        for (int r = 0; r < NFMA / 16; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 0.999f, (i & 1) ? cur.x : cur.y);
        }
        a[0] += cur.z + cur.w;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run_valu(const char *name, int n_inst, float *out, unsigned long long *cyc, float *coef) {
    const int iters = 20000;
    for (int w = 1; w <= 4; ++w) {
        int blocks = 256 * 4 * w;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, coef, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, coef, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        double cyc_per_inst_wave = avg / ((double)iters * n_inst);
        printf("%-8s waves/SIMD=%d  cycles/wave-instr=%.2f  => SIMD issue interval %.2f cyc  (kernel %.3f ms, clk~%.2f GHz)\n",
               name, w, cyc_per_inst_wave, cyc_per_inst_wave / w, ms, avg / (ms * 1e6));
    }
}

int main() {
    float *out, *coef; unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 4 * 8 * 64 * 4));
    CK(hipMemset(out, 0, 256 * 4 * 8 * 64 * 4));
    CK(hipMalloc(&cyc, 256 * 4 * 8 * 8));
    float hc[4] = {0.99f, 1.01f, 0.001f, 0.5f};
    CK(hipMalloc(&coef, 16)); CK(hipMemcpy(coef, hc, 16, hipMemcpyHostToDevice));
    run_valu<0>("fma", 24, out, cyc, coef);
    run_valu<1>("pkfma", 12, out, cyc, coef);
    run_valu<2>("chain", 25, out, cyc, coef);
    run_valu<3>("mov", 24, out, cyc, coef);
    run_valu<4>("fma+shfl", 28, out, cyc, coef);

    // gather: 720-wide rows, row stride 1440 floats (same-field rows), NFMA fmas per 4 samples
    const int width = 720, stride = 1440;
    for (int w = 1; w <= 3; ++w) {
        int blocks = 256 * 4 * w;
        size_t n = (size_t)blocks * 64 * stride;
        float *in; CK(hipMalloc(&in, n * 4)); CK(hipMemset(in, 0, n * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto go = [&](auto kern, const char *nm, int nf) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, in, out, cyc, width, stride);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, in, out, cyc, width, stride);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double px = (double)blocks * 64 * width;
            printf("gather nfma=%-4d waves/SIMD=%d  %.3f ms  %.1f Gpx/s read  (%.2f TB/s)  ideal-valu %.3f ms\n", nf, w, ms, px / ms * 1e-6,
                   px * 4 / ms * 1e-9, (double)blocks * (width / 4) * (nf + 2) * 2.0 / (256 * 4) / 2.4e6 );
        };
        go(k_gather<16>, "g16", 16);
        go(k_gather<128>, "g128", 128);
        go(k_gather<512>, "g512", 512);
        CK(hipFree(in));
    }
    return 0;
}

// Unit check (tool only): cm_blk_fir.h's BlkFir - one 20-tap chain, 16-sample blocks, v_mfma_f32_16x16x32_f16 with the data
// operand staged through LDS - against a float64 sum on the host; then the cost of a chain run in isolation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_ab/ubench_blk16_fir tools/ubench_blk16_fir.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../color_modem_amd/csrc/cm_blk_fir.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace cm;

// in / out: [64 lines][n]
__global__ __launch_bounds__(64, 1) void fir_kernel(const float *in, float *out, int n, const BlkTiles *tiles, float g17, float g18, float g19, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned char ops_store[kBlkSlots * kBlkSlotBytes];
    blk_lds_byte *ops = (blk_lds_byte *)ops_store;
    const int lane = threadIdx.x;
    for (int i = lane * 16; i < kBlkSlots * kBlkSlotBytes; i += 64 * 16) *(blk_lds_u4 *)(ops + i) = (blk_u4){0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    const BlkTiles tl = tiles[lane];
    BlkAddr ad;
    ad.init(lane);
    for (int rep = 0; rep < reps; ++rep) {
        BlkFir f;
        f.reset();
        BlkSlots sl;
        sl.cur = 0; sl.hist = kBlkSlotBytes;      // a single chain: ping-pong between two slots
        if (rep) {          // history of a fresh stream: zeros
            *(blk_lds_u4 *)(ops + sl.hist + lane * 64) = (blk_u4){0u, 0u, 0u, 0u};
            *(blk_lds_u4 *)(ops + sl.hist + lane * 64 + 16) = (blk_u4){0u, 0u, 0u, 0u};
            *(blk_lds_u4 *)(ops + sl.hist + lane * 64 + 32) = (blk_u4){0u, 0u, 0u, 0u};
            *(blk_lds_u4 *)(ops + sl.hist + lane * 64 + 48) = (blk_u4){0u, 0u, 0u, 0u};
        }
        const float *ip = in + (size_t)(blockIdx.x * 64 + lane) * n;
        float *op = out + (size_t)(blockIdx.x * 64 + lane) * n;
#pragma nounroll
        for (int tb = 0; tb < n; tb += kBlk) {
            float xs[kBlk], ys[kBlk];
#pragma unroll
            for (int s = 0; s < kBlk; ++s) xs[s] = ip[tb + s];
            f.run(xs, ys, tl, g17, g18, g19, ops, ad, sl);
            const int t = sl.cur; sl.cur = sl.hist; sl.hist = t;
#pragma unroll
            for (int s = 0; s < kBlk; ++s) op[tb + s] = ys[s];
        }
    }
}

int main(int argc, char **argv) {
    const int n = 720 + 16, blocks = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 1;
    // the product's taps are a Kaiser half-band; any symmetric 20-tap set checks the layout
    double c[10];
    for (int i = 0; i < 10; ++i) c[i] = (i % 2 ? -1.0 : 1.0) * 0.63 / (2 * (9 - i) + 1) * (0.4 + 0.06 * i);
    auto tap = [&](int k) { return c[k < 10 ? k : 19 - k]; };
    std::vector<_Float16> t(64 * 16);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int kk = blk_tile_tap(l, j);
            const float v = kk < 0 ? 0.f : (float)tap(kk) * kBlkScale;
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            t[(size_t)l * 16 + j] = hi;
            t[(size_t)l * 16 + 8 + j] = lo;
        }
    const size_t lines = (size_t)blocks * 64;
    std::vector<float> x(lines * n), y(lines * n);
    srand(3);
    for (auto &v : x) v = (float)((rand() / (double)RAND_MAX - 0.5) * 200.0);
    float *dx, *dy;
    void *dt;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dy, y.size() * 4)); CK(hipMalloc(&dt, t.size() * 2));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t.data(), t.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fir_kernel, dim3(blocks), dim3(64), 0, 0, dx, dy, n, (const BlkTiles *)dt, (float)tap(17), (float)tap(18), (float)tap(19), 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ymax = 0;
    int wl = -1, wt = -1;
    for (size_t l = 0; l < lines; ++l)
        for (int s = 0; s < n; ++s) {
            double ref = 0;
            for (int j = 0; j < 20 && j <= s; ++j) ref += (double)(float)tap(j) * x[l * n + s - j];
            const double e = std::fabs(ref - y[l * n + s]);
            if (e > worst) { worst = e; wl = (int)l; wt = s; }
            ymax = std::max(ymax, std::fabs(ref));
        }
    printf("check: max |err| %.3e at line %d sample %d, max |y| %.3e, relative %.3e\n", worst, wl, wt, ymax, worst / ymax);
    if (reps > 1) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fir_kernel, dim3(blocks), dim3(64), 0, 0, dx, dy, n, (const BlkTiles *)dt, (float)tap(17), (float)tap(18), (float)tap(19), reps);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double runs = (double)blocks * reps * (n / kBlk);
        printf("timing: %d workgroups x %d reps: %.3f ms, %.1f ns per chain run of a wave (16 samples x 64 lines)\n", blocks, reps, ms, ms * 1e6 / (reps * (n / kBlk)));
        (void)runs;
    }
    return worst / ymax < 2e-6 ? 0 : 1;
}

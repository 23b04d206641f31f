// Micro-benchmark (tool only): what a FIR update costs in wall time under the board's power cap, by where the tap comes from.
//   0: v_fma_f32 d, tap(VGPR), x, acc  - three VGPR reads, no bank conflict (the form the kernels have)
//   1: v_fmamk_f32 d, x, K, acc        - tap as a 32-bit literal in the instruction stream: two VGPR reads
//   2: v_fmac_f32 acc, K, x            - VOP2, literal in src0, accumulate in place
//   3: v_fma_f32 d, tap(SGPR), x, acc  - tap in an SGPR
//   4: v_pk_fma_f32 (VGPR taps)        - two FMAs per instruction
// 24 independent instructions per iteration, W waves per SIMD on every SIMD of the chip, long enough for the clock to settle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89"
#define FMA(d, a, b, c) "v_fma_f32 v" #d ", v" #a ", v" #b ", v" #c "\n"
#define FMK(d, x, k, c) "v_fmamk_f32 v" #d ", v" #x ", 0x3d" #k ", v" #c "\n"
#define FMC(d, k, x) "v_fmac_f32 v" #d ", 0x3d" #k ", v" #x "\n"
#define FMS(d, s, x, c) "v_fma_f32 v" #d ", s" #s ", v" #x ", v" #c "\n"
#define PK(d, a, b, c) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #c ":" #c "+1]\n"

template <int V>
__global__ __launch_bounds__(64) void k(unsigned long long *cyc, int iters) {
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0)
            asm volatile(FMA(40,0,32,41) FMA(41,1,32,42) FMA(42,2,32,43) FMA(43,3,32,44) FMA(44,4,32,45) FMA(45,5,32,46) FMA(46,6,32,47) FMA(47,7,32,48)
                         FMA(48,8,32,49) FMA(49,9,32,50) FMA(50,10,32,51) FMA(51,11,32,52) FMA(52,12,32,53) FMA(53,13,32,54) FMA(54,14,32,55) FMA(55,15,32,56)
                         FMA(56,16,32,57) FMA(57,17,32,58) FMA(58,18,32,59) FMA(59,19,32,60) FMA(60,20,32,61) FMA(61,21,32,62) FMA(62,22,32,63) FMA(63,23,32,33) ::: CLOB);
        if (V == 1)
            asm volatile(FMK(40,32,123450,41) FMK(41,32,223451,42) FMK(42,32,323452,43) FMK(43,32,423453,44) FMK(44,32,523454,45) FMK(45,32,623455,46) FMK(46,32,723456,47) FMK(47,32,023457,48)
                         FMK(48,32,133450,49) FMK(49,32,233451,50) FMK(50,32,333452,51) FMK(51,32,433453,52) FMK(52,32,533454,53) FMK(53,32,633455,54) FMK(54,32,733456,55) FMK(55,32,033457,56)
                         FMK(56,32,143450,57) FMK(57,32,243451,58) FMK(58,32,343452,59) FMK(59,32,443453,60) FMK(60,32,543454,61) FMK(61,32,643455,62) FMK(62,32,743456,63) FMK(63,32,043457,33) ::: CLOB);
        if (V == 2)
            asm volatile(FMC(40,123450,32) FMC(41,223451,32) FMC(42,323452,32) FMC(43,423453,32) FMC(44,523454,32) FMC(45,623455,32) FMC(46,723456,32) FMC(47,023457,32)
                         FMC(48,133450,32) FMC(49,233451,32) FMC(50,333452,32) FMC(51,433453,32) FMC(52,533454,32) FMC(53,633455,32) FMC(54,733456,32) FMC(55,033457,32)
                         FMC(56,143450,32) FMC(57,243451,32) FMC(58,343452,32) FMC(59,443453,32) FMC(60,543454,32) FMC(61,643455,32) FMC(62,743456,32) FMC(63,043457,32) ::: CLOB);
        if (V == 3)
            asm volatile(FMS(40,50,32,41) FMS(41,51,32,42) FMS(42,52,32,43) FMS(43,53,32,44) FMS(44,54,32,45) FMS(45,55,32,46) FMS(46,56,32,47) FMS(47,57,32,48)
                         FMS(48,58,32,49) FMS(49,59,32,50) FMS(50,60,32,51) FMS(51,61,32,52) FMS(52,62,32,53) FMS(53,63,32,54) FMS(54,64,32,55) FMS(55,65,32,56)
                         FMS(56,66,32,57) FMS(57,67,32,58) FMS(58,68,32,59) FMS(59,69,32,60) FMS(60,70,32,61) FMS(61,71,32,62) FMS(62,72,32,63) FMS(63,73,32,33)
                         ::: CLOB, "s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73");
        if (V == 4)   // 12 packed = 24 FMAs: d pair = tap pair * x pair + acc pair
            asm volatile(PK(40,0,32,42) PK(42,2,32,44) PK(44,4,32,46) PK(46,6,32,48) PK(48,8,32,50) PK(50,10,32,52)
                         PK(52,12,32,54) PK(54,14,32,56) PK(56,16,32,58) PK(58,18,32,60) PK(60,20,32,62) PK(62,22,32,64) ::: CLOB);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> void run(int w, unsigned long long *d, int iters, const char *name) {
    const int blocks = 256 * 4 * w;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);     // settle the clock
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(blocks);
    CK(hipMemcpy(c.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : c) avg += v; avg /= blocks;
    const double fmas = 24.0 * iters;
    printf("%-28s waves/SIMD=%d  counter ticks/FMA/SIMD %.3f  kernel %.2f ms  ns/FMA/SIMD %.4f  TFLOP/s %.1f\n", name, w, avg / fmas / w, ms,
           ms * 1e6 / (fmas * w), 2.0 * 64 * fmas * blocks / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned long long *d; CK(hipMalloc(&d, 8 * 256 * 4 * 8));
    const int it = 400000;
    for (int rep = 0; rep < 2; ++rep)
        for (int w : {2, 3}) {
            run<0>(w, d, it, "v_fma_f32 vgpr tap");
            run<1>(w, d, it, "v_fmamk_f32 literal tap");
            run<2>(w, d, it, "v_fmac_f32 literal tap");
            run<3>(w, d, it, "v_fma_f32 sgpr tap");
            run<4>(w, d, it, "v_pk_fma_f32 vgpr taps");
        }
    return 0;
}

// Micro-benchmark (tool only): does v_fma_f32 with three VGPR sources slow down when the sources share a register bank
// (index mod 4)?  24 independent FMAs per iteration on fixed registers; W waves per SIMD.
//   variant 0: sources in three different banks     1: two in one bank     2: all three in one bank
//   variant 3: one VGPR source + two literal-free SGPR-less forms (v_fma_f32 v, v, v, v with src1 == src2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define FMA(d, a, b, c) "v_fma_f32 v" #d ", v" #a ", v" #b ", v" #c "\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

template <int V>
__global__ __launch_bounds__(64) void k(unsigned long long *cyc, int iters) {
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0)   // banks 0, 1, 2 -> dst anywhere
            asm volatile(FMA(40,0,1,2) FMA(41,4,5,6) FMA(42,8,9,10) FMA(43,12,13,14) FMA(44,16,17,18) FMA(45,20,21,22) FMA(46,24,25,26) FMA(47,28,29,30)
                         FMA(48,0,5,10) FMA(49,4,9,14) FMA(50,8,13,18) FMA(51,12,17,22) FMA(52,16,21,26) FMA(53,20,25,30) FMA(54,24,29,2) FMA(55,28,1,6)
                         FMA(56,0,9,18) FMA(57,4,13,22) FMA(58,8,17,26) FMA(59,12,21,30) FMA(60,16,25,2) FMA(61,20,29,6) FMA(62,24,1,10) FMA(63,28,5,14) ::: CLOB);
        if (V == 1)   // src0, src1 in bank 0; src2 in bank 2
            asm volatile(FMA(40,0,4,2) FMA(41,4,8,6) FMA(42,8,12,10) FMA(43,12,16,14) FMA(44,16,20,18) FMA(45,20,24,22) FMA(46,24,28,26) FMA(47,28,0,30)
                         FMA(48,0,8,10) FMA(49,4,12,14) FMA(50,8,16,18) FMA(51,12,20,22) FMA(52,16,24,26) FMA(53,20,28,30) FMA(54,24,0,2) FMA(55,28,4,6)
                         FMA(56,0,12,18) FMA(57,4,16,22) FMA(58,8,20,26) FMA(59,12,24,30) FMA(60,16,28,2) FMA(61,20,0,6) FMA(62,24,4,10) FMA(63,28,8,14) ::: CLOB);
        if (V == 2)   // all three in bank 0
            asm volatile(FMA(40,0,4,8) FMA(41,4,8,12) FMA(42,8,12,16) FMA(43,12,16,20) FMA(44,16,20,24) FMA(45,20,24,28) FMA(46,24,28,0) FMA(47,28,0,4)
                         FMA(48,0,8,16) FMA(49,4,12,20) FMA(50,8,16,24) FMA(51,12,20,28) FMA(52,16,24,0) FMA(53,20,28,4) FMA(54,24,0,8) FMA(55,28,4,12)
                         FMA(56,0,12,24) FMA(57,4,16,28) FMA(58,8,20,0) FMA(59,12,24,4) FMA(60,16,28,8) FMA(61,20,0,12) FMA(62,24,4,16) FMA(63,28,8,20) ::: CLOB);
        if (V == 3)   // the transposed FIR update as the kernels have it: dst = acc chain, tap and x elsewhere: d = c * x + acc
            asm volatile(FMA(40,0,32,41) FMA(41,1,32,42) FMA(42,2,32,43) FMA(43,3,32,44) FMA(44,4,32,45) FMA(45,5,32,46) FMA(46,6,32,47) FMA(47,7,32,48)
                         FMA(48,8,32,49) FMA(49,9,32,50) FMA(50,10,32,51) FMA(51,11,32,52) FMA(52,12,32,53) FMA(53,13,32,54) FMA(54,14,32,55) FMA(55,15,32,56)
                         FMA(56,16,32,57) FMA(57,17,32,58) FMA(58,18,32,59) FMA(59,19,32,60) FMA(60,20,32,61) FMA(61,21,32,62) FMA(62,22,32,63) FMA(63,23,32,33) ::: CLOB);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> void run(int w, unsigned long long *d, int iters) {
    const int blocks = 256 * 4 * w;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(blocks);
    CK(hipMemcpy(c.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : c) avg += v; avg /= blocks;
    printf("variant %d  waves/SIMD=%d  cycles/wave-instr=%.2f  SIMD issue interval %.2f  kernel %.3f ms\n", V, w, avg / (24.0 * iters), avg / (24.0 * iters) / w, ms);
}
int main() {
    unsigned long long *d; CK(hipMalloc(&d, 8 * 256 * 4 * 8));
    for (int w : {1, 2, 3, 4}) { run<0>(w, d, 20000); run<1>(w, d, 20000); run<2>(w, d, 20000); run<3>(w, d, 20000); }
    return 0;
}

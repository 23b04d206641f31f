// Micro-benchmark (tool only): v_pk_fma_f32 issue interval against the placement of its three 64-bit sources.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define S2(x) #x
#define PK(d, a, b, c) "v_pk_fma_f32 v[" S2(d) ":" S2(d) "+1], v[" S2(a) ":" S2(a) "+1], v[" S2(b) ":" S2(b) "+1], v[" S2(c) ":" S2(c) "+1]\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
template <int V>
__global__ __launch_bounds__(64) void k(unsigned long long *cyc, int iters) {
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0)   // low registers in banks 0, 2, 0
            asm volatile(PK(40,0,2,4) PK(42,8,10,12) PK(44,16,18,20) PK(46,24,26,28) PK(48,4,6,8) PK(50,12,14,16) PK(52,20,22,24) PK(54,28,30,0)
                         PK(56,0,2,4) PK(58,8,10,12) PK(60,16,18,20) PK(62,24,26,28) ::: CLOB);
        if (V == 1)   // all low registers in bank 0
            asm volatile(PK(40,0,4,8) PK(42,4,8,12) PK(44,8,12,16) PK(46,12,16,20) PK(48,16,20,24) PK(50,20,24,28) PK(52,24,28,0) PK(54,28,0,4)
                         PK(56,0,8,16) PK(58,4,12,20) PK(60,8,16,24) PK(62,12,20,28) ::: CLOB);
        if (V == 2)   // the packed section update of the kernels: coefficient pair common to consecutive instructions
            asm volatile(PK(40,0,2,6) PK(42,0,8,14) PK(44,0,16,22) PK(46,0,24,30) PK(48,4,2,6) PK(50,4,10,14) PK(52,4,18,22) PK(54,4,26,30)
                         PK(56,0,6,10) PK(58,0,14,18) PK(60,4,22,26) PK(62,4,30,2) ::: CLOB);
        if (V == 3)   // src0 == src1 (two distinct sources)
            asm volatile(PK(40,0,0,2) PK(42,4,4,6) PK(44,8,8,10) PK(46,12,12,14) PK(48,16,16,18) PK(50,20,20,22) PK(52,24,24,26) PK(54,28,28,30)
                         PK(56,0,0,6) PK(58,4,4,10) PK(60,8,8,14) PK(62,12,12,18) ::: CLOB);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(int w, unsigned long long *d, int iters) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(blocks);
    CK(hipMemcpy(c.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : c) avg += v; avg /= blocks;
    printf("pk variant %d  waves/SIMD=%d  cycles/wave-instr=%.2f  SIMD issue interval %.2f\n", V, w, avg / (12.0 * iters), avg / (12.0 * iters) / w);
}
int main() {
    unsigned long long *d; CK(hipMalloc(&d, 8 * 256 * 4 * 8));
    for (int w : {1, 2, 3, 4}) { run<0>(w, d, 20000); run<1>(w, d, 20000); run<2>(w, d, 20000); run<3>(w, d, 20000); }
    return 0;
}

// Reproducer for profiles/r03_scan_notes.txt item 9 (tool, not product code): "the interpolators' edge sample summed two taps per instruction - garbage on
// the device (scalar-addressed LDS operands in v_pk_fma_f32)".  What that code did: read a PAIR of floats from LDS as one 8-byte value at a float index
// whose parity depends on the row width (X[W - 1 + 10 - q], W even -> odd index) - a ds_read_b64 at an address that is not a multiple of 8.
// This kernel reads pairs at even and at odd float indices, as one 8-byte access and as two 4-byte ones, and the host compares.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/repro_lds_pair tools/repro_lds_pair.hip && tools/_bin/repro_lds_pair
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f2 lds_f2;
__global__ void k(float *out, int base) {
    __shared__ __attribute__((aligned(16))) float row[512];
    for (int i = threadIdx.x; i < 512; i += 64) row[i] = (float)i;
    __syncthreads();
    const int at = base + 2 * (int)threadIdx.x;                 // wave-uniform parity
    const lds_float *p = (const lds_float *)row + at;
    const f2 pair = *(const lds_f2 *)p;                         // what the packed form did (the compiler is told nothing about alignment but the type's)
    out[threadIdx.x * 4 + 0] = pair.x;
    out[threadIdx.x * 4 + 1] = pair.y;
    out[threadIdx.x * 4 + 2] = p[0];                            // the one-tap form
    out[threadIdx.x * 4 + 3] = p[1];
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof h);
    for (int base = 8; base <= 9; ++base) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, base);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t) bad += h[4 * t] != h[4 * t + 2] || h[4 * t + 1] != h[4 * t + 3];
        printf("pairs at %s float indices: %d of 64 lanes differ between the 8-byte read and two 4-byte reads (lane 0: pair %.0f %.0f, scalars %.0f %.0f)\n",
               base & 1 ? "ODD " : "EVEN", bad, h[0], h[1], h[2], h[3]);
    }
    return 0;
}

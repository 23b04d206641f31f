// ubench_store.hip - HBM write throughput of the demodulators' store pattern: every wave owns 64 rows x 3 planes of a
// [F][3][H][W] float array and walks along them, writing SEG bytes per row and visit (rows of one wave are 2 lines apart).
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_store tools/ubench_store.hip ; run: ./ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 720, H = 576, F = 1000;

template <int SEG, bool NT>   // SEG bytes per row per visit: 64, 128, 256
__global__ __launch_bounds__(64) void store_kernel(float *out, int n_waves) {
    const int lane = threadIdx.x;
    const long long wave = blockIdx.x;
    constexpr int CH = SEG / 16;            // 16-byte chunks per row segment
    constexpr int ROWS = 64 / CH;           // rows per wave-instruction
    // wave -> 64 calls of the flattened [frame][field][row] list
    const long long c0 = wave * 64;
    for (int col = 0; col < W; col += SEG / 4) {
        for (int q = 0; q < CH; ++q) {
            const int r = lane / CH + ROWS * q;
            long long c = c0 + r;
            long long frame = c / H;
            int rem = (int)(c - frame * H);
            int line = rem < H / 2 ? 2 * rem : 2 * (rem - H / 2) + 1;
            int cc = col + 4 * (lane % CH);
            if (frame < F && cc < W) {
                for (int p = 0; p < 3; ++p) {
                    f4 *dst = (f4 *)(out + ((frame * 3 + p) * H + line) * (long long)W + cc);
                    f4 v = {1.f * lane, 2.f, 3.f, (float)col};
                    if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
            }
        }
    }
}

template <int SEG, bool NT>
void run(float *buf, const char *name) {
    const long long calls = (long long)F * H;
    const int waves = (int)((calls + 63) / 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((store_kernel<SEG, NT>), dim3(waves), dim3(64), 0, 0, buf, waves);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((store_kernel<SEG, NT>), dim3(waves), dim3(64), 0, 0, buf, waves);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double bytes = (double)F * 3 * H * W * 4;
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}

int main() {
    float *buf;
    size_t n = (size_t)F * 3 * H * W;
    if (hipMalloc(&buf, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<64, true>(buf, "64 B per row visit, nt");
    run<64, false>(buf, "64 B per row visit");
    run<128, true>(buf, "128 B per row visit, nt");
    run<128, false>(buf, "128 B per row visit");
    run<256, true>(buf, "256 B per row visit, nt");
    run<256, false>(buf, "256 B per row visit");
    return 0;
}

// Unit check (tool only): v_cvt_pk_u8_f32 against uint8(rint(255 * clip(x, 0, 1))) (image.py:7-8) on a sweep of floats:
// every half-integer tie, values around 0 and 255, negatives, large values, denormals.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__global__ void k(const float *in, unsigned char *a, unsigned char *b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    a[i] = (unsigned char)__builtin_rintf(255.f * __builtin_fminf(__builtin_fmaxf(x, 0.f), 1.f));
    b[i] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(255.f * x, 0, 0);
}
int main() {
    std::vector<float> x;
    for (int i = -600; i <= 600; ++i)                 // ties and their neighbours in units of 1/255
        for (int d = -2; d <= 2; ++d) x.push_back(std::nextafterf((i + 0.5f) / 255.f, d < 0 ? -10.f : 10.f) + d * 1e-9f * (d != 0));
    for (int i = 0; i < 2000000; ++i) x.push_back(-0.5f + 2.0f * (float)i / 2000000.f);
    const float extra[] = {0.f, -0.f, 1.f, 1.0000001f, 0.99999994f, 1e-40f, -1e-40f, 1e30f, -1e30f, 3.4e38f, 0.5f / 255.f, 254.5f / 255.f, 255.5f / 255.f};
    for (float v : extra) x.push_back(v);
    const int n = (int)x.size();
    float *dx; unsigned char *da, *db;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&da, n)); CK(hipMalloc(&db, n));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, n);
    CK(hipDeviceSynchronize());
    std::vector<unsigned char> a(n), b(n);
    CK(hipMemcpy(a.data(), da, n, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i]) { if (bad < 10) printf("x = %.9g: rint/clip %d, v_cvt_pk_u8_f32 %d\n", x[i], a[i], b[i]); ++bad; }
    printf("%d values, %d differences\n", n, bad);
    return bad ? 1 : 0;
}

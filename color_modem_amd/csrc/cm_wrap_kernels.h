// cm_wrap_kernels.h - SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem (ref comb.py:96-113 on top of
// pal.py:79-234).  These stacks mix the two front ends on the second line of every run (call 0 is the plain decode, call 1
// averages it with the first delay-line decode) or reach back three lines, so they do not fit the per-line coefficient
// tables of the fused decoders; they run as a composition on the device instead: the inner decoder's kernel in component
// mode (strip_chroma = False, exactly what the wrapper asks its backend for), the two element-wise kernels below around
// the inner modulator's kernel (the wrapper strips the luma by re-modulating the averaged chroma, comb.py:105-107).
// Memory-bound glue: one thread per 4 samples, 16-byte accesses.
#ifndef CM_WRAP_KERNELS_H
#define CM_WRAP_KERNELS_H

#include "cm_kernels.h"

namespace cm {

struct CombWrapArgs {
    const float *inner;     // [n][3][Wp]  (y, u, v) of the backend's calls
    float *uv;              // [n][3][Wp]  (0, u, v) for the re-modulation
    float *ysrc;            // [n][Wp]
    const float *remod;     // [n][Wp]
    float *rgb;             // [n][3][Wp]
    int n, Wp, k0, own_delay, minavg;
    float m[9];
};

__device__ __forceinline__ f4 minavg4(f4 a, f4 b) {
    return f4{minavg_(a.x, b.x), minavg_(a.y, b.y), minavg_(a.z, b.z), minavg_(a.w, b.w)};
}

// comb.py:96-104: (u, v) = avg(last, curr), luma source = the previous call's luma (delay) or this call's
__global__ __launch_bounds__(256) void comb_combine_kernel(const CombWrapArgs a) {
    const int quads = a.Wp >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.n * quads) return;
    const int i = (int)(idx / quads), q = (int)(idx - (long long)i * quads);
    const f4 *cur = (const f4 *)(a.inner + (long long)i * 3 * a.Wp) + q;
    const bool first = a.k0 + i == 0 || i == 0;          // call 0 of a run: (y, u, v) = curr; a run submitted without history likewise
    const f4 *last = first ? cur : (const f4 *)(a.inner + (long long)(i - 1) * 3 * a.Wp) + q;
    const f4 cy = cur[0], cu = cur[quads], cv = cur[2 * quads];
    const f4 ly = last[0], lu = last[quads], lv = last[2 * quads];
    f4 u, v, y;
    if (first) { u = cu; v = cv; y = cy; }
    else {
        u = a.minavg ? minavg4(lu, cu) : 0.5f * (lu + cu);
        v = a.minavg ? minavg4(lv, cv) : 0.5f * (lv + cv);
        y = a.own_delay ? ly : cy;
    }
    f4 *o = (f4 *)(a.uv + (long long)i * 3 * a.Wp) + q;
    o[0] = f4{0.f, 0.f, 0.f, 0.f};
    o[quads] = u;
    o[2 * quads] = v;
    ((f4 *)(a.ysrc + (long long)i * a.Wp))[q] = y;
}

// comb.py:105-107 + decode_components: luma = source - re-modulated chroma (not on the first call of a run), colour matrix
__global__ __launch_bounds__(256) void comb_finish_kernel(const CombWrapArgs a) {
    const int quads = a.Wp >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.n * quads) return;
    const int i = (int)(idx / quads), q = (int)(idx - (long long)i * quads);
    const f4 *uv = (const f4 *)(a.uv + (long long)i * 3 * a.Wp) + q;
    const f4 u = uv[quads], v = uv[2 * quads];
    f4 y = ((const f4 *)(a.ysrc + (long long)i * a.Wp))[q];
    if (a.k0 + i > 0) y = y - ((const f4 *)(a.remod + (long long)i * a.Wp))[q];
    f4 *o = (f4 *)(a.rgb + (long long)i * 3 * a.Wp) + q;
    o[0] = a.m[0] * y + a.m[1] * u + a.m[2] * v;
    o[quads] = a.m[3] * y + a.m[4] * u + a.m[5] * v;
    o[2 * quads] = a.m[6] * y + a.m[7] * u + a.m[8] * v;
}

}  // namespace cm
#endif

// cm_am_kernels.h - device-side lane drivers of the amplitude-modulated line-sequential standards (gfx950):
// Proto-SECAM 1957 and NIIR / SECAM-IV (cm_am_stages.h).
//
// Same execution model as the other kernels: one lane owns one call (= one scan line) and walks it one sample per step,
// a 64-lane wavefront walks 64 consecutive calls of the flattened [frame][run][call] list (cm_kernels.h: locate_call, one
// halo lane for the previous call's colour-difference signal / phase reference).  These are the plain kernels of the
// uncommon standards: rows are read with one 16-byte load per lane, plane and 4 steps straight
// from global memory, results leave through the usual LDS tile as 64-byte row segments; streams that have to wait for a
// longer path (the decoder's luma, the encoder's chroma) sit in a lane-private LDS delay ring; the 61 resampling taps are
// pinned in VGPRs (cm_stages.h: VPolicy).
#ifndef CM_AM_KERNELS_H
#define CM_AM_KERNELS_H

#include "cm_am_stages.h"
#include "cm_mod_kernels.h"

namespace cm {

constexpr int kAmRing = 32;                    // slots of the delay rings: delays up to 31 samples (checked by the host)
constexpr int kAmRingFloats = kAmRing * 64;

struct AmGeom {
    AmLine line;
    const float *carrier;      // {cos, sin}(n * carrier_phase_step), n < W
    int frame_base;            // first_frame mod (2 * frame_cycle): parity and phase cycle of the batch's first frame
};

template <typename T>
__device__ __forceinline__ void pin_taps3(Taps3<T> &t) {
#pragma unroll
    for (int i = 0; i < kAmTaps; ++i) pin_vgpr(t.h[i]);
}

struct ProtoDemodArgs {
    Geom g;
    AmGeom a;
    ProtoDemodK<float> k;
};

__global__ __launch_bounds__(64, 2) void proto_demod_kernel(const ProtoDemodArgs args) {
    constexpr int kTile = 16, DEPTH = 1;
    __shared__ __attribute__((aligned(16))) float lds_store[3 * 64 * kTile + kAmRingFloats];
    lds_float *otile_base = (lds_float *)lds_store;
    lds_float *ring = otile_base + 3 * 64 * kTile;
    const Geom &g = args.g;
    ProtoDemodK<float> k = args.k;
    pin_taps3(k.taps);
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
    const float *op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    const long long frame = (long long)args.a.frame_base + lc.frame;
    const bool alt = args.a.line.alternate(frame, lc.line);
    const float w_prev = lc.kk > 0 ? 1.f : 0.f;          // protosecam.py:93-94: the first line of a run has no previous chroma
    const int idx1 = ((lane + 63) & 63) * 4;
    ProtoDemod<float> st;
    st.reset();
    lds_float *otile = otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int lat_c = ProtoDemod<float>::lat_chroma(k), lat_y = ProtoDemod<float>::lat_luma(k);
    const int dly = lat_c - lat_y;                       // luma waits for the chroma path (0 .. kAmRing - 1)
    const int T = (g.Wp + lat_c + 3) & ~3;
    for (int j = 0; j < kAmRing; ++j) ring[j * 64 + lane] = 0.f;
    f4 xv = load_luma<false>(xp, 0, true, W);
    for (int tb = 0; tb < T; tb += 4) {
        const f4 xn = load_luma<false>(xp, tb + 4, true, W);      // next body's samples: a body hides the latency
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            float luma, chroma;
            st.step(k, t, xv[s], luma, chroma);
            ring[(t & (kAmRing - 1)) * 64 + lane] = luma;
            const float luma_d = ring[((t - dly) & (kAmRing - 1)) * 64 + lane];
            const float prev = lane_from(idx1, chroma) * w_prev;
            const float dr = alt ? prev : chroma, db = alt ? chroma : prev;             // protosecam.py:105-108
            Rgb<float> o;
            o.r = fmaf_(k.m[0][0], luma_d, fmaf_(k.m[0][1], dr, k.m[0][2] * db));
            o.g = fmaf_(k.m[1][0], luma_d, fmaf_(k.m[1][1], dr, k.m[1][2] * db));
            o.b = fmaf_(k.m[2][0], luma_d, fmaf_(k.m[2][1], dr, k.m[2][2] * db));
            const int n = t - lat_c;
            if (n >= 0 && n < W) put_rgb<false, kTile>(otile, wpos, n, o);
            if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
        }
        xv = xn;
    }
}

struct ProtoModArgs {
    Geom g;
    AmGeom a;
    ProtoModK<float> k;
    int averaging;
};

// DEPTH = 1: the encoder sits inside ColorAveragingModem (comb.py:141-152): a call modulates line - 2 with the previous
// call's luma and the mean of both calls' colour-difference signals (previous call = neighbouring lane)
template <int DEPTH>
__global__ __launch_bounds__(64, 2) void proto_mod_kernel(const ProtoModArgs args) {
    constexpr int kTile = 16;
    __shared__ __attribute__((aligned(16))) float lds_store[64 * kTile + kAmRingFloats];
    lds_float *otile_base = (lds_float *)lds_store;
    lds_float *ring = otile_base + 64 * kTile;
    const Geom &g = args.g;
    ProtoModK<float> k = args.k;
    pin_taps3(k.taps);
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *rp, *op;
    mod_rows<false>(g, lc, rp, op);
    const long long frame = (long long)args.a.frame_base + lc.frame;
    const int line = DEPTH ? lc.line - 2 : lc.line;       // the line that is modulated
    const bool alt = args.a.line.alternate(frame, line);
    float cph, sph;
    {
        const double phi = args.a.line.start_phase(frame, line);
        cph = (float)cos(phi);
        sph = (float)sin(phi);
    }
    const bool have_prev = lc.kk > 0;
    const int idx1 = ((lane + 63) & 63) * 4;
    ProtoMod<float> st;
    st.reset();
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int lat_y = ProtoMod<float>::lat_luma(k), lat_c = ProtoMod<float>::lat_chroma(k);
    const int lat = lat_y > lat_c ? lat_y : lat_c;
    const int d_c = lat - lat_c, d_y = lat - lat_y;      // one of them is 0: the shorter path's input waits in the ring
    const int dly = d_c > d_y ? d_c : d_y;
    const int T = (g.Wp + lat + 3) & ~3;
    for (int j = 0; j < kAmRing; ++j) ring[j * 64 + lane] = 0.f;
    f4 cur[3], nxt[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) nxt[p] = load_luma<false>(rp + p * g.in_plane_stride, 0, true, W);
    for (int tb = 0; tb < T; tb += 4) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            cur[p] = nxt[p];
            nxt[p] = load_luma<false>(rp + p * g.in_plane_stride, tb + 4, true, W);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            const float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
            float y = fmaf_(k.e[0][0], r, fmaf_(k.e[0][1], gg, k.e[0][2] * b));
            float dr = fmaf_(k.e[1][0], r, fmaf_(k.e[1][1], gg, k.e[1][2] * b));
            float db = fmaf_(k.e[2][0], r, fmaf_(k.e[2][1], gg, k.e[2][2] * b));
            if (DEPTH) {
                const float yp = lane_from(idx1, y), drp = lane_from(idx1, dr), dbp = lane_from(idx1, db);
                if (have_prev) {
                    y = yp;                                  // comb.py:147
                    dr = 0.5f * (dr + drp);                  // comb.py:148-149
                    db = 0.5f * (db + dbp);
                }
            }
            const float d = alt ? db : dr;                   // protosecam.py:75-78
            // the shorter path's input waits dly samples
            const float late = d_c > 0 ? d : y;
            ring[(t & (kAmRing - 1)) * 64 + lane] = late;
            const float waited = ring[((t - dly) & (kAmRing - 1)) * 64 + lane];
            const float d_in = d_c > 0 ? waited : d, y_in = d_c > 0 ? y : (d_y > 0 ? waited : y);
            float luma, chroma;
            st.step(k, t - d_c, d_in, t - d_y, y_in, luma, chroma);
            const int n = t - lat;
            int nc = n < 0 ? 0 : (n > W - 1 ? W - 1 : n);
            const f2 cs = ((const_f2 *)args.a.carrier)[nc];
            const float cosp = fmaf_(cph, cs.x, -(sph * cs.y));      // cos(phi + n step)
            put_composite<false, kTile>(g, otile_base, op, lane, wpos, n, fmaf_(cosp, chroma, luma));
        }
    }
}

}  // namespace cm
#endif

// cm_mod_kernels.h - device-side lane driver of the QAM modulators (gfx950).
//
// Same execution model as cm_kernels.h: one lane owns one call (= one output scan line), a
// wavefront walks 64 consecutive calls in lock-step.  The modulator is light (about 45 vector
// instructions per pixel), so its rows are read with one dwordx4 per plane, lane and 4 steps
// straight from global memory; the composite row leaves through the same LDS tile / 64-byte
// row-segment stores as the demodulators' outputs.
#ifndef CM_MOD_KERNELS_H
#define CM_MOD_KERNELS_H

#include "cm_kernels.h"

namespace cm {

template <int NP, int SP>
struct ModSys {
    static constexpr int kNP = NP, kSP = SP;
};

template <int NP>
struct ModArgs {
    Geom g;                            // g.lanes is reinterpreted as ModLaneK<float> table
    ModK<float, NP> k;
};

// Single-plane variant of flush_tile (composite output).
template <int kTile>
__device__ __forceinline__ void flush_tile1(const Geom &g, const lds_float *otile, const float *op, int first_col, int lane) {
    __builtin_amdgcn_wave_barrier();
    constexpr int kChunks = kTile / 4;
    constexpr int kRows = 64 / kChunks;
    const int chunk = lane & (kChunks - 1);
    const int col = first_col + 4 * chunk;
#pragma nounroll
    for (int q = 0; q < kChunks; ++q) {
        const int row = lane / kChunks + kRows * q;
        typedef __attribute__((address_space(1))) f4 global_f4;
        global_f4 *dst = (global_f4 *)(unsigned long long)ptr_from(row * 4, op);
        const int quad = chunk ^ ((row >> 1) & (kChunks - 1));
        if (dst != nullptr && col < g.W) {
            f4 v = *(const lds_f4 *)(otile + row * kTile + 4 * quad);
            __builtin_nontemporal_store(v, &dst[col >> 2]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Three-plane input tile of the modulators: [3][64 rows][32 samples], filled like the demodulators' tile
// (8 rows x 128 B per global_load_lds_dwordx4), single-buffered: refilled right after its last sample has
// been read, waited for before the first sample of the next tile is read.
constexpr int kLdsIn3 = 3 * kLdsIn;
__device__ __forceinline__ void fill_tile3(const Geom &g, lds_float *itile, const float *rp, int c, int lane) {
    int col = kInTile * c + 4 * (lane & 7);
    if (col > g.W - 4) col = g.W - 4;
#pragma nounroll
    for (int q = 0; q < 8; ++q) {
        const float *src = ptr_from((8 * q + (lane >> 3)) * 4, rp) + col;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * g.in_plane_stride),
                                             (__attribute__((address_space(3))) void *)(itile + p * kLdsIn + q * 256), 16, 0,
                                             CM_FILL_AUX);
    }
}
// r, g, b of samples first .. first + 3 of this lane's row out of the tile; zero beyond the row
__device__ __forceinline__ void read_tile3(const lds_float *itile, int lane, int first, int W, f4 out[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        f4 v = *(const lds_f4 *)(itile + p * kLdsIn + lane * kInTile + (first & (kInTile - 1)));
        if (first >= W) v = f4{0.f, 0.f, 0.f, 0.f};   // W % 4 == 0: a quad is inside or outside as a whole
        out[p] = v;
    }
}
// Advance the input stream by one body: returns samples nxt .. nxt + 3 and keeps the tile protocol.
__device__ __forceinline__ void next_tile3(const Geom &g, lds_float *itile, const float *rp, int lane, int nxt, f4 out[3]) {
    if ((nxt & (kInTile - 1)) == 0 && nxt < g.W) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    read_tile3(itile, lane, nxt, g.W, out);
    if ((nxt & (kInTile - 1)) == kInTile - 4 && nxt + 4 < g.W) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        fill_tile3(g, itile, rp, (nxt >> 5) + 1, lane);
    }
}

// DEPTH = 1: encoder-side line averaging (ColorAveragingModem) needs the previous call's components.
template <int NP, int SP, int DEPTH>
__global__ __launch_bounds__(64, 2) void qam_mod_kernel(const ModArgs<NP> args) {
    constexpr int kTile = 16;
    __shared__ __attribute__((aligned(16))) float lds_store[kLdsIn3 + 64 * kTile];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + kLdsIn3;
    const Geom &g = args.g;
    const ModK<float, NP> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * row_stride;
    const float *op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    ModLaneK<float> lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const ModLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
        float rc, rs;
        if (frame_turn(g, lc.frame, rc, rs)) {
            turn(lk.cph, lk.sph, rc, rs);
            turn(lk.vcph, lk.vsph, rc, rs);
        }
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    QamModCore<float, NP> core;
    core.reset();
    float yw[SP + 4];
#pragma unroll
    for (int j = 0; j < SP + 4; ++j) yw[j] = 0.f;
    lds_float *otile = otile_base + lane * kTile;
    const int wpos = ((lane >> 1) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int T = (W + SP + 3) & ~3;
    fill_tile3(g, itile, rp, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f4 cur[3], nxt[3];
    read_tile3(itile, lane, 0, W, nxt);
    for (int tb = 0; tb < T; tb += 4) {
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        next_tile3(g, itile, rp, lane, tb + 4, nxt);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
            float y = fmaf_(k.e[0][0], r, fmaf_(k.e[0][1], gg, k.e[0][2] * b));
            float u = fmaf_(k.e[1][0], r, fmaf_(k.e[1][1], gg, k.e[1][2] * b));
            float v = fmaf_(k.e[2][0], r, fmaf_(k.e[2][1], gg, k.e[2][2] * b));
            if (DEPTH >= 1) {
                float yp = lane_from(idx1, y), up = lane_from(idx1, u), vp = lane_from(idx1, v);
                y = fmaf_(lk.wy0, y, lk.wy1 * yp);
                u = fmaf_(lk.wc0, u, lk.wc1 * up);
                v = fmaf_(lk.wc0, v, lk.wc1 * vp);
            }
            yw[SP + s] = y;
            const int n7 = t - SP;
            int nc = n7 < 0 ? 0 : (n7 > W - 1 ? W - 1 : n7);
            f2 cc = ((const_f2 *)g.carrier2)[nc];
            float car[2] = {cc.x, cc.y};
            float comp = core.step(k, lk, t, yw[s], u, v, car);
            if (n7 >= 0 && n7 < W) otile[wpos ^ (n7 & (kTile - 1))] = comp;
            if (n7 >= 0 && ((n7 & (kTile - 1)) == kTile - 1 || n7 == W - 1))
                flush_tile1<kTile>(g, otile_base, op, n7 & ~(kTile - 1), lane);
        }
#pragma unroll
        for (int j = 0; j < SP; ++j) yw[j] = yw[j + 4];
    }
}

}  // namespace cm
#endif

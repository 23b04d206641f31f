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
        const int quad = chunk ^ ((row >> CM_TILE_SWZ) & (kChunks - 1));
        if (dst != nullptr && col < g.Wp) {
            f4 v = *(const lds_f4 *)(otile + row * kTile + 4 * quad);
            __builtin_nontemporal_store(v, &dst[col >> 2]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Three-plane float input tile of the modulators: kModBufs buffers of [3][64 rows][kModIT samples], filled like the
// demodulators' tile (global_load_lds_dwordx4: kModIT / 4 lanes per row).  One buffer: refilled right after its last
// sample has been read, waited for before the first sample of the next tile is read (one body of 4 steps to land).  Two
// buffers: tile c + 1 is asked for when tile c is first read (a whole tile of steps to land).
// Measured (profiles/r02_mod_tile_ab.txt, ms per 1000 frames PAL-S / NTSC / SECAM): 32-sample tiles (32 KiB with the output
// tile, 5 workgroups per CU) 2.02 / 1.70 / 2.61; 16-sample tiles (20 KiB, 8 per CU) 1.62 / 1.38 / 2.15; 8-sample tiles
// (32-byte row segments) 2.3 - 2.8 whether double-buffered or not.
#ifndef CM_MOD_IN_TILE
#define CM_MOD_IN_TILE 16
#endif
#ifndef CM_MOD_IN_BUFS
#define CM_MOD_IN_BUFS 1
#endif
constexpr int kModIT = CM_MOD_IN_TILE, kModBufs = CM_MOD_IN_BUFS;
constexpr int kModPlane = 64 * kModIT;                 // floats per plane of one buffer
constexpr int kLdsIn3 = kModBufs * 3 * kModPlane;
__device__ __forceinline__ void fill_tile3(const Geom &g, lds_float *itile, const float *rp, int c, int lane) {
    constexpr int kLanesPerRow = kModIT / 4, kRowsPerInstr = 64 / kLanesPerRow;
    lds_float *buf = itile + (kModBufs == 2 ? (c & 1) * 3 * kModPlane : 0);
    int col = kModIT * c + 4 * (lane & (kLanesPerRow - 1));
    if (col > g.Wp - 4) col = g.Wp - 4;
#pragma nounroll
    for (int q = 0; q < kLanesPerRow; ++q) {
        const float *src = ptr_from((kRowsPerInstr * q + lane / kLanesPerRow) * 4, rp) + col;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * g.in_plane_stride),
                                             (__attribute__((address_space(3))) void *)(buf + p * kModPlane + q * 256), 16, 0,
                                             kModIT == kInTile ? CM_FILL_AUX : 0);
    }
}
// r, g, b of samples first .. first + 3 of this lane's row out of the tile; zero beyond the row
__device__ __forceinline__ void read_tile3(const lds_float *itile, int lane, int first, int W, f4 out[3]) {
    const lds_float *buf = itile + (kModBufs == 2 ? ((first / kModIT) & 1) * 3 * kModPlane : 0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        f4 v = *(const lds_f4 *)(buf + p * kModPlane + lane * kModIT + (first & (kModIT - 1)));
        if (first + 3 >= W) {   // the last quad of a row may be partial (pitched rows)
            if (first >= W) v.x = 0.f;
            if (first + 1 >= W) v.y = 0.f;
            if (first + 2 >= W) v.z = 0.f;
            if (first + 3 >= W) v.w = 0.f;
        }
        out[p] = v;
    }
}
// Advance the input stream by one body: returns samples nxt .. nxt + 3 and keeps the tile protocol.
__device__ __forceinline__ void next_tile3(const Geom &g, lds_float *itile, const float *rp, int lane, int nxt, f4 out[3]) {
    const bool first_of_tile = (nxt & (kModIT - 1)) == 0 && nxt < g.W;
    if (first_of_tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    read_tile3(itile, lane, nxt, g.W, out);
    if (kModBufs == 2) {
        if (first_of_tile && nxt + kModIT < g.W) {      // the other buffer was last read a body ago
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            fill_tile3(g, itile, rp, nxt / kModIT + 1, lane);
        }
    } else if ((nxt & (kModIT - 1)) == kModIT - 4 && nxt + 4 < g.W) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        fill_tile3(g, itile, rp, nxt / kModIT + 1, lane);
    }
}

// ---- the ImageModem byte boundary of the encoders (ref image.py:27-56): mode-'RGB' images in, mode-'L' images out.
// Input rows are W x 3 interleaved bytes; a tile holds 32 pixels = 96 bytes of each of the 64 rows, filled with 16 bytes
// per lane (6 lanes per row, 10 rows per global_load_lds_dwordx4); rows are 16-byte aligned because W % 16 == 0 is
// required for this path.  The composite leaves as bytes through a [64 rows][64 pixels] tile, 64-byte row segments.
constexpr int kInTile3Bytes = 64 * 96;     // bytes
constexpr int kOutTileU8 = 64;             // pixels per row of the byte output tile
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) unsigned lds_word;

__device__ __forceinline__ void fill_tile3_u8(const Geom &g, lds_float *itile, const float *rp, int c, int lane) {
    const int r10 = lane / 6, j = lane - 6 * r10;        // lanes 60..63 idle
    long long off = 96LL * c + 16 * j;                   // byte offset in the row
    const long long last = 3LL * g.W - 16;
    if (off > last) off = last;                          // never read past the row; such bytes are masked by the consumer
#pragma nounroll
    for (int q = 0; q < 7; ++q) {
        const int row = 10 * q + r10;
        const unsigned char *src = (const unsigned char *)ptr_from((row < 64 ? row : 63) * 4, rp) + off;
        if (lane < 60 && row < 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)((lds_byte *)itile + q * 960), 16, 0,
                                             CM_FILL_AUX);
    }
}
// r, g, b of pixels first .. first + 3 of this lane's row (bytes / 255, ref image.py:43-45); zero beyond the row
__device__ __forceinline__ void read_tile3_u8(const lds_float *itile, int lane, int first, int W, f4 out[3]) {
    const lds_word *p = (const lds_word *)((const lds_byte *)itile + lane * 96 + 3 * (first & (kInTile - 1)));
    const unsigned w0 = p[0], w1 = p[1], w2 = p[2];      // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    const float s = 1.0f / 255.0f;
    out[0] = f4{(float)(w0 & 0xffu) * s, (float)(w0 >> 24) * s, (float)((w1 >> 16) & 0xffu) * s, (float)((w2 >> 8) & 0xffu) * s};
    out[1] = f4{(float)((w0 >> 8) & 0xffu) * s, (float)(w1 & 0xffu) * s, (float)(w1 >> 24) * s, (float)((w2 >> 16) & 0xffu) * s};
    out[2] = f4{(float)((w0 >> 16) & 0xffu) * s, (float)((w1 >> 8) & 0xffu) * s, (float)(w2 & 0xffu) * s, (float)(w2 >> 24) * s};
    if (first >= W) out[0] = out[1] = out[2] = f4{0.f, 0.f, 0.f, 0.f};
}
template <bool U8>
__device__ __forceinline__ void first_tile3(const Geom &g, lds_float *itile, const float *rp, int lane, f4 out[3]) {
    if (U8) fill_tile3_u8(g, itile, rp, 0, lane); else fill_tile3(g, itile, rp, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (!U8 && kModBufs == 2 && kModIT < g.W) fill_tile3(g, itile, rp, 1, lane);
    if (U8) read_tile3_u8(itile, lane, 0, g.W, out); else read_tile3(itile, lane, 0, g.W, out);
}
template <bool U8>
__device__ __forceinline__ void next_tile3x(const Geom &g, lds_float *itile, const float *rp, int lane, int nxt, f4 out[3]) {
    if (!U8) { next_tile3(g, itile, rp, lane, nxt, out); return; }
    if ((nxt & (kInTile - 1)) == 0 && nxt < g.W) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    read_tile3_u8(itile, lane, nxt, g.W, out);
    if ((nxt & (kInTile - 1)) == kInTile - 4 && nxt + 4 < g.W) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        fill_tile3_u8(g, itile, rp, (nxt >> 5) + 1, lane);
    }
}
// composite sample -> mode-'L' byte: encode_composite_level (image.py:20-21) then _as_bytes (image.py:7-8)
__device__ __forceinline__ unsigned char composite_byte(float comp) {
    const float v = __builtin_fmaf(0.6f, comp, 0.2f);
#if CM_CVT_PK_U8
    return (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(255.f * v, 0, 0);
#else
    return (unsigned char)__builtin_rintf(255.f * __builtin_fminf(__builtin_fmaxf(v, 0.f), 1.f));
#endif
}
// byte output tile [64 rows][64 pixels] -> 64-byte row segments, 16 rows per wave-instruction
__device__ __forceinline__ void flush_tile1_u8(const Geom &g, const lds_float *otile, const float *op, int first_col, int lane) {
    __builtin_amdgcn_wave_barrier();
    const int chunk = lane & 3;
    const int col = first_col + 16 * chunk;
#pragma nounroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 2) + 16 * q;
        typedef __attribute__((address_space(1))) f4 global_f4;
        unsigned char *dst = (unsigned char *)(unsigned long long)ptr_from(row * 4, op);
        if (dst != nullptr && col < g.W) {   // W % 16 == 0: a 16-byte chunk is inside or outside the row as a whole
            f4 v = *(const lds_f4 *)((const lds_byte *)otile + row * kOutTileU8 + 16 * chunk);
            __builtin_nontemporal_store(v, (global_f4 *)(dst + col));
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// one composite sample into the output tile + the flush when a tile (or the row) is complete
template <bool U8, int kTile>
__device__ __forceinline__ void put_composite(const Geom &g, lds_float *otile_base, const float *op, int lane, int wpos, int n7,
                                              float comp) {
    const int W = g.W;
    if (U8) {
        if (n7 >= 0 && n7 < W) ((lds_byte *)otile_base)[lane * kOutTileU8 + (n7 & (kOutTileU8 - 1))] = composite_byte(comp);
        if (n7 >= 0 && ((n7 & (kOutTileU8 - 1)) == kOutTileU8 - 1 || n7 == W - 1))
            flush_tile1_u8(g, otile_base, op, n7 & ~(kOutTileU8 - 1), lane);
    } else {
        if (n7 >= 0 && n7 < W) otile_base[lane * kTile + (wpos ^ (n7 & (kTile - 1)))] = comp;
        if (n7 >= 0 && ((n7 & (kTile - 1)) == kTile - 1 || n7 == g.Wp - 1))   // the row ends with the quad that holds W - 1
            flush_tile1<kTile>(g, otile_base, op, n7 & ~(kTile - 1), lane);
    }
}
// row pointers of a modulator lane; in the byte mode the strides of Geom count bytes
template <bool U8>
__device__ __forceinline__ void mod_rows(const Geom &g, const LaneCall &lc, const float *&rp, const float *&op) {
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    if (U8) {
        rp = (const float *)((const unsigned char *)g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * row_stride);
        op = lc.store_ok ? (const float *)((unsigned char *)g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride)
                         : nullptr;
    } else {
        rp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * row_stride;
        op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    }
}
// samples per output tile row of the encoders: the QAM encoders write 128-byte row segments (measured 2.31 -> 2.04 ms per
// 1000 frames against 64-byte segments: they are bandwidth-bound and the tile is a single plane, 8 KiB), the SECAM
// encoder keeps 64-byte segments (it is arithmetic-bound: the wider tile cost it 19 %)
#ifndef CM_QAM_MOD_TILE
#define CM_QAM_MOD_TILE 32
#endif
constexpr int kQamModTile = CM_QAM_MOD_TILE, kSecamModTile = 16;
template <int kTile> constexpr int mod_lds_floats() { return kLdsIn3 + 64 * kTile; }   // float mode: 3-plane input tile + output tile
constexpr int kModLdsFloatsU8 = (kInTile3Bytes + 64 * kOutTileU8) / 4;    // byte mode

// DEPTH = 1: encoder-side line averaging (ColorAveragingModem) needs the previous call's components.
// U8: the ImageModem byte boundary fused in (interleaved RGB bytes in, composite bytes out)
// RT: run-time shape - SP is the size of the luma delay window, the delay itself is k.s_p <= SP (any sampling rate)
#ifndef CM_QAM_MOD_WAVES      /* waves per SIMD the register allocation of the QAM encoders aims at */
#define CM_QAM_MOD_WAVES 2
#endif
template <int NP, int SP, int DEPTH, bool U8 = false, bool RT = false>
__global__ __launch_bounds__(64, CM_QAM_MOD_WAVES) void qam_mod_kernel(const ModArgs<NP> args) {
    constexpr int kTile = kQamModTile;
    __shared__ __attribute__((aligned(16))) float lds_store[U8 ? kModLdsFloatsU8 : mod_lds_floats<kTile>()];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + (U8 ? kInTile3Bytes / 4 : kLdsIn3);
    const Geom &g = args.g;
    const ModK<float, NP> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const float *rp, *op;
    mod_rows<U8>(g, lc, rp, op);
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
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int sp = RT ? k.s_p : SP;
    const int T = (g.Wp + sp + 3) & ~3;
    f4 cur[3], nxt[3];
    first_tile3<U8>(g, itile, rp, lane, nxt);
    // interior bodies (round 3; the SECAM encoder has had them since round 2): s_p <= t and t + 3 < W - 1 for the four steps - no
    // guard of QamModCore::step can fail, no carrier index needs a clamp and the output sample n7 = t - s_p lies inside the row
    int tb_mid0 = (sp + 3) & ~3, tb_mid1 = (W - 4) & ~3;
    if (tb_mid1 <= tb_mid0) tb_mid0 = tb_mid1 = 0;
    constexpr int kOT = U8 ? kOutTileU8 : kTile;          // pixels per output tile row
    const int s_flush = (sp + 3) & 3;                     // the step of a body whose output sample ends a quad
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        next_tile3x<U8>(g, itile, rp, lane, tb + 4, nxt);
        const const_f2 *carp = (const_f2 *)g.carrier2 + (tb - sp);
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
            const int n7 = t - sp;
            f2 cc;
            if (EDGE) cc = ((const_f2 *)g.carrier2)[n7 < 0 ? 0 : (n7 > W - 1 ? W - 1 : n7)];
            else cc = carp[s];
            float car[2] = {cc.x, cc.y};
            float y_d = yw[s];                 // luma of sample n7 = t - SP
            if (RT) {                          // ... = t - s_p: a chain of uniform selects instead of a dynamic register index
#pragma unroll
                for (int j = 0; j < SP; ++j)
                    if (sp == j) y_d = yw[SP - j + s];
            }
            float comp = core.template step<EDGE>(k, lk, t, y_d, u, v, car);
            if (EDGE) {
                put_composite<U8, kTile>(g, otile_base, op, lane, wpos, n7, comp);
            } else {
                if (U8) ((lds_byte *)otile_base)[lane * kOutTileU8 + (n7 & (kOutTileU8 - 1))] = composite_byte(comp);
                else otile_base[lane * kTile + (wpos ^ (n7 & (kTile - 1)))] = comp;
                if (s == s_flush && (n7 & (kOT - 1)) == kOT - 1) {
                    if (U8) flush_tile1_u8(g, otile_base, op, n7 & ~(kOT - 1), lane);
                    else flush_tile1<kTile>(g, otile_base, op, n7 & ~(kOT - 1), lane);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SP; ++j) yw[j] = yw[j + 4];
    };
    int tb = 0;
    for (; tb < tb_mid0; tb += 4) body(std::true_type(), tb);
    for (; tb < tb_mid1; tb += 4) body(std::false_type(), tb);
    for (; tb < T; tb += 4) body(std::true_type(), tb);
}

}  // namespace cm
#endif

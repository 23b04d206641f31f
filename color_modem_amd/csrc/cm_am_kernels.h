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

// ---------------------------------------------------------------------------------------------------------------------
// Wave-pair form of the Proto-SECAM decoder (round 3; as niir_demod_pair_kernel below):
//   wave 0 (stage A)  every global load (8-sample input tiles through global_load_lds, two buffers), up3, the band-pass with
//                     |.| (chroma) and the band-stop (luma) at the 3x rate; per body of two steps it leaves the two triples of
//                     each step in a double-buffered hand-over ring
//   wave 1 (stage B)  the chroma low-pass at the 3x rate, BOTH decimators as one packed pair (chroma | luma: Dn3Pk), the luma
//                     delay (LDS ring), the other colour-difference signal from the neighbouring lane, matrix, output tile and
//                     every global store
// ---------------------------------------------------------------------------------------------------------------------
// Input rows of the pair decoders.  float: tiles of 8 samples, two buffers, tile c in buffer c & 1 (fill_tile<8>).  U8: the
// ImageModem byte boundary fused in (image.py:24-25, 62): composite bytes, tiles of 32 samples (fill_tile_u8), level-decoded
// on the way out of the tile; the output tile then holds interleaved RGB bytes (put_rgb<true>, flush_tile_u8).
template <bool U8> struct AmInTile { static constexpr int kIT = U8 ? kInTile : 8, kBufFloats = U8 ? 64 * kInTile / 4 : 64 * 8; };
template <bool U8>
__device__ __forceinline__ void am_fill(const Geom &g, lds_float *itile, const float *xp, int c, int lane) {
    lds_float *buf = itile + (c & 1) * AmInTile<U8>::kBufFloats;
    if (U8) fill_tile_u8(g, buf, xp, c, lane); else fill_tile<8>(g, buf, xp, c, lane);
}
template <bool U8>
__device__ __forceinline__ f2 am_read2(const lds_float *itile, int lane, int first, int W) {
    constexpr int IT = AmInTile<U8>::kIT;
    const lds_float *buf = itile + ((first / IT) & 1) * AmInTile<U8>::kBufFloats;
    f2 v;
    if (U8) {
        typedef __attribute__((address_space(3))) unsigned short lds_u16;
        const unsigned w = *(const lds_u16 *)((const __attribute__((address_space(3))) unsigned char *)buf + lane * IT + (first & (IT - 1)));
        const f4 d = decode_bytes(w);
        v = f2{d.x, d.y};
    } else {
        typedef __attribute__((address_space(3))) f2 lds_f2_;
        v = *(const lds_f2_ *)(buf + lane * IT + (first & (IT - 1)));
    }
    if (first >= W) v.x = 0.f;
    if (first + 1 >= W) v.y = 0.f;
    return v;
}
template <bool U8>
__device__ __forceinline__ const float *am_in_row(const Geom &g, const LaneCall &lc) {
    if (U8) return (const float *)((const unsigned char *)g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.W);
    return g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
}
template <bool U8>
__device__ __forceinline__ const float *am_out_row(const Geom &g, const LaneCall &lc) {
    if (!lc.store_ok) return nullptr;
    if (U8) return (const float *)((unsigned char *)g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride);
    return g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride;
}
constexpr int kProtoIT = 8;
inline __host__ __device__ int proto_ring_slots(int dly) { return dly < 8 ? 8 : (dly < 16 ? 16 : 32); }
// the decoder's luma ring is read BEFORE it is written in a step, so a delay of dly samples needs dly slots, not dly + 1 (dly = 8 at the
// standard rates: 8 slots instead of 16 - with the shorter row delay below 26 KiB of LDS: six workgroups per CU, round 5)
inline __host__ __device__ int proto_demod_ring_slots(int dly) { return dly <= 8 ? 8 : (dly <= 16 ? 16 : 32); }
// floats of dynamic LDS: input tile (two buffers) | hand-over (2 buffers x 6 quantities) | luma delay ring | the row's last 16 samples (the
// interpolator's phase 0 is x[t - 10]) | output tile
template <bool U8> inline int proto_pair_lds_floats(int dly) {
    return 2 * AmInTile<U8>::kBufFloats + 2 * 6 * 128 + proto_demod_ring_slots(dly) * 64 + 4 * 128 + (U8 ? 64 * 3 * 16 / 4 : 3 * 64 * 16);
}

// Stage A's interior bodies: the six aligned samples of one filter (H 0: |band-pass|, 1: band-stop) of a body into their hand-over rows,
// R of them the tail of the previous group.  asm statements: written as plain stores, the three branches around them are merged into
// selects (36 moves per body).  The s_waitcnt lgkmcnt(0) in front of the body's barrier covers them.
template <int R, int H>
__device__ __forceinline__ void proto_put_rows(unsigned slot_addr, const pf2 (&raw)[8], const float (&ae)[8]) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {          // aligned sample a = 3 s + j: row 2 (3 H + j) + s
        const float v = H ? raw[2 + a - R].y : ae[2 + a - R];
        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(slot_addr), "v"(v), "n"((2 * (3 * H + a % 3) + a / 3) * 256) : "memory");
    }
}

template <bool U8>
__global__ __launch_bounds__(128, 2) void proto_demod_pair_kernel(const ProtoDemodArgs args) {
    constexpr int kTile = 16, DEPTH = 1, Q = 6;
    constexpr int kIT = AmInTile<U8>::kIT;
    extern __shared__ __attribute__((aligned(16))) float proto_pair_lds[];
    typedef __attribute__((address_space(3))) f2 lds_f2;
    lds_float *lds = (lds_float *)proto_pair_lds;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const Geom &g = args.g;
    const ProtoDemodK<float> &k = args.k;
    const int lane = threadIdx.x & 63;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const int W = g.W, L = 3 * W;
    const int lat_c = ProtoDemod<float>::lat_chroma(k), lat_y = ProtoDemod<float>::lat_luma(k);
    const int dly = lat_c - lat_y;                       // luma waits for the chroma path
    const int nring = proto_demod_ring_slots(dly);
    const int T = (g.Wp + lat_c + 1) & ~1;
    lds_float *itile = lds;                              // two buffers: tile c lives in buffer c & 1
    lds_float *hand = itile + 2 * AmInTile<U8>::kBufFloats;
    lds_float *ring = hand + 2 * Q * 128;
    lds_float *xdel = ring + nring * 64;                 // 4 blocks of 2 samples per lane: stage A's own delay of the row (+ one block in registers)
    lds_float *otile_base = xdel + 4 * 128;
    // interior bodies: t >= lat_c (every filter behind its delay), t + 1 < W - 6 (the end-of-row latches stay away)
    int t_mid0 = (lat_c + 1) & ~1, t_mid1 = (W - 8) & ~1;
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;

    if (role == 0) {
        // =================================== stage A ===========================================
        // Edge bodies: the scalar stages of cm_am_stages.h (taps and sections out of scalar registers).  Interior bodies (round 5): the
        // interpolator's partial sums and (band-pass | band-stop) in packed float32 - Up3Pk, BpSymPk3 - instantiated per (ge.r, gr.r) so
        // that aligning the raw outputs with the filters' groups is a choice of registers, not of instructions.
        const float *xp = am_in_row<U8>(g, lc);
        Up3<float> up;
        FF3<float, 3> ext, rem;
        up.reset(); ext.reset(); rem.reset();
        for (int j = 0; j < 4; ++j) *(lds_f2 *)(xdel + j * 128 + lane * 2) = f2{0.f, 0.f};
        f2 xlast = {0.f, 0.f};       // the previous body's samples: they go to the ring a body late, the slot read just before holds the block of five bodies ago
        am_fill<U8>(g, itile, xp, 0, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (kIT < W) am_fill<U8>(g, itile, xp, 1, lane);      // tile c + 1 is asked for when tile c is first read
        auto read_x = [&](int first) -> f2 { return am_read2<U8>(itile, lane, first, W); };
        f2 xv = read_x(0);
        auto next_x = [&](int tb) __attribute__((always_inline)) {
            const int nxt = tb + 2;
            if ((nxt & (kIT - 1)) == 0 && nxt < W) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int c = nxt / kIT;
                if ((c + 1) * kIT < W) am_fill<U8>(g, itile, xp, c + 1, lane);
            }
            xv = read_x(nxt);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // hand-over rows: quantity q (|band-pass| 0..2, band-stop 3..5) of step s in row 2 q + s of the body's buffer
        auto edge_body = [&](int tb) __attribute__((always_inline)) {
            const f2 xd = *(const lds_f2 *)(xdel + ((tb >> 1) & 3) * 128 + lane * 2);             // x[tb - 10], x[tb - 9]: written four bodies ago, a body old then
            *(lds_f2 *)(xdel + ((tb >> 1) & 3) * 128 + lane * 2) = xlast;
            xlast = xv;
            lds_float *slot = hand + ((tb >> 1) & 1) * (Q * 128) + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int n1 = tb + s - kAmHalf;
                float u[3], c1[3], y1[3];
                up.template push<false>(k.taps, xv[s], xd[s], u);
                ext.template step<AM_FORM_BP, true>(k.ext, k.ge, L, n1, u, c1);
                rem.template step<AM_FORM_SYM, true>(k.rem, k.gr, L, n1, u, y1);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    slot[(2 * j + s) * 64] = __builtin_fabsf(c1[j]);     // protosecam.py:98 (the factor pi / 2 is in chroma_gain)
                    slot[(2 * (3 + j) + s) * 64] = y1[j];
                }
            }
            next_x(tb);
        };
        int tb = 0;
        for (; tb < t_mid0; tb += 2) edge_body(tb);
        if (tb < t_mid1) {
            TapsPk3 kp;
            BpSymK3 ks;
            kp.load(k.taps);
            ks.load(k.ext, k.rem);
            Up3Pk upk;
            BpSymPk3 er;
            upk.from(up);
            er.from(ext, rem);
            // aligning the raw outputs with the filters' groups (FF3::step: g.r of them belong to the previous group) is a choice of
            // registers per value of r: one uniform three-way branch per filter and body around the six stores, no instruction else
            const int re = k.ge.r, rr = k.gr.r;
            for (; tb < t_mid1; tb += 2) {
                const f2 xd_ = *(const lds_f2 *)(xdel + ((tb >> 1) & 3) * 128 + lane * 2);
                *(lds_f2 *)(xdel + ((tb >> 1) & 3) * 128 + lane * 2) = xlast;
                xlast = xv;
                const pf2 xd = pf2{xd_.x, xd_.y}, xs = pf2{xv.x, xv.y};
                lds_float *slot = hand + ((tb >> 1) & 1) * (Q * 128) + lane;
                pf2 raw[8];                                // raw outputs -2 .. 5 of this body's two groups (.x: |band-pass|, .y: band-stop)
                raw[0] = er.h2;
                raw[1] = er.h1;
                auto step = [&](auto s_tag) __attribute__((always_inline)) {
                    constexpr int s = decltype(s_tag)::value;
                    const pf2 u12 = upk.template push<s>(kp, xs);
                    const pf2 u0 = pk_mul_xk<s, false>(xd, kp.c[10]);
                    raw[2 + 3 * s] = er.template step<0>(ks, u0);
                    raw[3 + 3 * s] = er.template step<0>(ks, u12);
                    raw[4 + 3 * s] = er.template step<1>(ks, u12);
                };
                step(std::integral_constant<int, 0>());
                step(std::integral_constant<int, 1>());
                er.h2 = raw[6];
                er.h1 = raw[7];
                float ae[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ae[i] = __builtin_fabsf(raw[i].x);     // protosecam.py:98 (the factor pi / 2 is in chroma_gain)
                const unsigned slot_addr = (unsigned)(size_t)slot;
                if (re == 0) proto_put_rows<0, 0>(slot_addr, raw, ae); else if (re == 1) proto_put_rows<1, 0>(slot_addr, raw, ae); else proto_put_rows<2, 0>(slot_addr, raw, ae);
                if (rr == 0) proto_put_rows<0, 1>(slot_addr, raw, ae); else if (rr == 1) proto_put_rows<1, 1>(slot_addr, raw, ae); else proto_put_rows<2, 1>(slot_addr, raw, ae);
                next_x(tb);
            }
            upk.to(up);
            er.to(ext, rem);
        }
        for (; tb < T; tb += 2) edge_body(tb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ======================================= stage B ===========================================
    const float *op = am_out_row<U8>(g, lc);
    const long long frame = (long long)args.a.frame_base + lc.frame;
    const bool alt = args.a.line.alternate(frame, lc.line);
    const float w_prev = lc.kk > 0 ? 1.f : 0.f;          // protosecam.py:93-94: the first line of a run has no previous chroma
    const int idx1 = ((lane + 63) & 63) * 4;
    TapsPk3 kp;
    kp.load(k.taps);
    FF3<float, 2> post;
    Dn3Pk dn;                                             // (chroma, luma)
    post.reset();
    dn.reset();
    for (int j = 0; j < nring; ++j) ring[j * 64 + lane] = 0.f;
    lds_float *otile = U8 ? (lds_float *)((__attribute__((address_space(3))) unsigned char *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    auto body = [&](auto edge_tag, auto r_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr int RP = decltype(r_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the hand-over of this body is complete
        const lds_float *slot = hand + ((tb >> 1) & 1) * (Q * 128) + lane;
        float hq[Q][2];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            hq[q][0] = slot[(2 * q) * 64];
            hq[q][1] = slot[(2 * q + 1) * 64];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int t = tb + s, n1 = t - kAmHalf;
            const float c1[3] = {hq[0][s], hq[1][s], hq[2][s]};
            float c2[3];
            post.template step<AM_FORM_GEN, EDGE, false, RP>(k.post, k.gp, L, n1 - k.ge.q, c1, c2);
            const pf2 d = dn.push(kp, pf2{c2[0], hq[3][s]}, pf2{c2[1], hq[4][s]}, pf2{c2[2], hq[5][s]});
            const float chroma = fmaf_(k.chroma_gain, d.x, -1.f), luma = k.luma_gain * d.y;
            const float luma_w = ring[((t - dly) & (nring - 1)) * 64 + lane];      // read first: with dly = nring this is the slot written next
            ring[(t & (nring - 1)) * 64 + lane] = luma;
            const float luma_d = dly ? luma_w : luma;
            const float prev = lane_from(idx1, chroma) * w_prev;
            const float dr = alt ? prev : chroma, db = alt ? chroma : prev;             // protosecam.py:105-108
            Rgb<float> o;
            o.r = fmaf_(k.m[0][0], luma_d, fmaf_(k.m[0][1], dr, k.m[0][2] * db));
            o.g = fmaf_(k.m[1][0], luma_d, fmaf_(k.m[1][1], dr, k.m[1][2] * db));
            o.b = fmaf_(k.m[2][0], luma_d, fmaf_(k.m[2][1], dr, k.m[2][2] * db));
            const int n = t - lat_c;
            if (!EDGE || (n >= 0 && n < W)) put_rgb<U8, kTile>(otile, wpos, n, o);
            if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) {
                if (U8) flush_tile_u8(g, otile_base, op, n & ~(kTile - 1), lane); else flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
            }
        }
    };
    typedef std::integral_constant<int, -1> RX;
    int tb = 0;
    for (; tb < t_mid0; tb += 2) body(std::true_type(), RX(), tb);
    if (k.gp.r == 0) for (; tb < t_mid1; tb += 2) body(std::false_type(), std::integral_constant<int, 0>(), tb);
    else if (k.gp.r == 1) for (; tb < t_mid1; tb += 2) body(std::false_type(), std::integral_constant<int, 1>(), tb);
    else for (; tb < t_mid1; tb += 2) body(std::false_type(), std::integral_constant<int, 2>(), tb);
    for (; tb < T; tb += 2) body(std::true_type(), RX(), tb);
}

struct ProtoModArgs {
    Geom g;
    AmGeom a;
    ProtoModK<float> k;
    int averaging;
};

// ---------------------------------------------------------------------------------------------------------------------
// Wave-pair form of the Proto-SECAM encoder (round 3):
//   wave 0 (stage A)  every global load (three-plane 16-sample input tiles through global_load_lds, as the QAM encoders), the
//                     colour matrix, the previous call's components (ColorAveragingModem), the delay of the shorter path's
//                     input (LDS ring), the pre-correction low-pass at the 1x rate and up3 of the luma; per body of four steps
//                     it leaves (U[3], chroma, plain luma) in a double-buffered hand-over ring
//   wave 1 (stage B)  the band-stop at the 3x rate, the decimator, the carrier, the composite sample, output tile and every
//                     global store
// ---------------------------------------------------------------------------------------------------------------------
template <bool U8> inline int proto_mod_pair_lds_floats(int dly) {
    return (U8 ? kInTile3Bytes / 4 : kLdsIn3) + proto_ring_slots(dly) * 64 + 8 * 64 + 2 * 5 * 256 + (U8 ? 64 * kOutTileU8 / 4 : 64 * 16);
}

// U8: the ImageModem byte boundary fused in (image.py:27-56): interleaved RGB bytes in, composite bytes out (as the QAM encoders)
template <int DEPTH, bool U8 = false>
__global__ __launch_bounds__(128, 2) void proto_mod_pair_kernel(const ProtoModArgs args) {
    constexpr int kTile = 16, Q = 5;
    extern __shared__ __attribute__((aligned(16))) float proto_mod_lds[];
    lds_float *lds = (lds_float *)proto_mod_lds;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const Geom &g = args.g;
    const ProtoModK<float> &k = args.k;
    const int lane = threadIdx.x & 63;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const int W = g.W;
    const int lat_y = ProtoMod<float>::lat_luma(k), lat_c = ProtoMod<float>::lat_chroma(k);
    const int lat = lat_y > lat_c ? lat_y : lat_c;
    const int d_c = lat - lat_c, d_y = lat - lat_y;      // one of them is 0: the shorter path's input waits in the ring
    const int dly = d_c > d_y ? d_c : d_y;
    const int nring = proto_ring_slots(dly);
    const int T = (g.Wp + lat + 3) & ~3;
    lds_float *itile = lds;
    lds_float *ring = itile + (U8 ? kInTile3Bytes / 4 : kLdsIn3);
    lds_float *ydel = ring + nring * 64;                 // the luma fed to the interpolator, 8 steps back (its phase 0 is the sample of ten steps ago:
    lds_float *hand = ydel + 8 * 64;                     // two more in registers - 2 KiB of LDS less: five instead of four workgroups per CU, round 5)
    lds_float *otile_base = hand + 2 * Q * 256;
    // interior bodies: t >= lat + 4 (both paths behind their delays), t + 3 < W - 4
    int t_mid0 = (lat + 4 + 3) & ~3, t_mid1 = (W - 8) & ~3;
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;
    const float *rp, *op;
    mod_rows<U8>(g, lc, rp, op);

    if (role == 0) {
        // =================================== stage A ===========================================
        // (round 5: the interior bodies run the interpolator on pairs of partial sums - Up3Pk, as stage A of the decoder; the edge bodies
        // keep the scalar one with its taps out of scalar registers)
        const ProtoModK<float> &ka = k;
        TapsPk3 kp;
        kp.load(k.taps);
        Up3Pk upk;
        const long long frame = (long long)args.a.frame_base + lc.frame;
        const int line = DEPTH ? lc.line - 2 : lc.line;       // the line that is modulated
        const bool alt = args.a.line.alternate(frame, line);
        const bool have_prev = lc.kk > 0;
        const int idx1 = ((lane + 63) & 63) * 4;
        FF1<float, 2> pre;
        Up3<float> up;
        pre.reset();
        up.reset();
        for (int j = 0; j < nring; ++j) ring[j * 64 + lane] = 0.f;
        for (int j = 0; j < 8; ++j) ydel[j * 64 + lane] = 0.f;
        float yd1 = 0.f, yd2 = 0.f;
        f4 cur[3], nxt[3];
        first_tile3<U8>(g, itile, rp, lane, nxt);
        auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
            next_tile3x<U8>(g, itile, rp, lane, tb + 4, nxt);
            float hq[Q][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int t = tb + s;
                const float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
                float y = fmaf_(ka.e[0][0], r, fmaf_(ka.e[0][1], gg, ka.e[0][2] * b));
                float dr = fmaf_(ka.e[1][0], r, fmaf_(ka.e[1][1], gg, ka.e[1][2] * b));
                float db = fmaf_(ka.e[2][0], r, fmaf_(ka.e[2][1], gg, ka.e[2][2] * b));
                if (DEPTH) {
                    const float yp = lane_from(idx1, y), drp = lane_from(idx1, dr), dbp = lane_from(idx1, db);
                    if (have_prev) {
                        y = yp;                                  // comb.py:147
                        dr = 0.5f * (dr + drp);                  // comb.py:148-149
                        db = 0.5f * (db + dbp);
                    }
                }
                const float d = alt ? db : dr;                   // protosecam.py:75-78
                const float late = d_c > 0 ? d : y;              // the shorter path's input waits dly samples
                ring[(t & (nring - 1)) * 64 + lane] = late;
                const float waited = ring[((t - dly) & (nring - 1)) * 64 + lane];
                const float d_in = d_c > 0 ? waited : d, y_in = d_c > 0 ? y : (d_y > 0 ? waited : y);
                const float c = pre.template step<AM_FORM_GEN, EDGE>(ka.pre, ka.s_c, ka.width, t - d_c, d_in);
                hq[3][s] = fmaf_(0.125f * ka.pre_gain, c, 0.125f);
                hq[4][s] = y_in;
                // hand-over rows: 0 the interpolator's phase 0 of the four steps, 1 / 2 its phases (1, 2) of steps 0, 1 / 2, 3
                float u0 = 0.f;
                pf2 u12 = pf2{0.f, 0.f};
                const int i_y = t - d_y;
                if (ka.luma_filter) {
                    const float y_fed = (!EDGE || (i_y >= 0 && i_y < ka.width)) ? y_in : 0.f;
                    const float y8 = ydel[(t & 7) * 64 + lane];       // read before written: the sample of eight steps ago
                    ydel[(t & 7) * 64 + lane] = y_fed;
                    const float y_del = yd2;
                    yd2 = yd1;
                    yd1 = y8;
                    if (EDGE) {
                        float u[3];
                        up.template push<false>(ka.taps, y_fed, y_del, u);
                        u0 = u[0];
                        u12 = pf2{u[1], u[2]};
                    } else {
                        u12 = upk.template push<0>(kp, pk_lo(y_fed));
                        u0 = kp.c[10].x * y_del;
                    }
                }
                hq[0][s] = u0;
                hq[1 + (s >> 1)][2 * (s & 1)] = u12.x;
                hq[1 + (s >> 1)][2 * (s & 1) + 1] = u12.y;
            }
            lds_float *slot = hand + ((tb >> 2) & 1) * (Q * 256) + lane * 4;
#pragma unroll
            for (int q = 0; q < Q; ++q) *(lds_f4 *)(slot + q * 256) = f4{hq[q][0], hq[q][1], hq[q][2], hq[q][3]};
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        int tb = 0;
        for (; tb < t_mid0; tb += 4) body(std::true_type(), tb);
        if (tb < t_mid1) {
            upk.from(up);
            for (; tb < t_mid1; tb += 4) body(std::false_type(), tb);
            upk.to(up);
        }
        for (; tb < T; tb += 4) body(std::true_type(), tb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ======================================= stage B ===========================================
    // (round 5: the interior bodies run the decimator two steps at a time on pairs of accumulators - Dn3Two - and are instantiated per
    // gr.r, so that aligning the band-stop's raw outputs with its groups is a choice of registers; the edge bodies keep the scalar stages)
    const ProtoModK<float> &kb = k;
    Dn3TwoK dk;
    dk.load(k.taps);
    Dn3Two dn2;
    const long long frame = (long long)args.a.frame_base + lc.frame;
    const int line = DEPTH ? lc.line - 2 : lc.line;
    float cph, sph;
    {
        const double phi = args.a.line.start_phase(frame, line);
        cph = (float)cos(phi);
        sph = (float)sin(phi);
    }
    FF3<float, 3> rem;
    Dn3<float> dn;
    rem.reset();
    dn.reset();
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    auto body = [&](auto edge_tag, auto r_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr int RR = decltype(r_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the hand-over of this body is complete
        const lds_float *slot = hand + ((tb >> 2) & 1) * (Q * 256) + lane * 4;
        f4 hq[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) hq[q] = *(const lds_f4 *)(slot + q * 256);
        float lumas[4] = {hq[4][0], hq[4][1], hq[4][2], hq[4][3]};
        if (kb.luma_filter) {
            float y1[4][3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float u[3] = {hq[0][s], hq[1 + (s >> 1)][2 * (s & 1)], hq[1 + (s >> 1)][2 * (s & 1) + 1]};
                rem.template step<AM_FORM_SYM, EDGE, false, RR>(kb.rem, kb.gr, 3 * kb.width, tb + s - d_y - kAmHalf, u, y1[s]);
                if (EDGE) lumas[s] = kb.luma_gain * dn.template push<false>(kb.taps, y1[s]);
            }
            if (!EDGE) {
                const pf2 l01 = dn2.push2(dk, y1[0][0], pf2{y1[0][1], y1[0][2]}, y1[1][0], pf2{y1[1][1], y1[1][2]});
                const pf2 l23 = dn2.push2(dk, y1[2][0], pf2{y1[2][1], y1[2][2]}, y1[3][0], pf2{y1[3][1], y1[3][2]});
                lumas[0] = kb.luma_gain * l01.x;
                lumas[1] = kb.luma_gain * l01.y;
                lumas[2] = kb.luma_gain * l23.x;
                lumas[3] = kb.luma_gain * l23.y;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            const float luma = lumas[s];
            const int n = t - lat;
            int nc = n;
            if (EDGE) nc = n < 0 ? 0 : (n > W - 1 ? W - 1 : n);
            const f2 cs = ((const_f2 *)args.a.carrier)[nc];
            const float cosp = fmaf_(cph, cs.x, -(sph * cs.y));      // cos(phi + n step)
            put_composite<U8, kTile>(g, otile_base, op, lane, wpos, n, fmaf_(cosp, hq[3][s], luma));
        }
    };
    typedef std::integral_constant<int, -1> RX;
    int tb = 0;
    for (; tb < t_mid0; tb += 4) body(std::true_type(), RX(), tb);
    if (tb < t_mid1) {
        dn2.from(dn);
        if (kb.gr.r == 0) for (; tb < t_mid1; tb += 4) body(std::false_type(), std::integral_constant<int, 0>(), tb);
        else if (kb.gr.r == 1) for (; tb < t_mid1; tb += 4) body(std::false_type(), std::integral_constant<int, 1>(), tb);
        else for (; tb < t_mid1; tb += 4) body(std::false_type(), std::integral_constant<int, 2>(), tb);
        dn2.to(dn);
    }
    for (; tb < T; tb += 4) body(std::true_type(), RX(), tb);
}

// ---- NIIR / SECAM-IV ------------------------------------------------------------------------------------------------------
struct NiirDemodArgs {
    Geom g;
    AmGeom a;
    NiirDemodK<float> k;
    NiirDemodK<double> kd;     // the float64 hue path: taps, band-pass, low-pass, c_pm, alt_scale (cm_am_stages.h: NiirHue)
    const double *syn;         // [2][3 W]: the band-passed reference carriers of niir.py:107-110 for cos / sin(n step) (cm_am_plan.h: build_niir_syn)
    double line_phase_shift, bandpass_phase_shift, carrier_phase_step;
    int strip;                 // 0: demodulate_components(..., strip_chroma=False)
};

__device__ __forceinline__ double lane_from(int byte_index, double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_index, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_index, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-pair form of the NIIR decoder (round 3; the structure of cm_kernels.h: run_pair), HUE PATH IN FLOAT64 since round 4
// (cm_am_stages.h: NiirHue - the float32 front end left isolated samples beyond 1e-5 wherever the decimated hue pair gets short).
// Two wavefronts walk the same 64 calls, two steps per body:
//   wave 0 (stage A)  every global load (8-sample input tiles through global_load_lds, two buffers), up3 -> band-pass -> |.| ->
//                     low-pass in float64 (NiirFront<double>; the interpolator's phase 0 is the row sample of ten steps ago, read
//                     back from the luma source ring), the band-pass triples delayed by the low-pass delay (LDS ring of doubles),
//                     the phasor phasemod = M / S (float64, reciprocal seed + Newton), the saturation decimator (float32); per body
//                     it leaves (phasor[3] as doubles, saturation) of both steps in a double-buffered hand-over ring and the two
//                     row samples in the luma source delay ring
//   wave 1 (stage B)  the previous call's phasor from the neighbouring lane (on the first line of a run: the synthetic reference of
//                     niir.py:107-110 - linear in (sin, cos) of the line's start phase, so two float64 tables of the plan replace
//                     the second interpolator + band-pass round 3 carried), the hue products and their two decimators in float64
//                     (NiirHue<double>), the two re-modulation decimators as one packed float32 pair, normalisation / rotation /
//                     offset / matrix (niir_finish), output tile and every global store
// One s_barrier per body; the main pass and the sparse first-line pass share ONE launch (workgroups [0, n_first) are the first-line
// ones; they differ in stage B only).
// ---------------------------------------------------------------------------------------------------------------------
struct NiirPairArgs {
    NiirDemodArgs m;           // m.g: the main pass
    Geom gf;                   // the first-line pass (sparse), when n_first > 0
    int n_first;
};
constexpr int kNiirIT = 8;                                 // samples per input tile row
constexpr int kNiirHandQ = 7;                              // hand-over quantities per body, 8 bytes per lane each: phasor[3] x 2 steps (doubles), saturation x 2 steps
// floats of dynamic LDS: input tile (two buffers) | M delay ring (doubles) | hand-over (2 buffers) | luma source ring | output tile
// slots of the M delay ring: the power of two that holds q_l + 1 triples (q_l = 3 at 13.5 MHz: 4 slots, 6 KiB)
inline __host__ __device__ int niir_mring_slots(int q_l) { return q_l < 2 ? 2 : (q_l < 4 ? 4 : 8); }
constexpr int kNiirRing = 8;   // (q_l <= 7, checked by the host)
template <bool U8> inline int niir_pair_lds_floats(int lat, int q_l) {
    const int hb = (lat + 1) >> 1;
    return 2 * AmInTile<U8>::kBufFloats + niir_mring_slots(q_l) * 3 * 128 + 2 * kNiirHandQ * 128 + (hb + 3) * 128 + (U8 ? 64 * 3 * 16 / 4 : 3 * 64 * 16);
}

template <bool FIRST, bool U8>
__device__ __forceinline__ void niir_pair_body(const NiirDemodArgs &args, const Geom &g, int block, lds_float *lds, int role) {
    constexpr int kTile = 16, DEPTH = 1, Q = kNiirHandQ;
    constexpr int kIT = AmInTile<U8>::kIT;
    typedef __attribute__((address_space(3))) f2 lds_f2;
    typedef __attribute__((address_space(3))) double lds_double;
    const NiirDemodK<float> &k = args.k;
    const int lane = threadIdx.x & 63;
    const LaneCall lc = locate_call(g, block, DEPTH, lane);
    const int W = g.W;
    const int q_l = k.gl.q;
    const int lat = 2 * kAmHalf + 1 + k.gb.q + q_l;
    const int T = (g.Wp + lat + 1) & ~1;
    const int hb = (lat + 1) >> 1, NB = hb + 3;            // luma source ring: blocks of 2 samples, hb bodies of delay
    lds_float *itile = lds;                                 // two buffers: tile c lives in buffer c & 1
    lds_double *mring = (lds_double *)(itile + 2 * AmInTile<U8>::kBufFloats);
    const int nring = niir_mring_slots(q_l);
    lds_float *hand = (lds_float *)(mring + nring * 3 * 64);
    lds_float *xring = hand + 2 * Q * 128;
    lds_float *otile_base = xring + NB * 128;
    // interior bodies: t >= lat + 4, t + 1 < W - 6
    int t_mid0 = (lat + 4 + 1) & ~1, t_mid1 = (W - 8) & ~1;
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;
    const long long frame = (long long)args.a.frame_base + lc.frame;

    if (role == 0) {
        // =================================== stage A ===========================================
        NiirDemodK<double> kd = args.kd;       // taps in scalar registers (21 distinct values), the sections' 14 coefficients in vector ones
#pragma unroll
        for (int j = 0; j < 3; ++j) { pin_vgpr(kd.bp.na1[j]); pin_vgpr(kd.bp.na2[j]); }
#pragma unroll
        for (int j = 0; j < 2; ++j) { pin_vgpr(kd.lp.na1[j]); pin_vgpr(kd.lp.na2[j]); pin_vgpr(kd.lp.b1[j]); pin_vgpr(kd.lp.b2[j]); }
        TapsPk3 kp;
        kp.load(k.taps);
        const float *xp = am_in_row<U8>(g, lc);
        NiirFront<double> front;
        Dn3S dn_sat;
        float s_prev[3] = {0.f, 0.f, 0.f};                  // S of the previous triple (the decimators run one triple late)
        front.reset();
        dn_sat.reset();
        for (int j = 0; j < nring * 3; ++j) mring[j * 64 + lane] = 0.0;
        for (int j = 0; j < NB; ++j) *(lds_f2 *)(xring + j * 128 + lane * 2) = f2{0.f, 0.f};
        am_fill<U8>(g, itile, xp, 0, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (kIT < W) am_fill<U8>(g, itile, xp, 1, lane);      // tile c + 1 is asked for when tile c is first read
        auto read_x = [&](int first) -> f2 { return am_read2<U8>(itile, lane, first, W); };
        f2 xv = read_x(0);
        int wx = 0, rx = NB - kAmHalf / 2;                     // rx: the block written five bodies ago = x[tb - 10], x[tb - 9]
        auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            const f2 xd = *(const lds_f2 *)(xring + rx * 128 + lane * 2);
            rx = rx + 1 == NB ? 0 : rx + 1;
            lds_float *slot = hand + ((tb >> 1) & 1) * (Q * 128) + lane * 2;
            float sat2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int t = tb + s;
                double m[3], sv[3];
                front.template step<EDGE, false, true>(kd, t, (double)xv[s], (double)xd[s], m, sv);
                const int wr = (t & (nring - 1)) * 3, rd = ((t - q_l) & (nring - 1)) * 3;
#pragma unroll
                for (int j = 0; j < 3; ++j) mring[(wr + j) * 64 + lane] = m[j];
                double md[3], ph[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) md[j] = mring[(rd + j) * 64 + lane];
                niir_phasemod<EDGE>(kd, t - kAmHalf - kd.gb.q - q_l, md, sv, ph);
#pragma unroll
                for (int j = 0; j < 3; ++j) *(lds_double *)(slot + (2 * j + s) * 128) = ph[j];
                sat2[s] = k.sat_gain * dn_sat.push(kp, s_prev);
#pragma unroll
                for (int j = 0; j < 3; ++j) s_prev[j] = (float)sv[j];
            }
            *(lds_f2 *)(slot + 6 * 128) = f2{sat2[0], sat2[1]};
            *(lds_f2 *)(xring + wx * 128 + lane * 2) = xv;
            wx = wx + 1 == NB ? 0 : wx + 1;
            const int nxt = tb + 2;
            if ((nxt & (kIT - 1)) == 0 && nxt < W) {
                // first read of tile c = nxt / kIT: its fill was issued a tile ago; the other buffer was last read a body ago
                // (lgkmcnt(0) at the barrier below) and takes tile c + 1 now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int c = nxt / kIT;
                if ((c + 1) * kIT < W) am_fill<U8>(g, itile, xp, c + 1, lane);
            }
            xv = read_x(nxt);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        int tb = 0;
        for (; tb < t_mid0; tb += 2) body(std::true_type(), tb);
        for (; tb < t_mid1; tb += 2) body(std::false_type(), tb);
        for (; tb < T; tb += 2) body(std::true_type(), tb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ======================================= stage B ===========================================
    const float *op = am_out_row<U8>(g, lc);
    NiirLineK<float> lk;
    {   // niir.py:117-124, 148-157
        lk.alt = args.a.line.alternate(frame, lc.line);
        const double shift = lk.alt ? -args.line_phase_shift : args.line_phase_shift;
        const double ps = (lk.alt ? 0.0 : args.line_phase_shift) + 3.14159265358979323846 - args.bandpass_phase_shift;
        lk.sin_shift = (float)sin(shift);
        lk.cos_shift = (float)cos(shift);
        lk.sin_ps = (float)sin(ps);
        lk.cos_ps = (float)cos(ps);
    }
    double syn_s = 0.0, syn_c = 0.0;          // the reference of the first line of a run: +-sin(phi + n step) = syn_s cos(n step) + syn_c sin(n step)
    if (FIRST) {
        const double phi = args.a.line.start_phase(frame, lc.line - 2);
        const double sg = args.a.line.alternate(frame, lc.line - 2) ? -1.0 : 1.0;
        syn_s = sg * sin(phi);
        syn_c = sg * cos(phi);
    }
    const NiirDemodK<double> &kd = args.kd;
    TapsPk3 kp;
    kp.load(k.taps);
    const int idx1 = ((lane + 63) & 63) * 4;
    NiirHue<double> hue;
    Dn3Pk dn_car;
    hue.reset();
    dn_car.reset();
    lds_float *otile = U8 ? (lds_float *)((__attribute__((address_space(3))) unsigned char *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const bool strip = args.strip != 0;
    const bool odd = (lat & 1) != 0;
    const int L = 3 * W;
    int ra = NB - hb, rb = NB - hb + 1;         // luma source blocks of bodies b - hb and b - hb + 1
    if (rb >= NB) rb -= NB;
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the hand-over of this body is complete
        const lds_float *slot = hand + ((tb >> 1) & 1) * (Q * 128) + lane * 2;
        double pq[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pq[j][0] = *(const lds_double *)(slot + (2 * j) * 128);
            pq[j][1] = *(const lds_double *)(slot + (2 * j + 1) * 128);
        }
        const f2 sat2 = *(const lds_f2 *)(slot + 6 * 128);
        const f2 xa = *(const lds_f2 *)(xring + ra * 128 + lane * 2), xb = *(const lds_f2 *)(xring + rb * 128 + lane * 2);
        ra = ra + 1 == NB ? 0 : ra + 1;
        rb = rb + 1 == NB ? 0 : rb + 1;
        const float cd[2] = {odd ? xa.y : xa.x, odd ? xb.x : xa.y};      // composite[t - lat] of the two steps
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int t = tb + s;
            const double p[3] = {pq[0][s], pq[1][s], pq[2][s]};
            double pv[3];
            const int n3 = t - kAmHalf - k.gb.q - q_l;
            if (FIRST) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int q = 3 * n3 + j;
                    pv[j] = (q >= 0 && q < L) ? __builtin_fma(syn_s, args.syn[q], syn_c * args.syn[L + q]) : 0.0;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) pv[j] = lane_from(idx1, p[j]);
            }
            double sp, cp, car[3], acar[3];
            hue.template step<EDGE, false>(kd.taps, kd.alt_scale, W, n3, p, pv, lk.alt, sp, cp, car, acar);
            const pf2 r2 = dn_car.push(kp, pf2{(float)car[0], (float)acar[0]}, pf2{(float)car[1], (float)acar[1]}, pf2{(float)car[2], (float)acar[2]});
            NiirOut<float> o;
            o.sinphi = (float)sp;
            o.cosphi = (float)cp;
            o.sat = sat2[s];
            o.sinc = k.third * r2.x;
            o.cosc = k.third * r2.y;
            const int n = t - lat;
            if (!EDGE || (n >= 0 && n < W)) put_rgb<U8, kTile>(otile, wpos, n, niir_finish(k, lk, o, cd[s], strip));
            if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) {
                if (U8) flush_tile_u8(g, otile_base, op, n & ~(kTile - 1), lane); else flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
            }
        }
    };
    int tb = 0;
    for (; tb < t_mid0; tb += 2) body(std::true_type(), tb);
    for (; tb < t_mid1; tb += 2) body(std::false_type(), tb);
    for (; tb < T; tb += 2) body(std::true_type(), tb);
}

template <bool U8>
__global__ __launch_bounds__(128, 2) void niir_demod_pair_kernel(const NiirPairArgs args) {
    extern __shared__ __attribute__((aligned(16))) float niir_pair_lds[];
    lds_float *lds = (lds_float *)niir_pair_lds;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
    if (bid < args.n_first) niir_pair_body<true, U8>(args.m, args.gf, bid, lds, role);
    else niir_pair_body<false, U8>(args.m, args.m.g, bid - args.n_first, lds, role);
}

struct NiirModArgs {
    Geom g;
    AmGeom a;
    NiirModK<float> k;
    const float *noise;        // [call][2][W]: the (db, dr) noise of niir.py:45-46 / 193-194 per call, or null (noise_level 0)
};

// DEPTH = 1: HueCorrectingNiirModem (niir.py:181-202): a call modulates line - 2 with the previous call's luma, the
// saturation-weighted mean hue of both calls and the previous call's saturation (previous call = neighbouring lane).
// The three input planes arrive through 16-sample LDS tiles (global_load_lds, as the QAM encoders; round 2 read them with one
// 16-byte load per lane and plane).  U8: the ImageModem byte boundary fused in (interleaved RGB bytes in, composite bytes out).
// RING: slots of the luma delay ring (8 / 16 / 32, the launch picks the smallest above the pre-correction shift: round 5 - with the fixed
// 32 slots = 8 KiB a workgroup of this one-wave kernel took 24.5 KiB of LDS, six waves per CU)
template <int DEPTH, bool U8 = false, int RING = kAmRing>
__global__ __launch_bounds__(64, 2) void niir_mod_kernel(const NiirModArgs args) {
    constexpr int kTile = 16;
    constexpr int kIn = U8 ? kInTile3Bytes / 4 : kLdsIn3, kOut = U8 ? 64 * kOutTileU8 / 4 : 64 * kTile;
    __shared__ __attribute__((aligned(16))) float lds_store[kIn + kOut + RING * 64];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + kIn;
    lds_float *ring = otile_base + kOut;
    const Geom &g = args.g;
    const NiirModK<float> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const float *rp, *op;
    mod_rows<U8>(g, lc, rp, op);
    const long long frame = (long long)args.a.frame_base + lc.frame;
    const int line = DEPTH ? lc.line - 2 : lc.line;       // the line that is modulated (niir.py:202)
    const bool alt = args.a.line.alternate(frame, line);
    float cph, sph;
    {
        const double phi = args.a.line.start_phase(frame, line);
        cph = (float)cos(phi);
        sph = (float)sin(phi);
    }
    const bool have_prev = lc.kk > 0;
    const int idx1 = ((lane + 63) & 63) * 4;
    NiirMod<float> st;
    st.reset();
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W, s_c = k.s_c;
    const int T = (g.Wp + s_c + 3) & ~3;
    for (int j = 0; j < RING; ++j) ring[j * 64 + lane] = 0.f;
    const float *np = args.noise ? args.noise + 2LL * lc.call * W : nullptr;
    f4 cur[3], nxt[3], nz[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
    first_tile3<U8>(g, itile, rp, lane, nxt);
    for (int tb = 0; tb < T; tb += 4) {
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        next_tile3x<U8>(g, itile, rp, lane, tb + 4, nxt);
        if (np) {
            nz[0] = load_luma<false>(np, tb, true, W);
            nz[1] = load_luma<false>(np + W, tb, true, W);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            const float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
            float y = fmaf_(k.e[0][0], r, fmaf_(k.e[0][1], gg, k.e[0][2] * b));
            float db = row3(k.e[1][0], k.e[1][1], k.e[1][2], r, gg, b);      // (unit rows - modulate_components - keep the sign of a zero: cm_am_stages.h)
            float dr = row3(k.e[2][0], k.e[2][1], k.e[2][2], r, gg, b);
            if (DEPTH) {
                float py = lane_from(idx1, y), pdb = lane_from(idx1, db), pdr = lane_from(idx1, dr);
                float pr = lane_from(idx1, r), pg = lane_from(idx1, gg), pb = lane_from(idx1, b);
                if (!have_prev) { py = y; pdb = db; pdr = dr; pr = r; pg = gg; pb = b; }         // niir.py:182-186
                float odb, odr;
                niir_hue_pixel<U8>(k.ed, r, gg, b, pr, pg, pb, db, dr, pdb, pdr, nz[0][s], nz[1][s], odb, odr);
                y = py;
                db = odb;
                dr = odr;
            } else {
                niir_offset_pixel<U8>(k.ed, r, gg, b, db, dr, nz[0][s], nz[1][s], np != nullptr);
            }
            if (t >= W) db = dr = 0.f;                                    // beyond the row the filter is fed its last sample anyway
            ring[(t & (RING - 1)) * 64 + lane] = y;
            const float y_d = ring[((t - s_c) & (RING - 1)) * 64 + lane];
            const int n = t - s_c;
            int nc = n < 0 ? 0 : (n > W - 1 ? W - 1 : n);
            const f2 cs = ((const_f2 *)args.a.carrier)[nc];
            const float sn = fmaf_(sph, cs.x, cph * cs.y), cn = fmaf_(cph, cs.x, -(sph * cs.y));
            const float c = st.step(k, t, db, dr, alt, sn, cn);
            put_composite<U8, kTile>(g, otile_base, op, lane, wpos, n, y_d + c);
        }
    }
}

}  // namespace cm
#endif

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
    // interior bodies: every stage index of the four steps inside its sequence (t >= lat_c puts every filter behind its
    // delay, t + 3 < W - 4 keeps the end-of-row latches away) - no guard, no clamp, no zero test
    int t_mid0 = (lat_c + 3) & ~3, t_mid1 = (W - 8) & ~3;
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const f4 xn = load_luma<false>(xp, tb + 4, EDGE, W);      // next body's samples: a body hides the latency
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            float luma, chroma;
            st.template step<EDGE>(k, t, xv[s], luma, chroma);
            ring[(t & (kAmRing - 1)) * 64 + lane] = luma;
            const float luma_d = ring[((t - dly) & (kAmRing - 1)) * 64 + lane];
            const float prev = lane_from(idx1, chroma) * w_prev;
            const float dr = alt ? prev : chroma, db = alt ? chroma : prev;             // protosecam.py:105-108
            Rgb<float> o;
            o.r = fmaf_(k.m[0][0], luma_d, fmaf_(k.m[0][1], dr, k.m[0][2] * db));
            o.g = fmaf_(k.m[1][0], luma_d, fmaf_(k.m[1][1], dr, k.m[1][2] * db));
            o.b = fmaf_(k.m[2][0], luma_d, fmaf_(k.m[2][1], dr, k.m[2][2] * db));
            const int n = t - lat_c;
            if (!EDGE || (n >= 0 && n < W)) put_rgb<false, kTile>(otile, wpos, n, o);
            if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
        }
        xv = xn;
    };
    int tb = 0;
    for (; tb < t_mid0; tb += 4) body(std::true_type(), tb);
    for (; tb < t_mid1; tb += 4) body(std::false_type(), tb);
    for (; tb < T; tb += 4) body(std::true_type(), tb);
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
    // interior bodies: t >= lat + 4 (both paths behind their delays), t + 3 < W - 4
    int t_mid0 = (lat + 4 + 3) & ~3, t_mid1 = (W - 8) & ~3;
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            cur[p] = nxt[p];
            nxt[p] = load_luma<false>(rp + p * g.in_plane_stride, tb + 4, EDGE, W);
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
            st.template step<EDGE>(k, t - d_c, d_in, t - d_y, y_in, luma, chroma);
            const int n = t - lat;
            int nc = n;
            if (EDGE) nc = n < 0 ? 0 : (n > W - 1 ? W - 1 : n);
            const f2 cs = ((const_f2 *)args.a.carrier)[nc];
            const float cosp = fmaf_(cph, cs.x, -(sph * cs.y));      // cos(phi + n step)
            put_composite<false, kTile>(g, otile_base, op, lane, wpos, n, fmaf_(cosp, chroma, luma));
        }
    };
    int tb = 0;
    for (; tb < t_mid0; tb += 4) body(std::true_type(), tb);
    for (; tb < t_mid1; tb += 4) body(std::false_type(), tb);
    for (; tb < T; tb += 4) body(std::true_type(), tb);
}

// ---- NIIR / SECAM-IV ------------------------------------------------------------------------------------------------------
struct NiirDemodArgs {
    Geom g;
    AmGeom a;
    NiirDemodK<float> k;
    double line_phase_shift, bandpass_phase_shift, carrier_phase_step;
    int strip;                 // 0: demodulate_components(..., strip_chroma=False)
};

constexpr int kNiirRing = 8;   // triples of the band-pass output waiting for the low-pass (q_l <= 7, checked by the host)

// FIRST = false: the main pass - the previous call's phase reference comes from the neighbouring lane; calls that open a run
//                are computed (the next call needs them) but written by the other pass (Geom::skip_first).
// FIRST = true:  one lane per run, its first call only (Geom::sparse): the phase reference is the synthetic carrier of
//                niir.py:107-110, produced by a second interpolator + band-pass in the same lane.
template <bool FIRST>
__global__ __launch_bounds__(64, 1) void niir_demod_kernel(const NiirDemodArgs args) {
    constexpr int kTile = 16, DEPTH = 1;
    __shared__ __attribute__((aligned(16))) float lds_store[3 * 64 * kTile + (FIRST ? 2 : 1) * kNiirRing * 3 * 64];
    lds_float *otile_base = (lds_float *)lds_store;
    lds_float *ring = otile_base + 3 * 64 * kTile;
    lds_float *ring_syn = ring + kNiirRing * 3 * 64;
    const Geom &g = args.g;
    NiirDemodK<float> k = args.k;
    pin_taps3(k.taps);
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
    const float *op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    const long long frame = (long long)args.a.frame_base + lc.frame;
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
    float syn_s = 0.f, syn_c = 0.f;       // +-(sin, cos) of the start phase of line - 2
    if (FIRST) {
        const double phi = args.a.line.start_phase(frame, lc.line - 2);
        const float sg = args.a.line.alternate(frame, lc.line - 2) ? -1.f : 1.f;
        syn_s = sg * (float)sin(phi);
        syn_c = sg * (float)cos(phi);
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    NiirFront<float> front;
    NiirBack<float> back;
    NiirSyn<float> syn;
    front.reset();
    back.reset();
    if (FIRST) syn.reset();
    lds_float *otile = otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int q_l = k.gl.q;
    const int lat = 2 * kAmHalf + 1 + k.gb.q + q_l;
    const int T = (g.Wp + lat + 3) & ~3;
    for (int j = 0; j < (FIRST ? 2 : 1) * kNiirRing * 3; ++j) ring[j * 64 + lane] = 0.f;
    const bool strip = args.strip != 0;
    f4 xv = load_luma<false>(xp, 0, true, W);
    // interior bodies (as in proto_demod_kernel): t >= lat + 4, t + 3 < W - 4
    int t_mid0 = (lat + 4 + 3) & ~3, t_mid1 = (W - 8) & ~3;
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const f4 xn = load_luma<false>(xp, tb + 4, EDGE, W);
        const f4 cd = load_luma<false>(xp, tb - lat, EDGE, W);        // composite[n5 ..]: the luma source of this body's outputs
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            float m[3], sv[3], md[3], p[3], pv[3];
            front.template step<EDGE>(k, t, xv[s], m, sv);
            const int wr = (t & (kNiirRing - 1)) * 3, rd = ((t - q_l) & (kNiirRing - 1)) * 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) ring[(wr + j) * 64 + lane] = m[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) md[j] = ring[(rd + j) * 64 + lane];
            const int n3 = t - kAmHalf - k.gb.q - q_l;
            niir_phasemod<EDGE>(k, n3, md, sv, p);
            if (FIRST) {
                float xs = 0.f;
                if (!EDGE || t < W) {
                    const f2 cs = ((const_f2 *)args.a.carrier)[t];
                    xs = fmaf_(syn_s, cs.x, syn_c * cs.y);            // +-sin(phi + t step)
                }
                float ms[3];
                syn.template step<EDGE>(k, t, xs, ms);
#pragma unroll
                for (int j = 0; j < 3; ++j) ring_syn[(wr + j) * 64 + lane] = ms[j];
#pragma unroll
                for (int j = 0; j < 3; ++j) pv[j] = k.g_b * ring_syn[(rd + j) * 64 + lane];
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) pv[j] = lane_from(idx1, p[j]);
            }
            const NiirOut<float> o = back.template step<EDGE>(k, n3, p, pv, sv, lk.alt);
            const int n = t - lat;
            if (!EDGE || (n >= 0 && n < W)) put_rgb<false, kTile>(otile, wpos, n, niir_finish(k, lk, o, cd[s], strip));
            if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
        }
        xv = xn;
    };
    int tb = 0;
    for (; tb < t_mid0; tb += 4) body(std::true_type(), tb);
    for (; tb < t_mid1; tb += 4) body(std::false_type(), tb);
    for (; tb < T; tb += 4) body(std::true_type(), tb);
}

struct NiirModArgs {
    Geom g;
    AmGeom a;
    NiirModK<float> k;
    const float *noise;        // [call][2][W]: the (db, dr) noise of niir.py:45-46 / 193-194 per call, or null (noise_level 0)
};

// DEPTH = 1: HueCorrectingNiirModem (niir.py:181-202): a call modulates line - 2 with the previous call's luma, the
// saturation-weighted mean hue of both calls and the previous call's saturation (previous call = neighbouring lane)
template <int DEPTH>
__global__ __launch_bounds__(64, 2) void niir_mod_kernel(const NiirModArgs args) {
    constexpr int kTile = 16;
    __shared__ __attribute__((aligned(16))) float lds_store[64 * kTile + kAmRingFloats];
    lds_float *otile_base = (lds_float *)lds_store;
    lds_float *ring = otile_base + 64 * kTile;
    const Geom &g = args.g;
    const NiirModK<float> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, blockIdx.x, DEPTH, lane);
    const float *rp, *op;
    mod_rows<false>(g, lc, rp, op);
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
    for (int j = 0; j < kAmRing; ++j) ring[j * 64 + lane] = 0.f;
    const float *np = args.noise ? args.noise + 2LL * lc.call * W : nullptr;
    f4 cur[3], nxt[3], nz[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int p = 0; p < 3; ++p) nxt[p] = load_luma<false>(rp + p * g.in_plane_stride, 0, true, W);
    for (int tb = 0; tb < T; tb += 4) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            cur[p] = nxt[p];
            nxt[p] = load_luma<false>(rp + p * g.in_plane_stride, tb + 4, true, W);
        }
        if (np) {
            nz[0] = load_luma<false>(np, tb, true, W);
            nz[1] = load_luma<false>(np + W, tb, true, W);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            const float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
            float y = fmaf_(k.e[0][0], r, fmaf_(k.e[0][1], gg, k.e[0][2] * b));
            float db = fmaf_(k.e[1][0], r, fmaf_(k.e[1][1], gg, k.e[1][2] * b));
            float dr = fmaf_(k.e[2][0], r, fmaf_(k.e[2][1], gg, k.e[2][2] * b));
            if (DEPTH) {
                float py = lane_from(idx1, y), pdb = lane_from(idx1, db), pdr = lane_from(idx1, dr);
                if (!have_prev) { py = y; pdb = db; pdr = dr; }         // niir.py:182-186
                float odb, odr;
                niir_hue_correct(db, dr, pdb, pdr, odb, odr, nz[0][s], nz[1][s]);
                y = py;
                db = odb;
                dr = odr;
            } else if (np) {
                niir_add_offset_noise(db, dr, nz[0][s], nz[1][s]);
            } else {
                niir_add_offset(db, dr);
            }
            if (t >= W) db = dr = 0.f;                                    // beyond the row the filter is fed its last sample anyway
            ring[(t & (kAmRing - 1)) * 64 + lane] = y;
            const float y_d = ring[((t - s_c) & (kAmRing - 1)) * 64 + lane];
            const int n = t - s_c;
            int nc = n < 0 ? 0 : (n > W - 1 ? W - 1 : n);
            const f2 cs = ((const_f2 *)args.a.carrier)[nc];
            const float sn = fmaf_(sph, cs.x, cph * cs.y), cn = fmaf_(cph, cs.x, -(sph * cs.y));
            const float c = st.step(k, t, db, dr, alt, sn, cn);
            put_composite<false, kTile>(g, otile_base, op, lane, wpos, n, y_d + c);
        }
    }
}

}  // namespace cm
#endif

// cm_kernels.h - device-side lane driver of the QAM-family demodulators (gfx950).
//
// One 64-lane workgroup (= one wavefront) walks 64 consecutive calls of the flattened call list
// [frame][run][call]; lane j owns call  block * (64 - DEPTH) - DEPTH + j,  i.e. consecutive
// workgroups overlap by DEPTH halo lanes that only feed their base pairs to their neighbours.
//
// Data movement per lane:
//   input   x[t - dl .. +3]   one unaligned global_load_dwordx4 per 4 steps straight from the
//                             lane's own row (64 rows per wave-instruction; the rows are re-used
//                             from L2 for 32 steps; measured cost in profiles/r01_ubench_valu.txt)
//   luma    x_l[n7 .. +3]     one aligned dwordx4 per 4 steps (second visit of the same row,
//                             lat_r samples later; served by L2 / Infinity Cache)
//   output  r, g, b           one ds_write_b32 per plane and step into a [3][64][16+4] LDS
//                             tile; every 16 steps the tile is read back row-wise
//                             (ds_read_b128) and stored as 64-byte row segments, 16 rows per
//                             wave-instruction, so that HBM sees full-width writes.
//   carrier cos/sin(m cps)    wave-uniform: scalar loads (constant address space) into SGPRs
//   neighbours' base pairs    ds_bpermute_b32 (no VALU cycles)
//
// `dl` delays the input stream by 0..3 steps so that the output index n7 = t - lat_r is congruent
// to t modulo 4: tile flushes and luma loads then fall on fixed sub-steps of the 4x unrolled body.
#ifndef CM_KERNELS_H
#define CM_KERNELS_H

#include <type_traits>

#include "cm_stages.h"

namespace cm {

enum { FRONT_QAM = 0, FRONT_PALD = 1 };

struct Geom {
    const float *in;
    float *out;
    const LaneK<float> *lanes;  // [cycle][3][n_lines]
    const float *carrier;       // {C[m], S[m]} interleaved, m < 2W
    long long in_frame_stride, out_frame_stride, out_plane_stride, out_row_stride;
    long long total_calls;      // main pass: n_frames * calls_per_frame; sparse pass: n_frames * runs_per_frame
    int first_frame, cycle, n_lines;
    int W, H;
    int calls_per_frame, calls_run0, runs_per_frame;
    int first_line[2];
    int k0;              // rows mode: index of the first submitted call within its run
    int delay;           // demodulation_delay (frames mode: output row = line - 2 * delay)
    int rows_mode;       // 1: input row i / output row i are the i-th submitted rows of one run
    int luma_prev_bits;  // bit r: regime r takes its luma from the previous call's input row
    int sparse;          // 1: one lane per run, call 0 of each run only (plain first-line pass)
    int skip_first;      // 1: calls with k == 0 are written by the sparse pass, not by this one
};

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f4 const_f4;
typedef const __attribute__((address_space(4))) f2 const_f2;

constexpr int kTile = 16;       // samples per output tile
constexpr int kTileStride = 20; // floats per tile row (16 B aligned, spreads ds_read_b128 over banks)

__device__ __forceinline__ float lane_from(int byte_index, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
}

template <class S, int FRONT, bool BSF, int DEPTH>
struct DemodLane {
    typedef DemodK<float, S> K;
    typedef typename std::conditional<FRONT == FRONT_PALD, PalDFront<float, S>, QamFront<float, S, BSF>>::type Front;
    static constexpr int SP = S::SP;

    Front front;
    DemodBack<float, S, DEPTH> back;
    LaneK<float> lk;
    float xw[14];
    float ew[FRONT == FRONT_PALD ? 14 : 1];
    float uw[SP + 4], vw[SP + 4];
    f4 lw;
    const float *xp, *lp;
    int idx1, idx2;

    __device__ __forceinline__ f4 load_x(const Geom &g, int first, bool edge) const {
        // x[first .. first + 3], zero outside [0, W)
        if (!edge || (first >= 0 && first + 3 < g.W)) {
            f4u v = *(const f4u *)(xp + first);
            return f4{v.x, v.y, v.z, v.w};
        }
        f4 r = {0.f, 0.f, 0.f, 0.f};
        if (first + 3 >= 0 && first < g.W) {
            if (first >= 0 && first < g.W) r.x = xp[first];
            if (first + 1 >= 0 && first + 1 < g.W) r.y = xp[first + 1];
            if (first + 2 >= 0 && first + 2 < g.W) r.z = xp[first + 2];
            if (first + 3 >= 0 && first + 3 < g.W) r.w = xp[first + 3];
        }
        return r;
    }
    __device__ __forceinline__ f4 load_luma(const Geom &g, int first, bool edge) const {
        if (BSF) return f4{0.f, 0.f, 0.f, 0.f};
        if (!edge || (first >= 0 && first < g.W)) return *(const f4 *)(lp + first);  // W % 4 == 0: all or nothing
        return f4{0.f, 0.f, 0.f, 0.f};
    }

    template <int SUB, bool EDGE>
    __device__ __forceinline__ void substep(const Geom &g, const K &k, int tau, int lat_front, int lat_luma, float *tile,
                                            float *yring, int lane) {
        const int W = g.W;
        const_f4 *car4 = (const_f4 *)g.carrier;
        float luma_bsf = 0.f;
        Pair<float> base;
        if constexpr (FRONT == FRONT_PALD) {
            int n4 = tau - (10 + k.q_e + 9 + 10);
            if (EDGE) n4 = n4 < 0 ? 0 : (n4 > W - 1 ? W - 1 : n4);
            f4 c = car4[n4];
            float car[4] = {c.x, c.y, c.z, c.w};
            float e_out;
            base = front.template step<EDGE>(k, lk, tau, xw[10 + SUB], xw[SUB], ew[FRONT == FRONT_PALD ? SUB : 0], car, e_out);
            ew[FRONT == FRONT_PALD ? 10 + SUB : 0] = e_out;
        } else {
            int n2 = tau - (10 + k.q_e);
            if (EDGE) n2 = n2 < 0 ? 0 : (n2 > W - 1 ? W - 1 : n2);
            f4 c = car4[n2];
            float car[4] = {c.x, c.y, c.z, c.w};
            base = front.template step<EDGE>(k, lk, tau, xw[10 + SUB], xw[SUB], car, luma_bsf);
        }
        const int n6 = tau - lat_front, n7 = n6 - SP;
        float y_src;
        if (BSF) {
            const int nl = tau - lat_luma;
            yring[(nl & 15) * 64 + lane] = luma_bsf;
            y_src = yring[(n7 & 15) * 64 + lane];
        } else {
            y_src = SUB == 0 ? lw.x : (SUB == 1 ? lw.y : (SUB == 2 ? lw.z : lw.w));
        }
        Pair<float> b1 = {0.f, 0.f}, b2 = {0.f, 0.f};
        if (DEPTH >= 1) { b1.s = lane_from(idx1, base.s); b1.c = lane_from(idx1, base.c); }
        if (DEPTH >= 2) { b2.s = lane_from(idx2, base.s); b2.c = lane_from(idx2, base.c); }
        float u, v;
        back.combine(lk, base, b1, b2, u, v);
        uw[SP + SUB] = u;
        vw[SP + SUB] = v;
        int n7c = n7;
        if (EDGE) n7c = n7 < 0 ? 0 : (n7 > W - 1 ? W - 1 : n7);
        const_f2 *car2 = (const_f2 *)g.carrier;
        f2 cc = car2[2 * n7c];
        float carb[2] = {cc.x, cc.y};
        Rgb<float> o = back.template step<EDGE>(k, lk, n6, u, v, uw[SUB], vw[SUB], y_src, carb);
        if (!EDGE || (n7 >= 0 && n7 < W)) {
            float *tp = tile + lane * kTileStride + (n7 & (kTile - 1));
            tp[0] = o.r;
            tp[64 * kTileStride] = o.g;
            tp[2 * 64 * kTileStride] = o.b;
        }
    }
};

// Row-wise read-back of the LDS tile and coalesced store: 16 rows x 64 B per wave-instruction.
__device__ __forceinline__ void flush_tile(const Geom &g, const float *tile, float *const *optr, int first_col, int lane) {
    __builtin_amdgcn_wave_barrier();
    const int chunk = lane & 3;
    const int col = first_col + 4 * chunk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 2) + 16 * q;
        float *dst = optr[row];
        if (dst != nullptr && col < g.W) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                f4 v = *(const f4 *)(tile + (p * 64 + row) * kTileStride + 4 * chunk);
                *(f4 *)(dst + p * g.out_plane_stride + col) = v;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <class S, int FRONT, bool BSF, int DEPTH>
__global__ __launch_bounds__(64, 2) void demod_kernel(const Geom g, const DemodK<float, S> k) {
    typedef DemodLane<S, FRONT, BSF, DEPTH> Lane;
    __shared__ __attribute__((aligned(16))) float tile[3 * 64 * kTileStride];
    __shared__ float *optr[64];
    __shared__ float yring[BSF ? 16 * 64 : 1];

    const int lane = threadIdx.x;
    // ---- which call does this lane own -------------------------------------------------------
    long long c;
    bool active;
    long long frame;
    int run, i;
    if (g.sparse) {
        c = (long long)blockIdx.x * 64 + lane;
        active = c < g.total_calls;
        if (!active) c = g.total_calls - 1;
        frame = c / g.runs_per_frame;
        run = (int)(c - frame * g.runs_per_frame);
        i = 0;
    } else {
        c = (long long)blockIdx.x * (64 - DEPTH) - DEPTH + lane;
        active = lane >= DEPTH && c < g.total_calls;
        if (c < 0) c = 0;
        if (c >= g.total_calls) c = g.total_calls - 1;
        frame = c / g.calls_per_frame;
        int rem = (int)(c - frame * g.calls_per_frame);
        run = rem >= g.calls_run0 ? 1 : 0;
        i = rem - (run ? g.calls_run0 : 0);
    }
    const int line = g.first_line[run] + 2 * i;
    const int kk = g.k0 + i;
    const int regime = kk < 2 ? kk : 2;
    const int dy = (g.luma_prev_bits >> regime) & 1;
    int src_row, luma_row, out_row;
    bool store_ok = active && !(g.skip_first && kk == 0);
    if (g.rows_mode) {
        src_row = i;
        luma_row = i - dy < 0 ? i : i - dy;
        out_row = i;
    } else {
        src_row = line;
        if (src_row >= g.H) src_row -= 2 * ((src_row - g.H) / 2 + 1);  // image.py:80-81: step back by 2 until inside
        luma_row = line - 2 * dy;
        if (luma_row < 0) luma_row = src_row;
        if (luma_row >= g.H) luma_row -= 2 * ((luma_row - g.H) / 2 + 1);
        out_row = line - 2 * g.delay;
        store_ok = store_ok && i >= g.delay && out_row >= 0 && out_row < g.H;
    }
    Lane L;
    L.xp = g.in + frame * g.in_frame_stride + (long long)src_row * g.W;
    L.lp = g.in + frame * g.in_frame_stride + (long long)luma_row * g.W;
    float *op = g.out + frame * g.out_frame_stride + (long long)out_row * g.out_row_stride;
    optr[lane] = store_ok ? op : nullptr;
    {
        int fmod = (int)((g.first_frame + frame) % g.cycle);
        L.lk = g.lanes[((long long)fmod * 3 + regime) * g.n_lines + line];
    }
    L.idx1 = ((lane + 63) & 63) * 4;
    L.idx2 = ((lane + 62) & 63) * 4;
    L.front.reset();
    L.back.reset();
#pragma unroll
    for (int j = 0; j < 14; ++j) L.xw[j] = 0.f;
#pragma unroll
    for (int j = 0; j < (FRONT == FRONT_PALD ? 14 : 1); ++j) L.ew[j] = 0.f;
#pragma unroll
    for (int j = 0; j < S::SP + 4; ++j) L.uw[j] = L.vw[j] = 0.f;
    if (BSF) {
        for (int j = 0; j < 16; ++j) yring[j * 64 + lane] = 0.f;
    }

    // ---- stream geometry ----------------------------------------------------------------------
    const int lat_front = Lane::Front::latency(k);
    int lat_luma = 0;
    if constexpr (FRONT == FRONT_QAM) lat_luma = Lane::Front::luma_latency(k);
    const int lat_total = lat_front + S::SP;
    const int lat_r = (lat_total + 3) & ~3;
    const int dl = lat_r - lat_total;          // input delay, 0..3
    const int W = g.W;
    const int T = W + lat_r;                   // multiple of 4
    int t_mid0 = lat_r;                        // first body whose every stage index is >= 0
    int t_mid1 = (W + dl - 3) & ~3;            // bodies below this never touch the end of the row
    if (t_mid1 < t_mid0) t_mid1 = t_mid0;

    f4 nx = L.load_x(g, -dl, true);
    f4 nl = L.load_luma(g, -lat_r, true);

    auto body = [&](int tb, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        L.xw[10] = nx.x; L.xw[11] = nx.y; L.xw[12] = nx.z; L.xw[13] = nx.w;
        L.lw = nl;
        // prefetch the next body's input (a whole body of arithmetic hides the latency)
        nx = L.load_x(g, tb + 4 - dl, EDGE || tb + 4 >= t_mid1);
        nl = L.load_luma(g, tb + 4 - lat_r, EDGE || tb + 4 >= t_mid1);
        const int tau = tb - dl;
        L.template substep<0, EDGE>(g, k, tau + 0, lat_front, lat_luma, tile, yring, lane);
        L.template substep<1, EDGE>(g, k, tau + 1, lat_front, lat_luma, tile, yring, lane);
        L.template substep<2, EDGE>(g, k, tau + 2, lat_front, lat_luma, tile, yring, lane);
        L.template substep<3, EDGE>(g, k, tau + 3, lat_front, lat_luma, tile, yring, lane);
#pragma unroll
        for (int j = 0; j < 10; ++j) L.xw[j] = L.xw[j + 4];
        if (FRONT == FRONT_PALD) {
#pragma unroll
            for (int j = 0; j < 10; ++j) L.ew[FRONT == FRONT_PALD ? j : 0] = L.ew[FRONT == FRONT_PALD ? j + 4 : 0];
        }
#pragma unroll
        for (int j = 0; j < S::SP; ++j) { L.uw[j] = L.uw[j + 4]; L.vw[j] = L.vw[j + 4]; }
        const int n7_last = tb + 3 - lat_r;    // congruent to 3 modulo 4
        if (n7_last >= 0 && ((n7_last & (kTile - 1)) == kTile - 1 || n7_last == W - 1))
            flush_tile(g, tile, optr, n7_last & ~(kTile - 1), lane);
    };
    int tb = 0;
    for (; tb < t_mid0; tb += 4) body(tb, std::true_type());
    for (; tb < t_mid1; tb += 4) body(tb, std::false_type());
    for (; tb < T; tb += 4) body(tb, std::true_type());
}

}  // namespace cm
#endif

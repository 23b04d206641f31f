// cm_secam_kernels.h - device-side lane drivers of the SECAM decoder and encoder (gfx950).
//
// Same execution model and data movement as cm_kernels.h (one lane = one scan line, input rows
// staged through a 64 x 32-sample LDS tile filled by global_load_lds, outputs transposed through an
// LDS tile into 64-byte row segments).  Differences:
//   * the chroma stream starts with the mirrored pre-roll cc[m] = x[P - m], m < P (secam.py:283-284),
//     read sample by sample out of the first input tile before the aligned main loop starts;
//   * luma (band-stop at 1x rate) runs on a second, delayed visit of the row so that it meets the
//     FM-decoded colour-difference sample of the same pixel without a delay line;
//   * the previous line's colour-difference signal comes from the neighbouring lane.
#ifndef CM_SECAM_KERNELS_H
#define CM_SECAM_KERNELS_H

#include "cm_kernels.h"
#include "cm_mod_kernels.h"

namespace cm {

struct SecamDemodArgs {
    Geom g;                    // g.lanes -> SecamDemodLaneK<float> table, g.carrier4 -> FM reference {cos, sin} pairs,
                               // g.carrier2 -> dc table (cm_plan.h: build_fm_dc)
    SecamDemodK<float> k;
    SecamBp64 e64;             // band-pass + bell of the guarded bodies in float64 (cm_stages.h)
};

// U8: the ImageModem byte boundary fused in, as in the PAL / NTSC decoders (cm_kernels.h: PassCfg::U8): composite bytes
// enter through (5 (byte / 255) - 1) / 3, interleaved RGB bytes leave; the tiles hold bytes and the strides count bytes.
// Occupancy knobs of the one-wave decoder: CM_SECAM_WAVES waves per SIMD the register allocation aims at (3 needs <= 168
// VGPRs: 2 spilled values) and CM_SECAM_TILE samples per float input tile row (8: 32-byte row segments, 2 KiB instead of 8,
// so that 11 workgroups instead of 8 fit the LDS of a CU).  Measured (profiles/r02_secam_notes.md): 3 / 8 runs 3.66 ms per
// 1000 frames against 3.45 ms for 2 / 32 - more resident waves do not pay on this kernel either; 2 / 32 stays.
#ifndef CM_SECAM_WAVES
#define CM_SECAM_WAVES 2
#endif
#ifndef CM_SECAM_TILE
#define CM_SECAM_TILE 32
#endif
// 1: the interior bodies run their four steps stage by stage and take the eight phase steps together - one wave-uniform
// branch to the library atan2f per body instead of one per phase step
#ifndef CM_SECAM_BATCH_ANGLES
#define CM_SECAM_BATCH_ANGLES 1
#endif
// samples per row of the float output tile (16: 64-byte row segments, 12 KiB; 8: 32-byte segments, 6 KiB)
#ifndef CM_SECAM_OUT_TILE
#define CM_SECAM_OUT_TILE 16
#endif
// samples per input tile row of the float32 wave pair
#ifndef CM_SECAM_PAIR_IN_TILE
#define CM_SECAM_PAIR_IN_TILE 8
#endif
// bodies (of 4 steps) the luma samples of the second row visit are asked for ahead of their use
#ifndef CM_SECAM_LUMA_AHEAD
#define CM_SECAM_LUMA_AHEAD 1
#endif
template <bool U8>
__global__ __launch_bounds__(64, CM_SECAM_WAVES) void secam_demod_kernel(const SecamDemodArgs args) {
    constexpr int kTile = U8 ? 16 : CM_SECAM_OUT_TILE, DEPTH = 1;
    constexpr int kIT = U8 ? kInTile : CM_SECAM_TILE;      // samples per input tile row (byte tiles stay 32 wide)
    constexpr int kIn = 64 * kIT;                           // floats (U8: bytes)
    __shared__ __attribute__((aligned(16))) float lds_store[U8 ? (kIn + 3 * 64 * kTile) / 4 : kIn + 3 * 64 * kTile];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + (U8 ? kIn / 4 : kIn);
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const Geom &g = args.g;
    SecamDemodK<float> k = args.k;
    typedef SecamDemodPk::VP VP;
    if (VP::VT) pin_block(k.taps);
    if (VP::VB) pin_block(k.bpf, false);
    SecamDemodKPk kp;      // the quadrature low-pass runs on the pair (I, Q) (cm_stages_pk.h)
    kp.load(k);
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const float *xp, *op;
    if (U8) {
        xp = (const float *)((const unsigned char *)g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.W);
        op = lc.store_ok ? (const float *)((unsigned char *)g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride)
                         : nullptr;
    } else {
        xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
        op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    }
    SecamDemodLaneK<float> lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamDemodLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    SecamDemodPk st;
    st.reset();
    float chw[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) chw[j] = 0.f;
    lds_float *otile = U8 ? (lds_float *)((lds_u8 *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const lds_float *xrow = itile + lane * kIT;
    const lds_u8 *xrow8 = (const lds_u8 *)itile + lane * kIT;

    const int W = g.W, P = k.preroll, Lc = W + P;
    const int lat = SecamDemod<float>::latency(k);          // chroma sample n = m - lat (the packed form keeps the schedule)
    const int d_luma = lat + 1 - P - k.s_y;                 // luma filter input x[xi - d_luma] at main-loop sample xi
    const int lat_out = d_luma + k.s_y;                     // output sample n' = xi - lat_out
    const int s_flush = (lat_out + 3) & 3;
    float own_prev = 0.f, nb_prev = 0.f;

    // one step of the stream: cc = cc[m]; x_l = x[n' + s_y] for the luma filter.  EDGE = false: interior of the row - no
    // stage touches a boundary, the table indices need no clamp, the output sample lies inside the row
    auto step = [&](auto edge_tag, int m, float cc, float x_l, int sub) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        int m2 = m - k.s_b - 10;
        if (EDGE) m2 = m2 < 0 ? 0 : (m2 > Lc - 1 ? Lc - 1 : m2);
        f4 c = ((const_f4 *)g.carrier4)[m2];
        int m4 = m - lat + P;                               // row-stream sample the decimator completes in this step
        if (EDGE) m4 = m4 < 0 ? 0 : (m4 > Lc - 1 ? Lc - 1 : m4);
        const float dc = ((const __attribute__((address_space(4))) float *)g.carrier2)[m4];
        float ch_out;
        float own = st.template chroma_step<EDGE>(k, kp, lk, m, cc, chw[sub], pf2{c.x, c.y}, pf2{c.z, c.w}, dc, ch_out, args.e64);
        chw[10 + sub] = ch_out;
        const int n = m - 1 - lat;                          // the back end runs one sample behind the exchange
        float luma = st.template luma_step<EDGE>(k, n, x_l);
        Rgb<float> o = st.finish(k, lk, luma, own_prev, nb_prev);
        own_prev = own;
        nb_prev = lane_from(idx1, own);
        if (!EDGE || (n >= 0 && n < W)) put_rgb<U8, kTile>(otile, wpos, n, o);
    };
    auto shift_window = [&]() {
#pragma unroll
        for (int j = 0; j < 10; ++j) chw[j] = chw[j + 4];
    };

    if (U8) fill_tile_u8(g, itile, xp, 0, lane); else fill_tile<kIT>(g, itile, xp, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- pre-roll: cc[m] = x[P - m] for m < P, in bodies of 4 steps so that the windows keep their phase
    const int m_start = -((4 - (P & 3)) & 3);
    for (int mb = m_start; mb < P; mb += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = mb + s;
            float cc = 0.f;
            if (m >= 0) {
                int xi = P - m;
                if (xi > W - 1) xi = W - 1;
                if (P < kIT)   // the mirrored samples lie in the first input tile
                    cc = U8 ? __builtin_fmaf((float)xrow8[xi], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : xrow[xi];
                else               // long pre-rolls (sampling rates above ~27 MHz): straight from the row, once per line
                    cc = U8 ? __builtin_fmaf((float)((const unsigned char *)xp)[xi], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : xp[xi];
            }
            step(std::true_type(), m, cc, 0.f, s);
        }
        shift_window();
    }
    // ---- main loop over the row samples xi = m - P
    auto read_x = [&](int first) -> f4 {
        f4 v;
        if (U8) v = decode_bytes(*(const lds_u32 *)(xrow8 + (first & (kIT - 1))));
        else v = *(const lds_f4 *)(xrow + (first & (kIT - 1)));
        if (first + 3 >= W) {
            if (first >= W) v.x = 0.f;
            if (first + 1 >= W) v.y = 0.f;
            if (first + 2 >= W) v.z = 0.f;
            if (first + 3 >= W) v.w = 0.f;
        }
        return v;
    };
    auto read_luma = [&](int first) -> f4 { return load_luma<U8>(xp, first, true, W); };
    const int T = (g.Wp + lat_out + 3) & ~3;
    f4 xv = read_x(0);
    f4 nl = read_luma(-d_luma);
#if CM_SECAM_LUMA_AHEAD == 2
    f4 nl2 = read_luma(4 - d_luma);
#endif
    // Interior bodies (cm_stages.h: secam_mid_bounds): no stage index touches a row boundary and the band-pass runs in float32
    int xb_mid0, xb_mid1;
    secam_mid_bounds(W, P, lat, args.e64, xb_mid0, xb_mid1);
    auto body = [&](auto edge_tag, int xb) __attribute__((always_inline)) {   // (not inlined, the line's state would live in scratch memory)
        constexpr bool EDGE = decltype(edge_tag)::value;
        const f4 lw = nl;
#if CM_SECAM_LUMA_AHEAD == 2
        nl = nl2;
        nl2 = read_luma(xb + 8 - d_luma);
#else
        nl = read_luma(xb + 4 - d_luma);
#endif
        auto flush_after = [&](int s) {
            if (s == s_flush) {
                const int n = xb + s - lat_out;
                if (n >= 0 && ((n & (kTile - 1)) == kTile - 1 || n == g.Wp - 1)) {
                    if (U8) flush_tile_u8(g, otile_base, op, n & ~(kTile - 1), lane);
                    else flush_tile<kTile>(g, otile_base, op, n & ~(kTile - 1), lane);
                }
            }
        };
        if (EDGE || !CM_SECAM_BATCH_ANGLES) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                step(edge_tag, P + xb + s, xv[s], lw[s], s);
                flush_after(s);
            }
        } else {
            // interior: the four steps stage by stage, so that their eight phase steps share one branch to the library atan2f
            const int m0 = P + xb;
            pf2 y0[4], y1[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f4 c = ((const_f4 *)g.carrier4)[m0 + s - k.s_b - 10];
                float ch_out;
                st.chroma_front_mid(k, kp, xv[s], chw[s], pf2{c.x, c.y}, pf2{c.z, c.w}, ch_out, y0[s], y1[s]);
                chw[10 + s] = ch_out;
            }
            PhaseStep pe[4], po[4];
            bool all_small = true;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const pf2 prev = s == 0 ? st.iq_prev : y1[s - 1];
                pe[s].set(prev.x, prev.y, y0[s].x, y0[s].y);
                po[s].set(y0[s].x, y0[s].y, y1[s].x, y1[s].y);
                all_small = all_small && pe[s].small() && po[s].small();
            }
            st.iq_prev = y1[3];
            st.have_prev = 1;
            float d_e[4], d_o[4];
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!all_small) == 0ull, 1)) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { d_e[s] = pe[s].series(); d_o[s] = po[s].series(); }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) { d_e[s] = pe[s].full(); d_o[s] = po[s].full(); }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = m0 + s;
                const float dc = ((const __attribute__((address_space(4))) float *)g.carrier2)[m - lat + P];
                const float own = st.chroma_back_mid(k, lk, d_e[s], d_o[s], dc);
                const int n = m - 1 - lat;
                const float luma = st.template luma_step<false>(k, n, lw[s]);
                const Rgb<float> o = st.finish(k, lk, luma, own_prev, nb_prev);
                own_prev = own;
                nb_prev = lane_from(idx1, own);
                put_rgb<U8, kTile>(otile, wpos, n, o);
                flush_after(s);
            }
        }
        shift_window();
        const int nxt = xb + 4;
        if ((nxt & (kIT - 1)) == 0 && nxt < W) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        xv = read_x(nxt);
        if ((nxt & (kIT - 1)) == kIT - 4 && nxt + 4 < W) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (U8) fill_tile_u8(g, itile, xp, nxt / kIT + 1, lane); else fill_tile<kIT>(g, itile, xp, nxt / kIT + 1, lane);
        }
    };
    int xb = 0;
    for (; xb < xb_mid0; xb += 4) body(std::true_type(), xb);
    st.to32();
    for (; xb < xb_mid1; xb += 4) body(std::false_type(), xb);
    st.to64();
    for (; xb < T; xb += 4) body(std::true_type(), xb);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-pair form of the decoder (the structure of cm_kernels.h: run_pair): two wavefronts walk the same 64 calls.
//   wave 0 (stage A)  every global load (the input tile, 16 samples per row), the mirrored pre-roll, band-pass + bell,
//                     up2, quadrature mix and the packed (I, Q) low-pass; per body of 4 steps it leaves the low-passed pairs
//                     (I0, Q0, I1, Q1) of each lane in a double-buffered LDS ring and the 4 row samples the luma filter will
//                     want d_luma steps later in a delay ring (no second visit of the row)
//   wave 1 (stage B)  phase steps, decimator, clip, de-emphasis, the luma band-stop on the delayed samples, the other
//                     colour-difference signal from the neighbouring lane, matrix, output tile and every global store
// One s_barrier per body: A after writing block b, B before reading it.
// ---------------------------------------------------------------------------------------------------------------------
// Round 1's form (guards in every step, 32 - 38 KiB of LDS) ran 4.0 - 4.1 ms per 1000 frames against 3.5 for the one-wave
// kernel.  With edge-free interior bodies in both stages, stage A at 165 VGPRs and 26 KiB of LDS (8-sample input tile, four
// bodies of the luma delay in stage B's registers: CM_SECAM_PAIR_REG_DELAY) six workgroups fit a CU and it runs 2.78 - 2.87
// against 2.93 - 2.99 (profiles/r02_notes.md section 3): the default for float rows where the luma delay allows;
// -DCM_SECAM_PAIR=0 brings the one-wave kernel back.  Byte rows: 2.75 - 2.78 against 2.85 - 2.90 ms (CM_SECAM_PAIR_U8).
#ifndef CM_SECAM_PAIR
#define CM_SECAM_PAIR 1
#endif
// 1: both stages run the interior bodies of a row without guards
// 1: byte rows on the wave pair as well
#ifndef CM_SECAM_PAIR_U8
#define CM_SECAM_PAIR_U8 1
#endif
#ifndef CM_SECAM_PAIR_MID
#define CM_SECAM_PAIR_MID 1
#endif
// 1: the shapes whose float32 margin is thin (cm_api.hip: create_secam) run on the wave pair with stage A in float64
#ifndef CM_SECAM_F64
#define CM_SECAM_F64 1
#endif
// bodies of luma delay stage B keeps in registers (a queue of float4) instead of the LDS delay ring: 1 KiB less LDS each
#ifndef CM_SECAM_PAIR_REG_DELAY
#define CM_SECAM_PAIR_REG_DELAY 4
#endif
#ifndef CM_SECAM_PAIR_LDS_PAD      /* occupancy experiments: unused floats of LDS per workgroup */
#define CM_SECAM_PAIR_LDS_PAD 0
#endif
// hand-over buffers of the wave pair: 2 = stage A fills one while B reads the other (one barrier per body); 1 = a second
// barrier per body (B has read) instead of the second buffer: 4 KiB less LDS
#ifndef CM_SECAM_PAIR_MID_BUFS
#define CM_SECAM_PAIR_MID_BUFS 2
#endif
constexpr int kSecamMid = CM_SECAM_PAIR_MID_BUFS * 4 * 256;        // floats: [buffer][I0 | Q0 | I1 | Q1][lane][4 steps]
constexpr int kSecamPairMaxLumaDelay = 4 + 4 * 14 + 3;   // delay ring of at most 16 blocks of [lane][4 samples]
template <bool U8> constexpr int secam_pair_lds_floats(int d_luma) {
    return (U8 ? 64 * kInTile / 4 : 64 * CM_SECAM_PAIR_IN_TILE) + kSecamMid + (U8 ? 64 * 3 * 16 / 4 : 3 * 64 * CM_SECAM_OUT_TILE) + (((d_luma - 4) >> 2) - CM_SECAM_PAIR_REG_DELAY + 2) * 256 + CM_SECAM_PAIR_LDS_PAD;
}

struct SecamDemodArgs64 {
    SecamDemodArgs a;
    SecamDemodK<double> k64;       // the constants of stage A in float64
    const double *fm_ref64;        // FM reference {cos, sin} pairs in float64 (same layout as g.carrier4)
};

// F64: stage A in float64 (SecamDemodA64); the launch passes SecamDemodArgs64
template <bool U8, bool F64, class Args>
__device__ __forceinline__ void secam_demod_pair_body(const Args &args_in);

// waves per SIMD the register allocation of the float32 pair aims at (3: <= 168 VGPRs)
#ifndef CM_SECAM_PAIR_WAVES
#define CM_SECAM_PAIR_WAVES 3
#endif
template <bool U8>
__global__ __launch_bounds__(128, CM_SECAM_PAIR_WAVES) void secam_demod_pair_kernel(const SecamDemodArgs args) {
    secam_demod_pair_body<U8, false>(args);
}
template <bool U8>
__global__ __launch_bounds__(128, 2) void secam_demod_pair64_kernel(const SecamDemodArgs64 args) {
    secam_demod_pair_body<U8, true>(args);
}

template <bool F64> struct SecamPairArgs;
template <> struct SecamPairArgs<false> {
    static __device__ __forceinline__ const SecamDemodArgs &base(const SecamDemodArgs &a) { return a; }
};
template <> struct SecamPairArgs<true> {
    static __device__ __forceinline__ const SecamDemodArgs &base(const SecamDemodArgs64 &a) { return a.a; }
};

template <bool U8, bool F64, class Args>
__device__ __forceinline__ void secam_demod_pair_body(const Args &args_in) {
    const SecamDemodArgs &args = SecamPairArgs<F64>::base(args_in);
    constexpr int kTile = U8 ? 16 : CM_SECAM_OUT_TILE, DEPTH = 1;
    constexpr int kIT = U8 ? kInTile : CM_SECAM_PAIR_IN_TILE;                // samples per input tile row
    constexpr int kIn = U8 ? 64 * kInTile / 4 : 64 * kIT, kOut = U8 ? 64 * 3 * kTile / 4 : 3 * 64 * kTile;   // floats
    extern __shared__ __attribute__((aligned(16))) float secam_pair_lds[];   // kIn + kSecamMid + kOut + blocks * 256 floats
    lds_float *itile = (lds_float *)secam_pair_lds;
    lds_float *ring = itile + kIn;
    lds_float *otile_base = ring + kSecamMid;
    lds_float *xring = otile_base + kOut;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const Geom &g = args.g;
    SecamDemodK<float> k = args.k;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);

    // ---- stream geometry (identical in both waves) ------------------------------------------------
    const int W = g.W, P = k.preroll, Lc = W + P;
    const int lat = SecamDemod<float>::latency(k);
    const int d_luma = lat + 1 - P - k.s_y;                 // luma filter input x[xi - d_luma] at main-loop sample xi
    const int lat_out = d_luma + k.s_y;                     // output sample n' = xi - lat_out
    const int s_flush = (lat_out + 3) & 3;
    const int m_start = -((4 - (P & 3)) & 3);               // the pre-roll in bodies of 4 steps
    const int n_pre = (P - m_start) >> 2;
    const int T = (g.Wp + lat_out + 3) & ~3;
    const int n_bodies = n_pre + (T >> 2);
    // A writes x[xb - 4 - lr_o .. + 3] (out of the samples of its last two bodies); B reads it lr_m bodies later
    const int lr_o = (d_luma - 4) & 3, lr_m = ((d_luma - 4) >> 2) - CM_SECAM_PAIR_REG_DELAY;   // (the host checks lr_m >= 0)
    const int n_blocks = lr_m + 2;                          // delay ring blocks (the host sizes the LDS with the same number)
    // interior bodies (cm_stages.h: secam_mid_bounds): every stage index of the four steps lies strictly inside its stream;
    // stage A of the float32 pair also keeps the float64 head and tail of its band-pass out of them
    int xb_mid0, xb_mid1, xb_a0, xb_a1;
    {
        SecamBp64 none;
        none.head = none.tail = 0;
        secam_mid_bounds(W, P, lat, none, xb_mid0, xb_mid1);
        if (F64) { xb_a0 = xb_mid0; xb_a1 = xb_mid1; }
        else secam_mid_bounds(W, P, lat, args.e64, xb_a0, xb_a1);
    }
    const int b_mid0 = n_pre + (xb_mid0 >> 2), b_mid1 = n_pre + (xb_mid1 >> 2);
    const int b_a0 = CM_SECAM_PAIR_MID ? n_pre + (xb_a0 >> 2) : 0, b_a1 = CM_SECAM_PAIR_MID ? n_pre + (xb_a1 >> 2) : 0;

    if (role == 0) {
        // =================================== stage A ===========================================
        typedef SecamDemodPkA::VP VP;
        if (VP::VT) pin_block(k.taps);
        if (VP::VB) pin_block(k.bpf, false);
        SecamDemodKPk kp;
        kp.load(k);
        const float *xp;
        if (U8) xp = (const float *)((const unsigned char *)g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.W);
        else xp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
        typename std::conditional<F64, SecamDemodA64, SecamDemodPkA>::type st;
        st.reset();
        typename std::conditional<F64, double, float>::type chw[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) chw[j] = 0;
        const lds_float *xrow = itile + lane * kIT;
        const lds_u8 *xrow8 = (const lds_u8 *)itile + lane * kInTile;
        for (int j = 0; j < n_blocks; ++j) *(lds_f4 *)(xring + j * 256 + lane * 4) = f4{0.f, 0.f, 0.f, 0.f};
        // the mirrored pre-roll wants x[1 .. P]: float rows get a 32-sample copy of the row start in the (still unused)
        // output tile area, byte tiles are 32 samples wide anyway
        const lds_float *pre_row = otile_base + lane * kInTile;
        if (U8) fill_tile_u8(g, itile, xp, 0, lane);
        else {
            fill_tile<kIT>(g, itile, xp, 0, lane);
            if (P < kInTile && kTile >= 16) fill_tile<kInTile>(g, otile_base, xp, 0, lane);   // (3 planes x 16 samples hold the 32)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        auto read_x = [&](int first, bool inside = false) -> f4 {   // inside: first + 3 < W is the caller's knowledge (the interior bodies)
            f4 v;
            if (U8) v = decode_bytes(*(const lds_u32 *)(xrow8 + (first & (kInTile - 1))));
            else v = *(const lds_f4 *)(xrow + (first & (kIT - 1)));
            if (!inside && first + 3 >= W) {
                if (first >= W) v.x = 0.f;
                if (first + 1 >= W) v.y = 0.f;
                if (first + 2 >= W) v.z = 0.f;
                if (first + 3 >= W) v.w = 0.f;
            }
            return v;
        };
        f4 xv = {0.f, 0.f, 0.f, 0.f}, xprev = xv, xprev2 = xv;
        int xw = 0;      // delay ring block of this body
        auto body_a = [&](auto mid_tag, int b) __attribute__((always_inline)) {
            constexpr bool mid = decltype(mid_tag)::value;
            const bool pre = b < n_pre;
            const int xb = (b - n_pre) << 2;
            if (b == n_pre) xv = read_x(0);
            pf2 y0[4], y1[4];      // the low-passed (I, Q) of the steps' two 2x-rate samples
            if constexpr (F64) {
                if (mid) {
                    const double *cp = args_in.fm_ref64 + 4 * (long long)(m_start + 4 * b - k.s_b - 10);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const double car[4] = {cp[4 * s], cp[4 * s + 1], cp[4 * s + 2], cp[4 * s + 3]};
                        double ch_out;
                        st.step_mid(args_in.k64, (double)xv[s], chw[s], car, ch_out, y0[s], y1[s]);
                        chw[10 + s] = ch_out;
                    }
                }
            } else {
                if (mid) {
                    const const_f4 *cp = (const_f4 *)g.carrier4 + (m_start + 4 * b - k.s_b - 10);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const f4 c = cp[s];
                        float ch_out;
                        st.step_mid(k, kp, xv[s], chw[s], pf2{c.x, c.y}, pf2{c.z, c.w}, ch_out, y0[s], y1[s]);
                        chw[10 + s] = ch_out;
                    }
                }
            }
            if (!mid)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = m_start + 4 * b + s;
                float cc = 0.f;
                if (pre) {
                    if (m >= 0) {      // cc[m] = x[P - m] (secam.py:283-284)
                        int xi = P - m;
                        if (xi > W - 1) xi = W - 1;
                        if (P < kInTile && (U8 || kTile >= 16))
                            cc = U8 ? __builtin_fmaf((float)xrow8[xi], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : pre_row[xi];
                        else
                            cc = U8 ? __builtin_fmaf((float)((const unsigned char *)xp)[xi], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : xp[xi];
                    }
                } else {
                    cc = xv[s];
                }
                int m2 = m - k.s_b - 10;
                m2 = m2 < 0 ? 0 : (m2 > Lc - 1 ? Lc - 1 : m2);
                if constexpr (F64) {
                    const double *cp = args_in.fm_ref64 + 4 * (long long)m2;
                    const double car[4] = {cp[0], cp[1], cp[2], cp[3]};
                    double ch_out;
                    st.step(args_in.k64, m, (double)cc, chw[s], car, ch_out, y0[s], y1[s]);
                    chw[10 + s] = ch_out;
                } else {
                    const f4 c = ((const_f4 *)g.carrier4)[m2];
                    float ch_out;
                    st.step(k, kp, m, cc, chw[s], pf2{c.x, c.y}, pf2{c.z, c.w}, ch_out, y0[s], y1[s], args.e64);
                    chw[10 + s] = ch_out;
                }
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) chw[j] = chw[j + 4];
            // the row samples the luma filter meets d_luma steps from now: x[xb - lr_o .. + 3] (zeros in the pre-roll)
            f4 blk = {0.f, 0.f, 0.f, 0.f};
            if (!pre) {
                blk = lr_o == 0 ? xprev
                    : lr_o == 1 ? f4{xprev2.w, xprev.x, xprev.y, xprev.z}
                    : lr_o == 2 ? f4{xprev2.z, xprev2.w, xprev.x, xprev.y} : f4{xprev2.y, xprev2.z, xprev2.w, xprev.x};
                xprev2 = xprev;
                xprev = xv;
                const int nxt = xb + 4;
                if ((nxt & (kIT - 1)) == 0 && nxt < W) {   // first read of a new tile: its fill was issued a body ago
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
                xv = read_x(nxt, mid);      // (interior: xb < (W - 8) & ~3, so nxt + 3 < W)
                if ((nxt & (kIT - 1)) == kIT - 4 && nxt + 4 < W) {   // that was the last read of this tile: refill it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (U8) fill_tile_u8(g, itile, xp, nxt / kIT + 1, lane); else fill_tile<kIT>(g, itile, xp, nxt / kIT + 1, lane);
                }
            }
            lds_float *slot = ring + (CM_SECAM_PAIR_MID_BUFS == 2 ? (b & 1) * (kSecamMid / 2) : 0) + lane * 4;
            if (CM_SECAM_PAIR_MID_BUFS == 1 && b > 0) asm volatile("s_barrier" ::: "memory");      // B has read body b - 1
            // the body's eight 2x-rate samples z[0 .. 7] = (y0[0], y1[0], y0[1], ...) as the pairs (z[j], z[j + 4]) stage B's packed phase
            // steps take: I as (z0, z4, z1, z5 | z2, z6, z3, z7), Q likewise
            // (real parts | imaginary parts of two samples: one v_pk_mov_b32 each)
            auto put_iq = [&](lds_float *at, pf2 a, pf2 b, pf2 c, pf2 d) __attribute__((always_inline)) {
                const pf2 i_ab = pk_lolo(a, b), i_cd = pk_lolo(c, d), q_ab = pk_hihi(a, b), q_cd = pk_hihi(c, d);
                *(lds_f4 *)at = f4{i_ab.x, i_ab.y, i_cd.x, i_cd.y};
                *(lds_f4 *)(at + 512) = f4{q_ab.x, q_ab.y, q_cd.x, q_cd.y};
            };
            if (mid && k.odd_l) {
                // odd low-pass shift: the stream one 2x-rate sample later - z'[0] = the sample held from the body before, z'[j] = z[j - 1]
                // (the guarded steps do this themselves, pair by pair: SecamDemodPkA::step)
                const pf2 h = st.hold();
                st.set_hold(y1[3]);
                put_iq(slot, h, y1[1], y0[0], y0[2]);               // (z'0, z'4, z'1, z'5) = (h, z3, z0, z4)
                put_iq(slot + 256, y1[0], y1[2], y0[1], y0[3]);     // (z'2, z'6, z'3, z'7) = (z1, z5, z2, z6)
            } else {
                put_iq(slot, y0[0], y0[2], y1[0], y1[2]);           // (z0, z4, z1, z5)
                put_iq(slot + 256, y0[1], y0[3], y1[1], y1[3]);     // (z2, z6, z3, z7)
            }
            *(lds_f4 *)(xring + xw * 256 + lane * 4) = blk;
            xw = xw + 1 == n_blocks ? 0 : xw + 1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // guarded bodies (band-pass + bell in float64 on the float32 pair) | interior | guarded bodies
        int b = 0;
        for (; b < b_a0; ++b) body_a(std::false_type(), b);
        st.to32();
        for (; b < b_a1; ++b) body_a(std::true_type(), b);
        st.to64();
        for (; b < n_bodies; ++b) body_a(std::false_type(), b);
        return;
    }

    // ======================================= stage B ===========================================
    const float *op;
    if (U8) op = lc.store_ok ? (const float *)((unsigned char *)g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride) : nullptr;
    else op = lc.store_ok ? g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    SecamDemodLaneK<float> lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamDemodLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    TapsPk kp;              // the decimator's taps: (even, odd) pairs in VGPRs, (odd, even) pairs in SGPRs (HalfbandDn2Pk)
    kp.load(k.taps);
    TapsPkOdd ko;
    ko.load(k.taps);
    PhaseKPk kph;
    kph.load(k.two_over_pi);
    SecamFinishK fk;
    fk.load(k, lk);
    const int idx1 = ((lane + 63) & 63) * 4;
    SecamDemodPkB st;
    st.reset();
    lds_float *otile = U8 ? (lds_float *)((lds_u8 *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    float own_prev = 0.f, nb_prev = 0.f;
    int xr = lr_m == 0 ? 0 : n_blocks - lr_m;    // delay ring block of body b: lr_m bodies behind A's
    constexpr int kRegDelay = CM_SECAM_PAIR_REG_DELAY;
    f4 lq[kRegDelay > 0 ? kRegDelay : 1];
#pragma unroll
    for (int j = 0; j < (kRegDelay > 0 ? kRegDelay : 1); ++j) lq[j] = f4{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < n_bodies; ++b) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // block b of the ring is complete
        const lds_float *slot = ring + (CM_SECAM_PAIR_MID_BUFS == 2 ? (b & 1) * (kSecamMid / 2) : 0) + lane * 4;
        const f4 ia = *(const lds_f4 *)slot, ib = *(const lds_f4 *)(slot + 256);         // I: (z0, z4, z1, z5), (z2, z6, z3, z7)
        const f4 qa = *(const lds_f4 *)(slot + 512), qb = *(const lds_f4 *)(slot + 768);   // Q likewise
        f4 lw = *(const lds_f4 *)(xring + xr * 256 + lane * 4);
        if (kRegDelay > 0) {      // the rest of the delay in registers
            const f4 in = lw;
            lw = lq[kRegDelay - 1];
#pragma unroll
            for (int j = kRegDelay - 1; j > 0; --j) lq[j] = lq[j - 1];
            lq[0] = in;
        }
        if (CM_SECAM_PAIR_MID_BUFS == 1 && b + 1 < n_bodies) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // A may refill
        xr = xr + 1 == n_blocks ? 0 : xr + 1;
        const int xb = (b - n_pre) << 2;
        auto flush_after = [&](int s) {
            if (s == s_flush && b >= n_pre) {
                const int nn = xb + s - lat_out;
                if (nn >= 0 && ((nn & (kTile - 1)) == kTile - 1 || nn == g.Wp - 1)) {
                    if (U8) flush_tile_u8(g, otile_base, op, nn & ~(kTile - 1), lane);
                    else flush_tile<kTile>(g, otile_base, op, nn & ~(kTile - 1), lane);
                }
            }
        };
        if (CM_SECAM_PAIR_MID && b >= b_mid0 && b < b_mid1) {
            // interior, no guards: the eight phase steps of the body as four packed ones - step j into sample z[j] beside step j + 4
            // into z[j + 4]: the earlier samples of step j > 0 are the pair (z[j - 1], z[j + 3]) as it came from the ring, those of
            // step 0 (z[-1], z[3]) one v_pk_mov - behind ONE wave-uniform branch to the library atan2f, then the decimator two steps
            // per packed update (cm_stages_pk.h: PhaseStepPk, HalfbandDn2Pk; bit-identical to the scalar steps)
            const pf2 ci[4] = {pf2{ia.x, ia.y}, pf2{ia.z, ia.w}, pf2{ib.x, ib.y}, pf2{ib.z, ib.w}};
            const pf2 cq[4] = {pf2{qa.x, qa.y}, pf2{qa.z, qa.w}, pf2{qb.x, qb.y}, pf2{qb.z, qb.w}};
            const pf2 pi[4] = {__builtin_shufflevector(st.last_i, ci[3], 1, 2), ci[0], ci[1], ci[2]};
            const pf2 pq[4] = {__builtin_shufflevector(st.last_q, cq[3], 1, 2), cq[0], cq[1], cq[2]};
            PhaseStepPk ph[4];
            PhaseStepPk::set4(ph, pi, pq, ci, cq);
            st.last_i = ci[3];
            st.last_q = cq[3];
            st.have_prev = 1;
            const float margin = __builtin_fminf(__builtin_fminf(ph[0].margin(), ph[1].margin()), __builtin_fminf(ph[2].margin(), ph[3].margin()));
            pf2 f[4];       // (2 / pi) (d[j], d[j + 4]): frequencies_up - fc (secam.py:148)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(margin > 0.f)) == 0ull, 1)) {
                PhaseStepPk::series4(ph, kph, f);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = ph[j].full(kph);
            }
            // steps s = 0, 1 decimate (d0, d1), (d2, d3) = the low halves, steps 2, 3 the high ones
            const pf2 g01 = st.dn.template push2<0>(kp, ko, f[0], f[1], f[2], f[3]);
            const pf2 g23 = st.dn.template push2<1>(kp, ko, f[0], f[1], f[2], f[3]);
            const float g2[4] = {g01.x, g01.y, g23.x, g23.y};
            const int m0 = m_start + 4 * b;
            const __attribute__((address_space(4))) float *dcp = (const __attribute__((address_space(4))) float *)g.carrier2 + (m0 - lat + P);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float own = st.chroma_back_mid(k, lk, g2[s], dcp[s]);
                const int n = m0 + s - 1 - lat;
                const float luma = st.luma_step_mid(k, lw[s]);
                const Rgb<float> o = st.finish(fk, luma, own_prev, nb_prev);
                own_prev = own;
                nb_prev = lane_from(idx1, own);
                put_rgb<U8, kTile>(otile, wpos, n, o);
                flush_after(s);
            }
            continue;
        }
        // guarded bodies: step s takes (z[2 s], z[2 s + 1])
        const f4 i0 = {ia.x, ib.x, ia.y, ib.y}, i1 = {ia.z, ib.z, ia.w, ib.w};
        const f4 q0 = {qa.x, qb.x, qa.y, qb.y}, q1 = {qa.z, qb.z, qa.w, qb.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m_start + 4 * b + s;
            int m4 = m - lat + P;                               // row-stream sample the decimator completes in this step
            m4 = m4 < 0 ? 0 : (m4 > Lc - 1 ? Lc - 1 : m4);
            const float dc = ((const __attribute__((address_space(4))) float *)g.carrier2)[m4];
            const float own = st.chroma_step(k, kp, lk, m, pf2{i0[s], q0[s]}, pf2{i1[s], q1[s]}, dc);
            const int n = m - 1 - lat;                          // the back end runs one sample behind the exchange
            const float luma = st.luma_step(k, n, lw[s]);
            Rgb<float> o = st.finish(fk, luma, own_prev, nb_prev);
            own_prev = own;
            nb_prev = lane_from(idx1, own);
            if (n >= 0 && n < W) put_rgb<U8, kTile>(otile, wpos, n, o);
            flush_after(s);
        }
    }
}

struct SecamModArgs {
    Geom g;                    // g.lanes -> SecamModLaneK<float, double> table
    SecamModK<float, double> k;
};

// SP = shift of the pre-correction low-pass (register window of the luma delay); DEPTH = 1: line averaging
// RT: SP is the size of the luma delay window, the delay itself is k.s_p <= SP (other sampling rates)
#ifndef CM_SECAM_MOD_WAVES     /* waves per SIMD the register allocation of the SECAM encoder aims at */
#define CM_SECAM_MOD_WAVES 2
#endif
template <int SP, int DEPTH, bool U8 = false, bool RT = false>
__global__ __launch_bounds__(64, CM_SECAM_MOD_WAVES) void secam_mod_kernel(const SecamModArgs args) {
    constexpr int kTile = kSecamModTile;
    __shared__ __attribute__((aligned(16))) float lds_store[U8 ? kModLdsFloatsU8 : mod_lds_floats<kTile>()];
    lds_float *itile = (lds_float *)lds_store;
    lds_float *otile_base = itile + (U8 ? kInTile3Bytes / 4 : kLdsIn3);
    const Geom &g = args.g;
    const SecamModK<float, double> &k = args.k;
    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, xcd_block((int)blockIdx.x, (int)gridDim.x), DEPTH, lane);
    const float *rp, *op;
    mod_rows<U8>(g, lc, rp, op);
    SecamModLaneK<float, double> lk;
    {
        int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamModLaneK<float, double> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    const int idx1 = ((lane + 63) & 63) * 4;
    SecamMod<float, double> st;
    st.reset();
    float yw[SP + 4];
#pragma unroll
    for (int j = 0; j < SP + 4; ++j) yw[j] = 0.f;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;
    const int W = g.W;
    const int sp = RT ? k.s_p : SP;
    const int T = (g.Wp + sp + 3) & ~3;
    f4 cur[3], nxt[3];
    first_tile3<U8>(g, itile, rp, lane, nxt);
    // interior bodies: s_p < t and t + 3 < W - 1 for the four steps - no guard of SecamMod::step can fail and the output
    // sample n7 = t - s_p lies inside the row
    int tb_mid0 = (sp + 1 + 3) & ~3, tb_mid1 = (W - 5) & ~3;
    if (tb_mid1 <= tb_mid0) tb_mid0 = tb_mid1 = 0;
    auto body = [&](auto edge_tag, int tb) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        next_tile3x<U8>(g, itile, rp, lane, tb + 4, nxt);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tb + s;
            float r = cur[0][s], gg = cur[1][s], b = cur[2][s];
            float y = fmaf_(k.e[0][0], r, fmaf_(k.e[0][1], gg, k.e[0][2] * b));
            float dr = fmaf_(k.e[1][0], r, fmaf_(k.e[1][1], gg, k.e[1][2] * b));
            float db = fmaf_(k.e[2][0], r, fmaf_(k.e[2][1], gg, k.e[2][2] * b));
            float d = lk.own_is_db != 0.f ? db : dr;     // secam.py:262-271 (the line being modulated decides)
            if (DEPTH >= 1) {
                // the previous call's components: its lane computed them with its own parity, so exchange both
                float yp = lane_from(idx1, y), drp = lane_from(idx1, dr), dbp = lane_from(idx1, db);
                float dp = lk.own_is_db != 0.f ? dbp : drp;
                y = fmaf_(lk.wy0, y, lk.wy1 * yp);
                d = fmaf_(lk.wc0, d, lk.wc1 * dp);
            }
            yw[SP + s] = y;
            const int n7 = t - sp;
            float y_d = yw[s];                 // luma of sample t - SP
            if (RT) {                          // ... t - s_p: uniform selects instead of a dynamic register index
#pragma unroll
                for (int j = 0; j < SP; ++j)
                    if (sp == j) y_d = yw[SP - j + s];
            }
            float comp = st.template step<EDGE>(k, lk, t, y_d, d);
            put_composite<U8, kTile>(g, otile_base, op, lane, wpos, n7, comp);
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

// cm_kernels.h - device-side lane drivers of the QAM-family demodulators (gfx950).
//
// 64 consecutive calls of the flattened call list [frame][run][call] form one workgroup; lane j owns call
// block * (64 - DEPTH) - DEPTH + j, i.e. consecutive workgroups overlap by DEPTH halo lanes that only feed their base
// pairs to their neighbours.  Two drivers run the same stages (cm_stages.h, cm_stages_pk.h):
//   run_pair  (demod_pair_kernel)  two wavefronts per workgroup: stage A = loads + front end, stage B = detectors in packed
//             float32 + back end + stores, hand-over through an LDS ring; 2.5 - 3 waves per SIMD (2 where stage B carries a
//             second line of history, the notch or minavg).  Every instance of the default build.  See the comment above run_pair.
//   run_lane  (demod_kernel)       one wavefront per workgroup, 2 waves per SIMD: the earlier structure, kept for A/B builds
//             (-DCM_PAIR=0, -DCM_ONE_WAVE_SELECT: PassCfg::kUsePair).
//
// Data movement of a workgroup:
//   input   64 rows x 16 samples per tile in the pair kernels (4 global_load_lds_dwordx4, 16 rows x 64 B each; 32 samples
//           for byte rows and in the one-wave kernel), no VGPR staging; lane i then reads its own row with one
//           ds_read_b128 per 4 steps.  Every input byte crosses the fabric once for this stream.
//   luma    x_l[n7 .. +3]: in the pair kernels stage A leaves the samples in an LDS delay ring straight out of its x window
//           (CM_LUMA_RING: every input byte is read from memory once); the one-wave kernel visits the (own or previous) row
//           a second time, one unaligned global_load_dwordx4 per lane and 4 steps.
//   output  r, g, b: one ds_write_b32 per plane and step into a [3][64][16] LDS tile (quad-swizzled columns); every
//           16 steps the tile is read back row-wise (ds_read_b128) and stored as 64-byte row segments, 16 rows per
//           wave-instruction.
//   carrier cos/sin(m cps): wave-uniform scalar loads from tables padded by kCarrierPad entries at both ends.
//   neighbours' base pairs: ds_bpermute_b32, consumed one step later (the back end runs one sample behind the
//           detectors so that the permute latency is never waited for).
#ifndef CM_KERNELS_H
#define CM_KERNELS_H

#include <type_traits>

#include "cm_stages.h"
#include "cm_stages_pk.h"

namespace cm {

enum { FRONT_QAM = 0, FRONT_PALD = 1 };

struct Geom {
    const float *in;
    float *out;
    const LaneK<float> *lanes;  // [cycle][3][n_lines]
    const float *carrier4;      // {C[2n], S[2n], C[2n+1], S[2n+1]}, n < W   (C/S = cos/sin(m cps))
    const float *carrier2;      // {C[2n], S[2n]}, n < W
    long long in_frame_stride, in_plane_stride, in_row_stride;   // in_row_stride = 0 means W
    long long out_frame_stride, out_plane_stride, out_row_stride;
    long long total_calls;      // main pass: n_frames * calls_per_frame; sparse pass: n_frames * runs_per_frame
    int first_frame, cycle, n_lines;
    const float *frame_rot;     // {cos, sin} per frame of the rotation cycle, or null (cm_plan_desc::frame_rotation)
    int rot_first, rot_cycle;
    int W, H;
    int Wp;              // float rows: row pitch in samples = W rounded up to a multiple of 4 (rows move as 16-byte vectors; for
                         // other widths the library stages the images through pitched buffers); byte modes: = W
    int calls_per_frame, calls_run0, runs_per_frame;
    int first_line[2];
    int k0;              // rows mode: index of the first submitted call within its run
    int delay;           // demodulation_delay (frames mode: output row = line - 2 * delay)
    int rows_mode;       // 1: input row i / output row i are the i-th submitted rows of one run
    int luma_prev_bits;  // bit r: regime r takes its luma from the previous call's input row; bit 8 + r: from the call before that one
                         // (two-level combs)
    int wrap_mode;       // PassCfg::WRAP instances: 1 = comb.avg, 2 = comb.minavg of consecutive calls' (u, v)
    int sparse;          // 1: one lane per run, call 0 of each run only (plain first-line pass)
    int seg_len;         // > 0: small batches - the row is cut into segments of seg_len samples (a multiple of 16) and every
    int seg_blocks;      // workgroup walks ONE segment of its 64 calls, starting seg_warm samples early from a zero state (the
    int seg_warm;        // recursive filters forget it: cm_api.hip: segment_geometry); blocks [seg * seg_blocks, ...) own segment seg
    int in_calls;        // frames geometry, but input row = the call's index within its frame ([frame][call] buffers: the comb
    int out_calls;       // wrappers' scratch, cm_wrap_kernels.h); likewise the output row, and EVERY call is stored
    int skip_first;      // n: the calls with k < n of every run are written by another launch (1: the sparse plain first-line pass;
                         // 2: the fused wrapped comb, whose first two calls mix two front ends - cm_api.hip: wrap_frames_fused)
    int keep_calls;      // n > 0: ONLY the calls with k < n are stored (the launch that supplies what skip_first = n leaves out)
    unsigned long long *diag;  // diagnostic builds only (-DCM_DIAG): per-workgroup cycle sums; null otherwise
    unsigned *simd_load;       // wave-pair kernels (CM_SIMD_BALANCE): live load per (XCC, CU, SIMD); null: waves keep their order
    const void *blk_tiles;     // blocked decoder (cm_blk_kernels.h): the Toeplitz tiles of the half-band FIR, or null
};

// Long sub-carrier cycles: advance every phase a lane constant carries by the frame's angle (c, s) = {cos, sin}.
// A coefficient pair (a, b) standing for a cos(theta) - b sin(theta) / a sin(theta) + b cos(theta) turns with it.
__device__ __forceinline__ void turn(float &a, float &b, float c, float s) {
    const float a2 = a * c - b * s, b2 = a * s + b * c;
    a = a2;
    b = b2;
}
__device__ __forceinline__ bool frame_turn(const Geom &g, long long frame, float &c, float &s) {
    if (!g.frame_rot) return false;
    const int i = (int)((g.rot_first + frame) % g.rot_cycle);
    c = g.frame_rot[2 * i];
    s = g.frame_rot[2 * i + 1];
    return true;
}
__device__ __forceinline__ void apply_frame_rotation(const Geom &g, long long frame, LaneK<float> &lk) {
    float c, s;
    if (!frame_turn(g, frame, c, s)) return;
    turn(lk.cph, lk.sph, c, s);
    turn(lk.vcph, lk.vsph, c, s);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        turn(lk.cu[j][0], lk.cu[j][1], c, s);
        turn(lk.cv[j][0], lk.cv[j][1], c, s);
        turn(lk.cu2[j][0], lk.cu2[j][1], c, s);
        turn(lk.cv2[j][0], lk.cv2[j][1], c, s);
    }
}

#ifdef CM_DIAG
__device__ __forceinline__ unsigned long long cm_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long cm_realtime() {   // 100 MHz reference
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define CM_STAMP(var) unsigned long long var = cm_stamp()
#define CM_ACC(acc, t0) acc += cm_stamp() - (t0)
#else
#define CM_STAMP(var)
#define CM_ACC(acc, t0)
#endif

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8u __attribute__((ext_vector_type(8), aligned(4)));
typedef float f16u __attribute__((ext_vector_type(16), aligned(4)));
typedef const __attribute__((address_space(4))) f4 const_f4;
typedef const __attribute__((address_space(4))) f2 const_f2;
typedef const __attribute__((address_space(4))) f8u const_f8;
typedef const __attribute__((address_space(4))) f16u const_f16;

// Cache policy of the streaming accesses (measured, profiles/r01_notes.md): the input tile fill is
// marked nt (each line is needed once by this stream) and the output stores are nt; both keep the
// 4 MiB L2 of an XCD for the luma re-read, whose lines are touched 8 times.
#ifndef CM_FILL_AUX
#define CM_FILL_AUX 2
#endif
// Wave-pair kernels without a band-stop luma: 1 = stage A leaves the luma source samples x_l[n7] in an LDS delay ring straight
// out of its x window (every input byte is read from memory once); 0 = stage A fetches them a second time with one
// global_load_dwordx4 per lane and body (the one-wave kernels always do).
#ifndef CM_LUMA_RING
#define CM_LUMA_RING 1
#endif
// LDS budget knobs of the pair kernels (profiles/r01_pair_notes.md section 11): samples per float input tile row (8: 32-byte
// row segments, 2 KiB) and extra x samples stage A of the PAL-D front end keeps in registers behind its 14-sample window (a
// multiple of 4; every 4 shorten the delay ring by one 1 KiB block).  8 / 12 bring the PAL-BG decoder from 31 to 26 KiB =
// 6 workgroups per CU (measured 2.86 -> 2.74 ms per 1000 frames against 16 / 0; either knob alone changes nothing).
#ifndef CM_PAIR_TILE
#define CM_PAIR_TILE 8
#endif
#ifndef CM_RING_WINDOW
#define CM_RING_WINDOW 12
#endif
#ifndef CM_RING_WINDOW_QAM
#define CM_RING_WINDOW_QAM 0
#endif
// PAL-D front end on the PAL-BG filter shape: CM_RING_WINDOW; on the order-6 band-pass shapes its stage A has no registers
// to spare.  QAM front end: CM_RING_WINDOW_QAM (16 brings the NTSC comb decoder to 26 KiB as well, measured without effect:
// 2.165 ms either way - off).  Run-time shape: none (it runs 4 workgroups per CU).
// LC: the instance cuts the pair behind the detector low-pass (two-line combs on the QAM front end: CM_QAM_LPF_IN_A); its
// hand-over ring is 8 KiB instead of 4, and CM_RING_WINDOW_LCUT samples in stage A's registers bring it back to 25 - 26 KiB.
#ifndef CM_RING_WINDOW_LCUT
#define CM_RING_WINDOW_LCUT 12
#endif
template <class S, int FRONT, bool LC = false> constexpr int ring_window() {
    return S::RT ? 0 : (FRONT == 1 ? (S::NE < 3 ? CM_RING_WINDOW : 0) : (LC ? CM_RING_WINDOW_LCUT : CM_RING_WINDOW_QAM));
}
// Output tiles are [row = lane][kTile samples]; the 16-byte quad a lane writes is XORed with lane bits CM_TILE_SWZ, +1 so that
// the one ds_write_b32 per plane and step of 64 lanes spreads over more banks (rows are 64 bytes apart).
#ifndef CM_TILE_SWZ
#define CM_TILE_SWZ 1
#endif
#ifndef CM_CVT_PK_U8
#define CM_CVT_PK_U8 1
#endif
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f4 lds_f4;
constexpr int kInTile = 32;        // samples per input tile (one 128-byte line per row)
constexpr int kCarrierPad = 128;   // entries before and after the carrier tables (copies of the first / last entry)
constexpr int kLdsIn = 64 * kInTile;        // floats
constexpr int kLdsRing = 16 * 64;           // floats (band-stop luma delay ring)

__device__ __forceinline__ float lane_from(int byte_index, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ const float *ptr_from(int byte_index, const float *p) {
    unsigned long long v = (unsigned long long)p;
    unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_index, (int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_index, (int)(unsigned)(v >> 32));
    return (const float *)(((unsigned long long)hi << 32) | lo);
}

// TILE: samples per output tile (16: 64-byte row segments; 8: 32-byte segments, used where LDS is
// short and the pass writes a negligible share of the rows)
// U8: the ImageModem byte boundary fused into the kernel (ref image.py:7-8, 24-25, 62, 65-71): composite is
// uint8 [F][H][W] and enters through (5 * (byte / 255) - 1) / 3; the output is interleaved uint8 RGB
// [F][H][W][3] = rint(255 * clip(x, 0, 1)).  The LDS tiles hold bytes in that mode (same float-sized budget).
#ifndef CM_QAM_LPF_IN_A
#define CM_QAM_LPF_IN_A 1
#endif
#ifndef CM_LCUT_DEPTH2_WAVES
#define CM_LCUT_DEPTH2_WAVES 3
#endif
// the PAL-D front end on the shapes of the wide rasters (cm_shapes_wide.h): waves per SIMD the instances are compiled for.  Their luma delay
// ring (30 - 34 KiB of LDS with it) leaves room for five or four workgroups per CU, and measured (profiles/r06_wide_shapes.txt) the 168-register
// cap of 3 waves per SIMD - 8 to 72 bytes of scratch - loses to 2 waves per SIMD without spills from 1024 samples per line on (1280: 124 -> 130
// Gpixel/s, 1920: 116 -> 120); the fused wrapper instances (two lines of history) of the 800 - 1024 rasters are the exception (118 -> 120 / 108 -> 111)
#ifndef CM_WIDE_PALD_WAVES
#define CM_WIDE_PALD_WAVES 2
#endif
#ifndef CM_WIDE_PALD_UP2     /* ... and whether their stage A runs the third half-band chain two steps at a time (HalfbandUp2Pk), which needs the registers of 2 waves per SIMD */
#define CM_WIDE_PALD_UP2 0
#endif
// WRAP_: two-level comb (round 5; SimpleCombModem / Simple3DCombModem around Pal3DModem, comb.py:96-113 over pal.py:180-234): the lane
// tables are the inner decoder's, and stage B averages (Geom::wrap_mode 1) or min-averages (2) the (u, v) they give with the (u, v) the
// neighbouring lane - the previous call of the run - formed the same way, one step later in the stream; DEPTH_ + 1 halo lanes.
template <class S_, int FRONT_, bool BSF_, int DEPTH_, int TILE_, bool U8_ = false, bool NOTCH_ = false, bool MINAVG_ = false, bool WRAP_ = false>
struct PassCfg {
    typedef S_ S;
    static constexpr int FRONT = FRONT_, DEPTH = DEPTH_, TILE = TILE_;
    static constexpr bool BSF = BSF_, U8 = U8_, NOTCH = NOTCH_, MINAVG = MINAVG_, WRAP = WRAP_;
    static constexpr int HALO = DEPTH_ + (WRAP_ ? 1 : 0);      // lanes of a workgroup that only recompute earlier calls
    static constexpr int kLdsInF = U8_ ? 64 * kInTile / 4 : kLdsIn;            // floats: byte tiles are a quarter
    static constexpr int kLdsOut = U8_ ? 64 * 3 * TILE_ / 4 : 3 * 64 * TILE_;  // floats
    static constexpr int kLdsFloats = kLdsInF + kLdsOut + (BSF_ ? kLdsRing : 0);
    // wave-pair kernels: float rows enter through 16-sample tiles (64-byte row segments) when the luma delay ring takes the
    // LDS (CM_LUMA_RING); byte tiles stay 32 samples wide
    // wave-pair kernels: 3 waves per SIMD need <= 168 VGPRs; the instances with more per-lane state in stage B (a second
    // line of history, the notch, the second combination of minavg) would spill there and get 2 waves per SIMD instead
    // (with the cut behind the detector low-pass, CM_QAM_LPF_IN_A, stage B of the two-line combs sheds the low-pass state and
    // its coefficients: CM_LCUT_DEPTH2_WAVES = 3 asks for 168 VGPRs there too)
    static constexpr bool kLcutCfg = CM_QAM_LPF_IN_A != 0 && FRONT_ == 0 && !BSF_ && DEPTH_ >= 2 && !S_::RT;
    // Round 5 measured the instances that sit just above 168 again (same-box A/Bs, profiles/r05_wrapped_fused.txt): capped at 168 the fused
    // comb wrappers - PAL-D front end with two lines of history, Pal3DModem's two-level comb - spill 2 - 16 registers outside their interior
    // bodies and run 9 - 11 % faster at 3 waves per SIMD, the one-line decoders with a notch on the NTSC shape (169 - 171 VGPRs, one spill) 7 %;
    // the run-time filter shape loses 4 - 6 % there and stays at 2, the other instances are indifferent.
    static constexpr bool kWrapperCfg = !S_::RT && S_::NE < 4 && S_::NP < 2 && ((FRONT_ == 1 && DEPTH_ >= 2) || WRAP_ || (DEPTH_ == 1 && NOTCH_ && !MINAVG_));
    // Which wide PAL-D-front instances run WITHOUT the luma delay ring (stage A fetches the luma source a second time: one load per lane and
    // body, L2 hits) at 3 waves per SIMD - 20 KiB of LDS, six workgroups per CU.  Bit 0: the byte instances - the decoder (+3 ... +5 % at
    // 800 - 1920 samples per line) and, from a pre-correction shift of 4 on (1280 ...), the fused wrapper (+2.5 ... +5.5 %; at 800 / 1024 it
    // already runs 3 waves with its ring and loses 23 % without).  Bit 1: the float one-line decoder of the 1920 class (+2 %, and 142 ... 151
    // instead of 136 ... 151 Gpixel/s over 30 buffer placements: more resident waves hide the slow placements).  The float wrappers (two
    // lines of history) do not fit 168 registers: -7 ... -9 %.  profiles/r06_xcd_remap.txt, follow-up 4.
#ifndef CM_WIDE_PALD_NORING
#define CM_WIDE_PALD_NORING 3
#endif
    static constexpr bool kNoRing = S_::NORING || (S_::WIDE && FRONT_ == 1 && CM_LUMA_RING != 0 &&
                                                   ((((CM_WIDE_PALD_NORING) & 1) != 0 && U8_ && (DEPTH_ == 1 || S_::SP >= 4)) ||
                                                    (((CM_WIDE_PALD_NORING) & 2) != 0 && !U8_ && DEPTH_ == 1 && S_::SP >= 6)));
    static constexpr int kPairWaves = (S_::WIDE && FRONT_ == 1) ? ((kNoRing || (DEPTH_ >= 2 && S_::SP <= 3)) ? 3 : CM_WIDE_PALD_WAVES)
                                    : kWrapperCfg ? 3
                                    : (NOTCH_ || MINAVG_ || S_::NE >= 4 || S_::NP >= 2 || S_::RT) ? 2
                                    : (DEPTH_ >= 2 ? (kLcutCfg ? CM_LCUT_DEPTH2_WAVES : 2) : 3);
    // The widest shapes (pre-correction shift >= 6: 1920 samples per line) at 2 waves per SIMD keep TWO input tiles per row, tile t + 1 asked
    // for at the first read of tile t: the fill has two bodies (8 steps) to arrive instead of one.  With one tile a fill that takes longer
    // than a body stalls stage A and, through the barrier, the pair - the slow mode PAL-D and the NTSC combs fell into at that width
    // (137 -> 152 Gpixel/s on a box that showed it; profiles/r06_xcd_remap.txt).  Not elsewhere: the instances whose registers allow
    // 3 waves per SIMD lose a workgroup per CU to the 2 KiB (5 - 7 %), the other 2-wave instances measured 0 +- 2 %.
#ifndef CM_WIDE_TWO_TILES
#define CM_WIDE_TWO_TILES 1
#endif
#ifndef CM_WIDE_TWO_TILES_SP
#define CM_WIDE_TWO_TILES_SP 6
#endif
#ifndef CM_WIDE_TILE          /* samples per input tile row of that class; 16 (64-byte row segments), one or two tiles: mean of 30 placements 139 - 140 against 144 Gpixel/s with two tiles of 8 (one of 8: 136), profiles/r06_xcd_remap.txt */
#define CM_WIDE_TILE 8
#endif
    static constexpr bool kWideLatClass = S_::WIDE && S_::SP >= CM_WIDE_TWO_TILES_SP && !BSF_ && !U8_ && CM_LUMA_RING != 0 && kPairWaves == 2;
    static constexpr int kPairInTile = (U8_ || !CM_LUMA_RING) ? kInTile : (kWideLatClass ? CM_WIDE_TILE : CM_PAIR_TILE);
    static constexpr bool kPairTwoTiles = CM_WIDE_TWO_TILES != 0 && kWideLatClass;
    static constexpr int kPairLdsIn = U8_ ? 64 * kInTile / 4 : 64 * kPairInTile * (kPairTwoTiles ? 2 : 1);   // floats
    // which kernel structure runs this instance.  Since the luma delay ring (CM_LUMA_RING) every instance runs on the wave
    // pair - those with more per-lane state in stage B at 2 waves per SIMD (measured: Pal3D 2.92 -> 2.72 ms, Simple3DComb(
    // NtscComb) 2.43 -> 2.34 ms per 1000 frames against the one-wave kernel, profiles/r01_pair_notes.md section 9).
    // -DCM_ONE_WAVE_SELECT restores the earlier choice: the pair only where it gets 3 waves per SIMD or one wave's 256 VGPRs
    // do not hold the line (PAL-D front end with the notch, the run-time shape).
#ifdef CM_ONE_WAVE_SELECT
    static constexpr bool kUsePair = kPairWaves == 3 || (FRONT_ == 1 && NOTCH_) || S_::RT;
#else
    static constexpr bool kUsePair = true;
#endif
};
struct NoPass {
    static constexpr int kLdsFloats = 0;
};

// uint8 composite sample -> level-decoded float (image.py:24-25, 62)
__device__ __forceinline__ f4 decode_bytes(unsigned w) {
    const float a = 5.0f / (255.0f * 3.0f), b = -1.0f / 3.0f;
    f4 v;
    v.x = __builtin_fmaf((float)(w & 0xffu), a, b);
    v.y = __builtin_fmaf((float)((w >> 8) & 0xffu), a, b);
    v.z = __builtin_fmaf((float)((w >> 16) & 0xffu), a, b);
    v.w = __builtin_fmaf((float)(w >> 24), a, b);
    return v;
}

// x_l[first .. first + 3] of the luma source row lp (the second visit of the own or the previous row), zero outside
// the row.  check = false: the caller knows the four samples lie inside.  U8: the row holds bytes (level-decoded here).
template <bool U8>
__device__ __forceinline__ f4 load_luma(const float *lp, int first, bool check, int W) {
#ifdef CM_EXP_NO_LUMA   /* timing experiment (profiles/r01_pair_notes.md) */
    return f4{0.1f, 0.2f, 0.3f, 0.4f};
#endif
    if (U8) {
        const unsigned char *lb = (const unsigned char *)lp;
        if (!check || (first >= 0 && first + 3 < W)) {
            typedef unsigned u32u __attribute__((aligned(1)));
            return decode_bytes(*(const u32u *)(lb + first));
        }
        f4 r = {0.f, 0.f, 0.f, 0.f};
        if (first + 3 >= 0 && first < W) {
            unsigned w = 0;
            if (first >= 0 && first < W) w |= lb[first];
            if (first + 1 >= 0 && first + 1 < W) w |= (unsigned)lb[first + 1] << 8;
            if (first + 2 >= 0 && first + 2 < W) w |= (unsigned)lb[first + 2] << 16;
            if (first + 3 >= 0 && first + 3 < W) w |= (unsigned)lb[first + 3] << 24;
            f4 d = decode_bytes(w);
            if (first >= 0 && first < W) r.x = d.x;
            if (first + 1 >= 0 && first + 1 < W) r.y = d.y;
            if (first + 2 >= 0 && first + 2 < W) r.z = d.z;
            if (first + 3 >= 0 && first + 3 < W) r.w = d.w;
        }
        return r;
    }
    if (!check || (first >= 0 && first + 3 < W)) {
        f4u v = *(const f4u *)(lp + first);
        return f4{v.x, v.y, v.z, v.w};
    }
    f4 r = {0.f, 0.f, 0.f, 0.f};
    if (first + 3 >= 0 && first < W) {
        if (first >= 0 && first < W) r.x = lp[first];
        if (first + 1 >= 0 && first + 1 < W) r.y = lp[first + 1];
        if (first + 2 >= 0 && first + 2 < W) r.z = lp[first + 2];
        if (first + 3 >= 0 && first + 3 < W) r.w = lp[first + 3];
    }
    return r;
}

// One output sample into this lane's row of the output tile (float: three quad-swizzled planes; U8: interleaved R, G, B
// bytes = uint8(rint(255 * clip(x, 0, 1))), image.py:7-8).
template <bool U8, int kTile>
__device__ __forceinline__ void put_rgb(lds_float *otile, int wpos, int n7, const Rgb<float> &o) {
    if constexpr (U8) {
        typedef __attribute__((address_space(3))) unsigned char lds_u8;
        lds_u8 *tb = (lds_u8 *)otile + 3 * (n7 & (kTile - 1));   // otile: this lane's 48-byte row
#if CM_CVT_PK_U8   /* v_cvt_pk_u8_f32: round to nearest even, saturate to 0 .. 255, pack - one instruction per byte (tools/ubench_cvt_u8.hip) */
        unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(255.f * o.r, 0, 0);
        w = __builtin_amdgcn_cvt_pk_u8_f32(255.f * o.g, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(255.f * o.b, 2, w);
        tb[0] = (unsigned char)w;
        tb[1] = (unsigned char)(w >> 8);
        tb[2] = (unsigned char)(w >> 16);
#else
        tb[0] = (unsigned char)__builtin_rintf(255.f * __builtin_fminf(__builtin_fmaxf(o.r, 0.f), 1.f));
        tb[1] = (unsigned char)__builtin_rintf(255.f * __builtin_fminf(__builtin_fmaxf(o.g, 0.f), 1.f));
        tb[2] = (unsigned char)__builtin_rintf(255.f * __builtin_fminf(__builtin_fmaxf(o.b, 0.f), 1.f));
#endif
    } else {
        lds_float *tp = otile + (wpos ^ (n7 & (kTile - 1)));
        tp[0] = o.r;
        tp[64 * kTile] = o.g;
        tp[2 * 64 * kTile] = o.b;
    }
}

template <class Cfg>
struct DemodLane {
    typedef typename Cfg::S S;
    static constexpr int FRONT = Cfg::FRONT, DEPTH = Cfg::DEPTH, SP = S::SP, kTile = Cfg::TILE;
    static constexpr bool BSF = Cfg::BSF;
    typedef DemodK<float, S> K;
    typedef typename std::conditional<FRONT == FRONT_PALD, PalDFront<float, S>, QamFront<float, S, BSF>>::type Front;

    Front front;
    DemodBack<float, S, DEPTH, Cfg::NOTCH, Cfg::MINAVG> back;
    LaneK<float> lk;
    float xw[14];
    float ew[FRONT == FRONT_PALD ? 14 : 1];
    float ud[SP > 0 ? SP : 1], vd[SP > 0 ? SP : 1];   // u, v of the last SP steps (newest first)
    f4 lw;
    Pair<float> base_prev, b1_prev, b2_prev;
    const float *lp;
    int idx1, idx2;

    // One step.  car: detector carriers of this step, carb: re-modulation carrier of the back-end
    // sample n7 = tau - lat_front - 1 - SP.  Returns true when an output tile has just been completed.
    template <int SUB, bool EDGE>
    __device__ __forceinline__ void substep(const Geom &g, const K &k, FrontLatch<float> &fla, BackLatch<float> &bla, int tau,
                                            int lat_front, int lat_luma, const float car[4], const float carb[2],
                                            lds_float *otile, lds_float *yring, int lane, int wpos) {
        const int W = g.W;
        float luma_bsf = 0.f;
        Pair<float> base;
        if constexpr (FRONT == FRONT_PALD) {
            float e_out;
            base = front.template step<EDGE>(k, fla, tau, xw[10 + SUB], xw[SUB], ew[FRONT == FRONT_PALD ? SUB : 0], car, e_out);
            ew[FRONT == FRONT_PALD ? 10 + SUB : 0] = e_out;
        } else {
            base = front.template step<EDGE>(k, fla, tau, xw[10 + SUB], xw[SUB], car, luma_bsf);
        }
        // the back end handles the PREVIOUS step's base pair: its neighbours were requested then
        const int sp = S::RT ? k.s_p : SP;     // run-time shapes: SP is the window size, k.s_p the delay
        const int n6 = tau - lat_front - 1, n7 = n6 - sp;
        float u, v;
        back.combine(lk, base_prev, b1_prev, b2_prev, u, v);
        base_prev = base;
        if (DEPTH >= 1) { b1_prev.s = lane_from(idx1, base.s); b1_prev.c = lane_from(idx1, base.c); }
        if (DEPTH >= 2) { b2_prev.s = lane_from(idx2, base.s); b2_prev.c = lane_from(idx2, base.c); }
        float y_src;
        if (BSF) {
            const int nl = tau - lat_luma;
            yring[(nl & 15) * 64 + lane] = luma_bsf;
            y_src = yring[(n7 & 15) * 64 + lane];
        } else {
            y_src = SUB == 0 ? lw.x : (SUB == 1 ? lw.y : (SUB == 2 ? lw.z : lw.w));
        }
        float u_d = SP > 0 ? ud[SP > 0 ? SP - 1 : 0] : u, v_d = SP > 0 ? vd[SP > 0 ? SP - 1 : 0] : v;
        if (S::RT) {   // (u, v)[n6 - s_p] out of the window: a chain of uniform selects instead of a dynamic register index
            if (sp == 0) { u_d = u; v_d = v; }
#pragma unroll
            for (int j = 0; j + 1 < SP; ++j)
                if (sp == j + 1) { u_d = ud[j]; v_d = vd[j]; }
        }
        Rgb<float> o = back.template step<EDGE>(k, lk, bla, n6, u, v, u_d, v_d, y_src, carb);
#pragma unroll
        for (int j = SP - 1; j > 0; --j) { ud[j] = ud[j - 1]; vd[j] = vd[j - 1]; }
        if (SP > 0) { ud[0] = u; vd[0] = v; }
        if (!EDGE || (n7 >= 0 && n7 < W)) put_rgb<Cfg::U8, kTile>(otile, wpos, n7, o);
    }
};


// Byte variant of flush_tile: the tile row of a lane is 16 pixels x 3 bytes; 16 rows x 48 bytes per wave-instruction.
__device__ __forceinline__ void flush_tile_u8(const Geom &g, const lds_float *otile, const float *op, int first_col, int lane) {
    __builtin_amdgcn_wave_barrier();
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(1))) unsigned global_u32;
    const int chunk = lane & 3;               // 4 pixels = 12 bytes
    const int col = first_col + 4 * chunk;
#pragma nounroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 2) + 16 * q;
        global_u32 *dst = (global_u32 *)(unsigned long long)ptr_from(row * 4, op);
        if (dst != nullptr && col < g.W) {
            const lds_u32 *src = (const lds_u32 *)otile + row * 12 + 3 * chunk;
            unsigned w0 = src[0], w1 = src[1], w2 = src[2];
            dst += (3 * col) >> 2;
            __builtin_nontemporal_store(w0, &dst[0]);
            __builtin_nontemporal_store(w1, &dst[1]);
            __builtin_nontemporal_store(w2, &dst[2]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Row-wise read-back of the output tile and coalesced store: (256 / kTile) rows x (4 kTile) bytes per
// wave-instruction.
template <int kTile>
__device__ __forceinline__ void flush_tile(const Geom &g, const lds_float *otile, const float *op, int first_col, int lane) {
    __builtin_amdgcn_wave_barrier();
    constexpr int kChunks = kTile / 4;        // 16-byte chunks per row
    constexpr int kRows = 64 / kChunks;       // rows per wave-instruction
    const int chunk = lane & (kChunks - 1);
    const int col = first_col + 4 * chunk;
    // rolled on purpose: this runs once per 16 steps, and unrolling it would add its temporaries to
    // the register peak of the whole kernel
#pragma nounroll
    for (int q = 0; q < kChunks; ++q) {
        const int row = lane / kChunks + kRows * q;
        typedef __attribute__((address_space(1))) f4 global_f4;
        global_f4 *dst = (global_f4 *)(unsigned long long)ptr_from(row * 4, op);
        const int quad = chunk ^ ((row >> CM_TILE_SWZ) & (kChunks - 1));
        if (dst != nullptr && col < g.Wp) {
            dst += col >> 2;
#ifndef CM_FLUSH_SERIAL   /* the three planes of a row group in one LDS round trip (-1 % kernel time) */
            f4 v0 = *(const lds_f4 *)(otile + 0 * 64 * kTile + row * kTile + 4 * quad);
            f4 v1 = *(const lds_f4 *)(otile + 1 * 64 * kTile + row * kTile + 4 * quad);
            f4 v2 = *(const lds_f4 *)(otile + 2 * 64 * kTile + row * kTile + 4 * quad);
#ifdef CM_EXP_NO_STORE   /* timing experiment: results are not written */
            if (v0.x == 12345.678f)
#endif
            {
#ifdef CM_EXP_PLAIN_STORE   /* experiment: default cache policy for the output stores */
                dst[0] = v0;
                dst[g.out_plane_stride >> 2] = v1;
                dst[(2 * g.out_plane_stride) >> 2] = v2;
#else
                __builtin_nontemporal_store(v0, &dst[0]);
                __builtin_nontemporal_store(v1, &dst[g.out_plane_stride >> 2]);
                __builtin_nontemporal_store(v2, &dst[(2 * g.out_plane_stride) >> 2]);
#endif
            }
#else
#pragma nounroll
            for (int p = 0; p < 3; ++p) {
                f4 v = *(const lds_f4 *)(otile + p * 64 * kTile + row * kTile + 4 * quad);
#ifdef CM_EXP_NO_STORE
                if (v.x == 12345.678f)
#endif
                __builtin_nontemporal_store(v, &dst[(p * g.out_plane_stride) >> 2]);
            }
#endif
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Fill input tile `c` (samples 32 c .. 32 c + 31 of all 64 rows) straight into LDS.
// IT samples per tile row: IT / 4 lanes x 16 bytes per row, 256 / IT rows per instruction, IT / 4 instructions.
template <int IT = kInTile>
__device__ __forceinline__ void fill_tile(const Geom &g, lds_float *itile, const float *xp, int c, int lane) {
    constexpr int kLanesPerRow = IT / 4, kRowsPerInstr = 64 / kLanesPerRow;
    // 64-byte segments: both halves of a 128-byte line are asked for 16 steps apart - no streaming hint there
    constexpr int kAux = IT == kInTile ? CM_FILL_AUX : 0;
    int col = IT * c + 4 * (lane & (kLanesPerRow - 1));
    if (col > g.Wp - 4) col = g.Wp - 4;  // never read past the (pitched) row; such samples are masked by the consumer
#pragma nounroll
    for (int q = 0; q < kLanesPerRow; ++q) {
        const float *src = ptr_from((kRowsPerInstr * q + lane / kLanesPerRow) * 4, xp) + col;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(itile + q * 256), 16, 0, kAux);
    }
}

// Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md: blocks b and b + 8 share one).  With the linear block
// id as the position in the batch, every XCD walks the whole batch at once: each of its translation caches and its L2 see the pages and
// the halo rows of all ~1000 resident workgroups.  1: the blocks of one XCD take a contiguous eighth of the batch instead (bijective for
// any grid size) - a choice of speed only, the result does not depend on it.  Measured (profiles/r06_xcd_remap.txt): rows whose pitch
// is a multiple of 4 KiB + 6 - 7 %, and the slow mode some widths fell into on some runs (PAL-D at 1600 / 1920 samples per line, the NTSC
// combs at 1920: 125 - 135 instead of 150 - 165 Gpixel/s, by process) is gone; the other shapes +- 1 %.
#ifndef CM_XCD_REMAP
#define CM_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_block(int b, int n) {
#if CM_XCD_REMAP
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
#else
    (void)n;
    return b;
#endif
}
// Which call of the flattened [frame][run][call] list a lane owns, and where its rows live.
struct LaneCall {
    long long frame;
    long long call;     // index of the call in the flattened [frame][run][call] list of the batch (the order the reference makes them in)
    int line, kk, regime, src_row, prev_row, out_row;
    bool store_ok;
};
// c: index of the call in the launch's list (sparse pass: of the run), already known to be wanted or not (active)
__device__ __forceinline__ LaneCall locate_call_at(const Geom &g, long long c, bool active) {
    long long frame;
    int run, i;
    int rem;     // index of the call within its frame
    if (g.sparse) {
        if (c >= g.total_calls) { active = false; c = g.total_calls - 1; }
        frame = c / g.runs_per_frame;
        run = (int)(c - frame * g.runs_per_frame);
        i = 0;
        rem = run ? g.calls_run0 : 0;
    } else {
        if (c >= g.total_calls) active = false;
        if (c < 0) c = 0;
        if (c >= g.total_calls) c = g.total_calls - 1;
        frame = c / g.calls_per_frame;
        rem = (int)(c - frame * g.calls_per_frame);
        run = rem >= g.calls_run0 ? 1 : 0;
        i = rem - (run ? g.calls_run0 : 0);
    }
    LaneCall r;
    r.frame = frame;
    r.call = frame * g.calls_per_frame + rem;
    r.line = (run ? g.first_line[1] : g.first_line[0]) + 2 * i;
    r.kk = g.k0 + i;
    r.regime = r.kk < 2 ? r.kk : 2;
    r.store_ok = active && r.kk >= g.skip_first && (g.keep_calls == 0 || r.kk < g.keep_calls);
    if (g.rows_mode) {
        r.src_row = i;
        r.prev_row = i - 1 < 0 ? i : i - 1;
        r.out_row = i;
    } else {
        r.src_row = r.line;
        if (r.src_row >= g.H) r.src_row -= 2 * ((r.src_row - g.H) / 2 + 1);  // image.py:80-81: step back by 2 until inside
        r.prev_row = r.line - 2;
        if (r.prev_row < 0) r.prev_row = r.src_row;
        if (r.prev_row >= g.H) r.prev_row -= 2 * ((r.prev_row - g.H) / 2 + 1);
        if (g.in_calls) {
            r.src_row = rem;
            r.prev_row = i > 0 ? rem - 1 : rem;
        }
        if (g.out_calls) {
            r.out_row = rem;
        } else {
            r.out_row = r.line - 2 * g.delay;
            r.store_ok = r.store_ok && i >= g.delay && r.out_row >= 0 && r.out_row < g.H;
        }
    }
    return r;
}
// one lane per call, 64 - depth calls per workgroup behind depth halo lanes
__device__ __forceinline__ LaneCall locate_call(const Geom &g, int block, int depth, int lane) {
    if (g.sparse) return locate_call_at(g, (long long)block * 64 + lane, true);
    return locate_call_at(g, (long long)block * (64 - depth) - depth + lane, lane >= depth);
}

// Byte variant: rows of W bytes; a tile row is 32 bytes, 8 lanes x 4 bytes per row, 8 rows per instruction.
__device__ __forceinline__ void fill_tile_u8(const Geom &g, lds_float *itile, const float *xp, int c, int lane) {
    int col = kInTile * c + 4 * (lane & 7);
    if (col > g.W - 4) col = g.W - 4;
#pragma nounroll
    for (int q = 0; q < 8; ++q) {
        const unsigned char *src = (const unsigned char *)ptr_from((8 * q + (lane >> 3)) * 4, xp) + col;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(itile + q * 64), 4, 0, CM_FILL_AUX);
    }
}

template <class Cfg>
__device__ __forceinline__ void run_lane(const Geom &g, const DemodK<float, typename Cfg::S> &k_in, int block, lds_float *lds) {
    typedef DemodLane<Cfg> Lane;
    // hot coefficient blocks live in VGPRs (see cm_stages.h: CM_V_*)
    DemodK<float, typename Cfg::S> k = k_in;
    typedef typename Lane::Front::VP VP;
    if (VP::VT) pin_block(k.taps);
    if (VP::VL) pin_block(k.lpf, true);
    if (VP::VB) pin_block(k.ext, false);
    typedef typename Cfg::S S;
    constexpr int FRONT = Cfg::FRONT, DEPTH = Cfg::DEPTH, kTile = Cfg::TILE;
    constexpr bool BSF = Cfg::BSF;
    lds_float *itile = lds;
    lds_float *otile_base = lds + Cfg::kLdsInF;
    lds_float *yring = lds + Cfg::kLdsInF + Cfg::kLdsOut;

    const int lane = threadIdx.x;
    const LaneCall lc = locate_call(g, block, DEPTH, lane);
    const long long frame = lc.frame;
    const int line = lc.line, regime = lc.regime;
    const int dy = (g.luma_prev_bits >> regime) & 1;
    const int src_row = lc.src_row, luma_row = dy ? lc.prev_row : lc.src_row, out_row = lc.out_row;
    const bool store_ok = lc.store_ok;
    Lane L;
    constexpr bool U8 = Cfg::U8;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const float *xp, *op;
    if (U8) {   // strides count bytes in this mode; the pointers are carried as opaque 64-bit values
        const unsigned char *ib = (const unsigned char *)g.in + frame * g.in_frame_stride;
        xp = (const float *)(ib + (long long)src_row * g.W);
        L.lp = (const float *)(ib + (long long)luma_row * g.W);
        op = store_ok ? (const float *)((unsigned char *)g.out + frame * g.out_frame_stride + (long long)out_row * g.out_row_stride)
                      : nullptr;
    } else {
        xp = g.in + frame * g.in_frame_stride + (long long)src_row * g.Wp;
        L.lp = g.in + frame * g.in_frame_stride + (long long)luma_row * g.Wp;
        op = store_ok ? g.out + frame * g.out_frame_stride + (long long)out_row * g.out_row_stride : nullptr;
    }
    {
        int fmod = (int)((g.first_frame + frame) % g.cycle);
        L.lk = g.lanes[((long long)fmod * 3 + regime) * g.n_lines + line];
        apply_frame_rotation(g, frame, L.lk);
    }
    L.idx1 = ((lane + 63) & 63) * 4;
    L.idx2 = ((lane + 62) & 63) * 4;
    L.front.reset();
    L.back.reset();
    L.base_prev = L.b1_prev = L.b2_prev = Pair<float>{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 14; ++j) L.xw[j] = 0.f;
#pragma unroll
    for (int j = 0; j < (FRONT == FRONT_PALD ? 14 : 1); ++j) L.ew[j] = 0.f;
#pragma unroll
    for (int j = 0; j < (S::SP > 0 ? S::SP : 1); ++j) L.ud[j] = L.vd[j] = 0.f;
    if (BSF) {
        for (int j = 0; j < 16; ++j) yring[j * 64 + lane] = 0.f;
    }
    // output tile address of this lane's row: row * 16 + (column ^ quad swizzle)
    lds_float *otile = U8 ? (lds_float *)((lds_u8 *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;

    // ---- stream geometry ----------------------------------------------------------------------
    const int W = g.W;
    const int lat_front = Lane::Front::latency(k);
    int lat_luma = 0;
    if constexpr (FRONT == FRONT_QAM) lat_luma = Lane::Front::luma_latency(k);
    const int lat_out = lat_front + 1 + (S::RT ? k.s_p : S::SP);     // n7 = t - lat_out
    const int Wp = g.Wp;                          // the output row ends with the quad that holds sample W - 1
    const int T = (Wp + lat_out + 3) & ~3;
    const int front_off = FRONT == FRONT_PALD ? 10 + k.q_e + 9 + 10 : 10 + k.q_e;  // detector sample pair = t - front_off
    int t_mid0 = (lat_out + 3) & ~3;           // every stage index >= 0 from here on
    int t_mid1 = (W - 4) & ~3;                 // bodies below this never touch the end of the row

    const lds_float *xrow = itile + lane * kInTile;
    auto read_x = [&](int first) -> f4 {  // x[first .. first + 3] from the input tile, zero outside the row
        f4 v;
        if (U8)
            v = decode_bytes(*(const lds_u32 *)((const lds_u8 *)itile + lane * kInTile + (first & (kInTile - 1))));
        else
            v = *(const lds_f4 *)(xrow + (first & (kInTile - 1)));
        if (first + 3 >= W) {
            if (first >= W) v.x = 0.f;
            if (first + 1 >= W) v.y = 0.f;
            if (first + 2 >= W) v.z = 0.f;
            if (first + 3 >= W) v.w = 0.f;
        }
        return v;
    };
    auto read_luma = [&](int first, bool check) -> f4 {  // x_l[first .. first + 3], zero outside the row
        if (BSF) return f4{0.f, 0.f, 0.f, 0.f};
        return load_luma<U8>(L.lp, first, check, W);
    };

    if (U8) fill_tile_u8(g, itile, xp, 0, lane); else fill_tile(g, itile, xp, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        f4 x0 = read_x(0);
        L.xw[10] = x0.x; L.xw[11] = x0.y; L.xw[12] = x0.z; L.xw[13] = x0.w;
    }
    f4 nl = read_luma(-lat_out, true);

    // carriers of step base + sub; in the edge-free body the two table pointers are formed once per body
    // and the sub-steps differ by immediate offsets only
    auto carriers = [&](const_f4 *c4, const_f2 *c2, int t, int sub, bool edge, float car[4], float carb[2]) {
        f4 c;
        f2 d;
        if (edge) {
            int nf = t - front_off, nb = t - lat_out;
            nf = nf < 0 ? 0 : (nf > W - 1 ? W - 1 : nf);
            nb = nb < 0 ? 0 : (nb > W - 1 ? W - 1 : nb);
            c = ((const_f4 *)g.carrier4)[nf];
            d = ((const_f2 *)g.carrier2)[nb];
        } else {
            c = c4[sub];
            d = c2[sub];
        }
        car[0] = c.x; car[1] = c.y; car[2] = c.z; car[3] = c.w;
        carb[0] = d.x; carb[1] = d.y;
    };

    // An output tile (or the row) ends at samples n7 = 3 (mod 4) only (W and kTile are multiples of 4),
    // i.e. always after the same sub-step of the 4x unrolled body.
    const int s_flush = (lat_out + 3) & 3;
#ifdef CM_DIAG
    unsigned long long d_flush = 0, d_fill = 0, d_xread = 0, d_luma = 0, d_sub = 0;
    const unsigned long long d_begin = cm_stamp(), d_rbegin = cm_realtime();
#endif
    auto maybe_flush = [&](int t) {
        const int n7 = t - lat_out;
        if (n7 >= 0 && ((n7 & (kTile - 1)) == kTile - 1 || n7 == Wp - 1)) {
            CM_STAMP(t0);
            if (U8) flush_tile_u8(g, otile_base, op, n7 & ~(kTile - 1), lane);
            else flush_tile<kTile>(g, otile_base, op, n7 & ~(kTile - 1), lane);
#ifdef CM_DIAG
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
            CM_ACC(d_flush, t0);
        }
    };

    auto body = [&](int tb, auto edge_tag, FrontLatch<float> &fla, BackLatch<float> &bla) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#ifdef CM_DIAG
        {
            CM_STAMP(t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // how long the luma prefetch still needs
            CM_ACC(d_luma, t0);
        }
#endif
        L.lw = nl;
        const int nxt = tb + 4;
        nl = read_luma(nxt - lat_out, EDGE || nxt >= t_mid1);  // next body's luma: a whole body hides the latency
        CM_STAMP(t_sub);
        float carA[4], carbA[2], carB[4], carbB[2];
        const_f4 *c4 = (const_f4 *)g.carrier4 + (tb - front_off);
        const_f2 *c2 = (const_f2 *)g.carrier2 + (tb - lat_out);
        carriers(c4, c2, tb + 0, 0, EDGE, carA, carbA);
        carriers(c4, c2, tb + 1, 1, EDGE, carB, carbB);
        L.template substep<0, EDGE>(g, k, fla, bla, tb + 0, lat_front, lat_luma, carA, carbA, otile, yring, lane, wpos);
        if (s_flush == 0) maybe_flush(tb + 0);
        carriers(c4, c2, tb + 2, 2, EDGE, carA, carbA);
        L.template substep<1, EDGE>(g, k, fla, bla, tb + 1, lat_front, lat_luma, carB, carbB, otile, yring, lane, wpos);
        if (s_flush == 1) maybe_flush(tb + 1);
        carriers(c4, c2, tb + 3, 3, EDGE, carB, carbB);
        L.template substep<2, EDGE>(g, k, fla, bla, tb + 2, lat_front, lat_luma, carA, carbA, otile, yring, lane, wpos);
        if (s_flush == 2) maybe_flush(tb + 2);
        L.template substep<3, EDGE>(g, k, fla, bla, tb + 3, lat_front, lat_luma, carB, carbB, otile, yring, lane, wpos);
        if (s_flush == 3) maybe_flush(tb + 3);
        CM_ACC(d_sub, t_sub);
#pragma unroll
        for (int j = 0; j < 10; ++j) L.xw[j] = L.xw[j + 4];
        // ---- next body's input from the LDS tile ------------------------------------------------
        if ((nxt & (kInTile - 1)) == 0 && nxt < W) {  // first read of a new tile: its fill was issued a body ago
            CM_STAMP(t0);
#ifndef CM_EXP_NO_FILL_WAIT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __builtin_amdgcn_wave_barrier();
            CM_ACC(d_fill, t0);
        }
        {
            CM_STAMP(t0);
            f4 xn = read_x(nxt);
            L.xw[10] = xn.x; L.xw[11] = xn.y; L.xw[12] = xn.z; L.xw[13] = xn.w;
#ifdef CM_DIAG
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            CM_ACC(d_xread, t0);
        }
        if ((nxt & (kInTile - 1)) == kInTile - 4 && nxt + 4 < W) {  // that was the last read of this tile: refill it
            CM_STAMP(t0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (U8) fill_tile_u8(g, itile, xp, (nxt >> 5) + 1, lane); else fill_tile(g, itile, xp, (nxt >> 5) + 1, lane);
            CM_ACC(d_fill, t0);
        }
        if (FRONT == FRONT_PALD) {
#pragma unroll
            for (int j = 0; j < 10; ++j) L.ew[FRONT == FRONT_PALD ? j : 0] = L.ew[FRONT == FRONT_PALD ? j + 4 : 0];
        }
    };
    // The end-of-row latches are only ever set at t >= W - 1.  When an edge-free region exists they
    // are therefore dead until the tail loop; without one (tiny rows) the tail loop runs everything.
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0;
    int tb = 0;
    {
        FrontLatch<float> fla;
        BackLatch<float> bla;
        fla.reset();
        bla.reset();
        for (; tb < t_mid0; tb += 4) body(tb, std::true_type(), fla, bla);
        for (; tb < t_mid1; tb += 4) body(tb, std::false_type(), fla, bla);
    }
    FrontLatch<float> fla;
    BackLatch<float> bla;
    fla.reset();
    bla.reset();
    for (; tb < T; tb += 4) body(tb, std::true_type(), fla, bla);
#ifdef CM_DIAG
    if (g.diag && lane == 0 && !g.sparse) {
        unsigned long long *d = g.diag + 8ull * block;
        d[0] = cm_stamp() - d_begin; d[1] = d_flush; d[2] = d_fill; d[3] = d_xread; d[4] = d_luma; d[5] = d_sub;
        d[6] = cm_realtime() - d_rbegin;
    }
#endif
}

// =============================================================================================
// Wave-pair driver: one 128-thread workgroup = two wavefronts that walk the same 64 calls.
//   wave 0 (stage A)  owns every global LOAD: the input tile and the luma source samples (second visit of the own or
//                     the previous row), which it leaves in LDS for B; runs the front end up to the 2x-rate pair the
//                     product detectors multiply (cm_stages.h: StageA); every 4 steps it leaves the 4 pairs of each
//                     lane in an LDS ring
//   wave 1 (stage B)  picks them up one block later: detectors, low-passes, decimators, comb combination, back end,
//                     output tile and every global STORE.  vmcnt counts loads and stores of a wave in one queue, so a
//                     wave that does both drains its stores each time it waits for a load (0.5 ms of 3.0 per 1000
//                     frames, profiles/r01_pair_notes.md); B never waits on vmcnt
// The halves keep 110-130 VGPRs each instead of 250 for the whole line, so 3-4 waves share a SIMD instead of 2 and
// a wave that waits (LDS round trips, tile refills, flushes, scalar loads) no longer idles the vector pipe.
// Synchronisation: the ring is double-buffered and both waves execute one s_barrier per block - A after writing
// block i, B before reading it - so A runs at most one block ahead and B never reads a block that is not complete.
// =============================================================================================
constexpr int kMidRing = 2 * 2 * 64 * 4;    // floats: [buffer][even | odd][lane][4 steps]
// 1: the instances on the QAM front end without band-stop luma (the combs: NtscComb, Pal3D, SimpleComb, Simple3DComb ...)
// cut the pair BEHIND the detector low-pass: stage A there is only up2 + band-pass (about 45 FMA-equivalents per pixel
// against 145 in stage B), so it takes the detector products and the packed low-pass as well (52 more) and hands over the
// two low-passed pairs instead of one 2x-rate pair
#ifndef CM_QAM_LPF_IN_A
#define CM_QAM_LPF_IN_A 1
#endif
// band-stop luma ring of the pair kernels, written by A and read by B (lat_out - lat_luma + up to 12) steps later: 32 slots
// for the tuned shapes, 64 for the run-time shape (high sampling rates; it runs 4 workgroups per CU, so the LDS is there)
template <class S> constexpr int luma_ring_slots() { return S::YS; }

#ifndef CM_QAM_SHORT_RING
#define CM_QAM_SHORT_RING 1
#endif
#ifndef CM_RT_UV_RING
#define CM_RT_UV_RING 1
#endif
constexpr int kLumaSlots = 2 * 64 * 4;      // floats: [buffer][lane][4 steps] luma source samples fetched by A for B
// CM_LUMA_RING: blocks of [lane][4 steps] x samples; A writes x[tb - 10 + o .. + 3] at the end of body tb, B reads the block
// m bodies later, lat_out = 4 m + 10 - o; A runs at most two blocks ahead of B's read, so m + 2 blocks are live
// (the host checks lat_out against this: cm_api.hip).  11 KiB for PAL-BG (lat_out 46 = the limit), 12 KiB for the order-6 band-pass shapes, 20 KiB for the run-time shape.
// The QAM front end is shorter (lat_out 26 / 28 / 32 for the PAL-BG / NTSC / NTSC-A shapes against 46 / 47 behind the PAL-D
// front end): its ring is sized for that, so that those instances fit six workgroups per CU as well (25 instead of 30 KiB).
// X: extra steps of output latency (PassCfg::WRAP: the second exchange of the two-level combs)
template <class S, int FRONT = 0, bool LC = false, int X = 0> constexpr int luma_delay_blocks() {
    if (S::DYN) return 20;     // sized at launch (pair_lds_floats): this is the limit
    if (FRONT == FRONT_QAM && CM_QAM_SHORT_RING) return ((S::NE >= 4 ? 32 : (S::NE == 3 ? 28 : 26)) + X - 10 - ring_window<S, FRONT, LC>() + 3) / 4 + 2;
    return (S::NE >= 3 ? 12 : 11 - ring_window<S, FRONT, LC>() / 4) + (X + 3) / 4;
}
template <class S, int FRONT = 0, bool LC = false, int X = 0> constexpr int luma_delay_max_latency() {
    return 4 * (luma_delay_blocks<S, FRONT, LC, X>() - 2) + 10 + ring_window<S, FRONT, LC>();
}

template <class Cfg>
struct PairLds {
    static constexpr int kIn = Cfg::kPairLdsIn, kOut = Cfg::kLdsOut;
    static constexpr int kY = Cfg::BSF ? luma_ring_slots<typename Cfg::S>() * 64
                                       : (CM_LUMA_RING && !Cfg::kNoRing ? luma_delay_blocks<typename Cfg::S, Cfg::FRONT, Cfg::kLcutCfg, Cfg::WRAP ? 1 : 0>() * 256 : kLumaSlots);
    // hand-over ring: the 2x-rate pair (even, odd) per step, or - where stage A also takes the detector products and the
    // low-pass (LCUT: the QAM front end without the band-stop luma) - the two low-passed pairs (q_e, q_o)
    static constexpr bool kLcut = Cfg::kLcutCfg;
    static constexpr int kMid = kLcut ? 2 * kMidRing : kMidRing;
    // run-time shape: the pre-correction delay of (u, v) (up to 12 steps, known at run time only) is an LDS ring of 16 slots
    // instead of a register window with a chain of uniform selects (24 registers, 11 selects and 11 packed moves per step)
    static constexpr int kUv = (Cfg::S::RT && CM_RT_UV_RING) ? 16 * 64 * 2 : 0;
    static constexpr int kFloats = kIn + kMid + kOut + kUv + kY;
};
template <>
struct PairLds<NoPass> {
    static constexpr int kFloats = 0;
};
// LDS floats of one pass of the pair kernel (the launch passes the larger of main and first pass as dynamic LDS): the tuned
// shapes' constant, or - run-time shape, luma delay ring - what the plan's latency needs (run_pair: kLB = lr_m + 2 blocks)
template <class Cfg>
inline int pair_lds_floats(const DemodK<float, typename Cfg::S> &k) {
    typedef typename Cfg::S S;
    if (!S::DYN || Cfg::BSF || !CM_LUMA_RING || Cfg::kNoRing) return PairLds<Cfg>::kFloats;
    const int lat_front = Cfg::FRONT == FRONT_PALD ? 10 + k.q_e + 9 + 10 + k.q_l + 9 : 10 + k.q_e + k.q_l + 9;
    const int lat_out = lat_front + 1 + k.s_p + (Cfg::WRAP ? 1 : 0);
    constexpr int kWinX = ring_window<S, Cfg::FRONT, Cfg::kLcutCfg>();
    const int lr_o = (10 + kWinX - lat_out) & 3, lr_m = (lat_out - 10 - kWinX + lr_o) >> 2;
    return PairLds<Cfg>::kIn + PairLds<Cfg>::kMid + PairLds<Cfg>::kOut + PairLds<Cfg>::kUv + (lr_m + 2) * 256;
}

#ifdef CM_DIAG
// diagnostic builds: cycles spent waiting in the barrier are summed into acc
#define PAIR_BARRIER(acc)                                                        \
    do {                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        unsigned long long t0_ = cm_stamp();                                     \
        asm volatile("s_barrier" ::: "memory");                                  \
        acc += cm_stamp() - t0_;                                                 \
    } while (0)
#else
#define PAIR_BARRIER(acc) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// 1: stage A of the PAL-D front end runs two of its three half-band chains as one packed chain (PalDFrontAPk)
#ifndef CM_PALD_PK_FRONT
#define CM_PALD_PK_FRONT 1
#endif
// 1: ... and the third one (up2 of e) two steps at a time, packed along the accumulator index (HalfbandUp2Pk)
#ifndef CM_PALD_UP2
#define CM_PALD_UP2 1
#endif
#ifndef CM_SEGMENTS          /* 0: never cut rows into segments (host: cm_api.hip: segment_geometry) */
#define CM_SEGMENTS 1
#endif
template <class Cfg>
__device__ __forceinline__ void run_pair(const Geom &g, const DemodK<float, typename Cfg::S> &k_in, int block, lds_float *lds,
                                         int role) {
    typedef typename Cfg::S S;
    typedef DemodK<float, S> K;
    constexpr int FRONT = Cfg::FRONT, DEPTH = Cfg::DEPTH, kTile = Cfg::TILE, SP = S::SP;
    constexpr bool BSF = Cfg::BSF, U8 = Cfg::U8, PALD = FRONT == FRONT_PALD;
    constexpr int kYSlots = luma_ring_slots<S>();
    constexpr bool LRING = CM_LUMA_RING && !BSF && !Cfg::kNoRing;      // luma source samples through the LDS delay ring
    constexpr int kIT = Cfg::kPairInTile;                // samples per input tile row
    constexpr int kLBmax = luma_delay_blocks<S, FRONT, Cfg::kLcutCfg, Cfg::WRAP ? 1 : 0>();   // tuned shapes: the ring's size; run-time shape: its limit
    constexpr int kWinX = ring_window<S, FRONT, Cfg::kLcutCfg>();        // extra x samples stage A keeps behind its window
    typedef typename std::conditional<PALD, PalDFront<float, S>, QamFront<float, S, BSF>>::type Front;
    typedef typename Front::StageA StageA;
    typedef typename Front::StageB StageB;
    typedef typename Front::VP VP;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) unsigned lds_u32;

    lds_float *itile = lds;
    lds_float *ring = lds + PairLds<Cfg>::kIn;
    constexpr bool LCUT = PairLds<Cfg>::kLcut;
    constexpr int kMid = PairLds<Cfg>::kMid;
    lds_float *otile_base = ring + kMid;
    lds_float *uvring = otile_base + PairLds<Cfg>::kOut;            // run-time shape only (PairLds::kUv)
    lds_float *yring = uvring + PairLds<Cfg>::kUv;

    const int lane = threadIdx.x & 63;
    int seg = 0;
    if (g.seg_len) {
        seg = block / g.seg_blocks;
        block -= seg * g.seg_blocks;
    }
    const LaneCall lc = locate_call(g, block, Cfg::HALO, lane);
    const long long frame = lc.frame;
    const int regime = lc.regime;

    // ---- stream geometry (identical in both waves) ---------------------------------------------
    K k = k_in;
    const int W = g.W;
    const int lat_front = Front::latency(k);
    int lat_luma = 0;
    if constexpr (!PALD) lat_luma = Front::luma_latency(k);
    const int sp = S::RT ? k.s_p : SP;          // run-time shapes: SP is the window size, k.s_p the delay
    const int lat_out = lat_front + 1 + sp + (Cfg::WRAP ? 1 : 0);     // n7 = t - lat_out (WRAP: one more step for the second exchange)
    const int Wp = g.Wp;                          // the output row ends with the quad that holds sample W - 1
    const int T = (Wp + lat_out + 3) & ~3;
    const int front_off = StageA::pair_offset(k);   // detector pair index nd = t - front_off
    int t_mid0 = (lat_out + 3) & ~3;           // every stage index >= 0 from here on
    int t_mid1 = (W - 4) & ~3;                 // bodies below this never touch the end of the row
    if (t_mid1 <= t_mid0) t_mid0 = t_mid1 = 0; // tiny rows: the guarded body runs everything
#ifdef CM_EXP_ALL_EDGE   /* timing experiment: every body is a guarded one */
    t_mid0 = t_mid1 = 0;
#endif
    // Row segments (small batches, Geom::seg_len): this workgroup produces the output samples [x_lo, x_hi) only.  It enters
    // the stream at tb0 = x_lo - seg_warm (on an input tile boundary) with every filter state zero - the start-of-row state when
    // tb0 = 0, else a state the recursive filters have forgotten by x_lo (to 1e-8) - and leaves it once sample x_hi - 1 is out.
#if CM_SEGMENTS
    int x_lo = 0, tb0 = 0, T_end = T;
    if (g.seg_len) {
        x_lo = seg * g.seg_len;
        const int x_hi = x_lo + g.seg_len;
        tb0 = x_lo - g.seg_warm;
        tb0 = tb0 < 0 ? 0 : tb0 & ~(kIT - 1);
        if (x_hi < Wp) T_end = (x_hi - 1 + lat_out + 4) & ~3;
    }
#else     /* -DCM_SEGMENTS=0: whole rows only (A/B of what the segment windows cost the large-batch kernel) */
    const int x_lo = 0, tb0 = 0, T_end = T;
    (void)seg;
#endif

#ifdef CM_DIAG
    unsigned long long d_bar = 0, d_flush = 0, d_other = 0;
    const unsigned long long d_begin = cm_stamp(), d_rbegin = cm_realtime();
#endif
    lds_float *lring = yring;   // luma hand-off slots share the place of the band-stop ring (the two exclude each other)
    // delay ring: A leaves x[tb - 10 + lr_o .. + 3] at the end of body tb, B needs x_l[tb - lat_out .. + 3] at the start of
    // body tb, i.e. the block A wrote lr_m bodies before: lat_out = 4 lr_m + 10 - lr_o
    const int lr_o = (10 + kWinX - lat_out) & 3, lr_m = (lat_out - 10 - kWinX + lr_o) >> 2;
    // the run-time shape sizes its ring (the last region of the dynamic LDS) by the plan's latency: pair_lds_floats()
    const int kLB = S::DYN ? lr_m + 2 : kLBmax;

    if (role == 0) {
        // =================================== stage A ===========================================
        const int luma_row = ((g.luma_prev_bits >> regime) & 1) ? lc.prev_row : lc.src_row;
        const float *lp;
        if (U8) lp = (const float *)((const unsigned char *)g.in + frame * g.in_frame_stride + (long long)luma_row * g.W);
        else lp = g.in + frame * g.in_frame_stride + (long long)luma_row * g.Wp;
        // PAL-D front: two of the three chains packed (cm_stages_pk.h: PalDFrontAPk); its interpolator is fed x[t + 1]
        constexpr bool PKF = CM_PALD_PK_FRONT != 0 && (PALD || BSF);
        TapsPk tkp;
        TapsPkOdd tko;
        Taps<float> tks;
        constexpr bool UP2 = PKF && PALD && CM_PALD_UP2 != 0 && (S::NE < 3 || (S::WIDE && CM_WIDE_PALD_UP2 != 0));   // (the three-section band-pass has no registers for it at 3 waves per SIMD: it would spill)
        if constexpr (UP2) tko.load(k.taps);
        if constexpr (PKF) {
            tkp.load(k.taps);
#pragma unroll
            for (int i = 0; i < 10; ++i) tks.c[i] = (i & 1) ? tkp.c2[i >> 1].y : tkp.c2[i >> 1].x;
            tks.c0 = tkp.c0.x;
        } else {
            if (VP::VT) pin_block(k.taps);
        }
        if (VP::VB) pin_block(k.ext, false);
        const float *xp;
        if (U8) xp = (const float *)((const unsigned char *)g.in + frame * g.in_frame_stride + (long long)lc.src_row * g.W);
        else xp = g.in + frame * g.in_frame_stride + (long long)lc.src_row * g.Wp;
        typename std::conditional<PKF, typename std::conditional<PALD, PalDFrontAPk<S>, QamBsfFrontAPk<S>>::type, StageA>::type fa;
        fa.reset();
        // LCUT: detector products and low-pass in this stage
        SosPk<S::NL> a_lpk;
        DetectorLpfPk<S> a_lpf;
        pf2 a_p_last = {0.f, 0.f};
        if constexpr (LCUT) {
            a_lpk.load(k.lpf, false);
            a_lpf.reset();
        }
        f4 xq = {0.f, 0.f, 0.f, 0.f};          // PKF: x[tb + 4 .. tb + 7], read at the start of a body
        float xw[14], ew[PALD ? 14 : 1];
        float xo[kWinX >= 8 ? kWinX : 1];        // x[tb - 10 - kWinX ..]: older than xw
#pragma unroll
        for (int j = 0; j < (kWinX >= 8 ? kWinX : 1); ++j) xo[j] = 0.f;
#pragma unroll
        for (int j = 0; j < 14; ++j) xw[j] = 0.f;
#pragma unroll
        for (int j = 0; j < (PALD ? 14 : 1); ++j) ew[j] = 0.f;
        if (BSF) {
            for (int j = 0; j < kYSlots; ++j) yring[j * 64 + lane] = 0.f;
        }
        const lds_float *xrow = itile + lane * kIT;
        constexpr bool TT = Cfg::kPairTwoTiles;     // two tiles [64][kIT] side by side: tile t lives in half t & 1
        auto read_x = [&](int first) -> f4 {  // x[first .. first + 3] from the input tile, zero outside the row
            f4 v;
            if (U8)
                v = decode_bytes(*(const lds_u32 *)((const lds_u8 *)itile + lane * kIT + (first & (kIT - 1))));
            else if (TT)
                v = *(const lds_f4 *)(xrow + ((first / kIT) & 1) * (64 * kIT) + (first & (kIT - 1)));
            else
                v = *(const lds_f4 *)(xrow + (first & (kIT - 1)));
            if (first + 3 >= W) {
                if (first >= W) v.x = 0.f;
                if (first + 1 >= W) v.y = 0.f;
                if (first + 2 >= W) v.z = 0.f;
                if (first + 3 >= W) v.w = 0.f;
            }
            return v;
        };
        auto read_luma = [&](int first, bool check) -> f4 {  // x_l[first .. first + 3], zero outside the row
            if (BSF || LRING) return f4{0.f, 0.f, 0.f, 0.f};
            return load_luma<U8>(lp, first, check, W);
        };
        if (U8) fill_tile_u8(g, itile, xp, tb0 / kInTile, lane);
        else fill_tile<kIT>(g, itile + (TT ? ((tb0 / kIT) & 1) * (64 * kIT) : 0), xp, tb0 / kIT, lane);
        f4 lum_cur = read_luma(tb0 - lat_out, true);   // luma source of B's first block
        if (LRING) {   // blocks B reads before A has written them lie before the row: zeros
            for (int j = 0; j < kLB; ++j) *(lds_f4 *)(lring + j * 256 + lane * 4) = f4{0.f, 0.f, 0.f, 0.f};
        }
        int lr_w = 0;   // ring block of this body
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            f4 x0 = read_x(tb0);
            xw[10] = x0.x; xw[11] = x0.y; xw[12] = x0.z; xw[13] = x0.w;
            if constexpr (PKF) fa.prime(tkp, x0.x);
        }
        // two tiles: the first read of tile t asks for tile t + 1 (the other half: its last read was a body ago)
        auto next_tile = [&](int first) {
            if (first + kIT < W && first + kIT < T_end) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                fill_tile<kIT>(g, itile + ((first / kIT + 1) & 1) * (64 * kIT), xp, first / kIT + 1, lane);
            }
        };
        if constexpr (TT) next_tile(tb0);
        auto sub_a = [&](auto sub_tag, auto edge_tag, FrontLatch<float> &fla, int tau, float &m_even, float &m_odd) {
            constexpr int SUB = decltype(sub_tag)::value;
            constexpr bool EDGE = decltype(edge_tag)::value;
            Mid<float> m;
            if constexpr (PKF && PALD) {
                float e_out;
                const float x_next = SUB < 3 ? xw[SUB < 3 ? 11 + SUB : 0] : xq.x;
                m = fa.template step<EDGE>(k, tkp, tks, fla, tau, x_next, xw[SUB], ew[PALD ? SUB : 0], e_out);
                ew[PALD ? 10 + SUB : 0] = e_out;
            } else if constexpr (PKF) {
                float luma_bsf = 0.f;
                const float x_next = SUB < 3 ? xw[SUB < 3 ? 11 + SUB : 0] : xq.x;
                m = fa.template step<EDGE>(k, tkp, fla, tau, x_next, xw[SUB], luma_bsf);
                yring[((tau - lat_luma) & (kYSlots - 1)) * 64 + lane] = luma_bsf;
            } else if constexpr (PALD) {
                float e_out;
                m = fa.template step<EDGE>(k, fla, tau, xw[10 + SUB], xw[SUB], ew[PALD ? SUB : 0], e_out);
                ew[PALD ? 10 + SUB : 0] = e_out;
            } else {
                float luma_bsf = 0.f;
                m = fa.template step<EDGE>(k, fla, tau, xw[10 + SUB], xw[SUB], luma_bsf);
                if (BSF) yring[((tau - lat_luma) & (kYSlots - 1)) * 64 + lane] = luma_bsf;
            }
            m_even = m.even;
            m_odd = m.odd;
        };
        auto body_a = [&](int tb, auto edge_tag, FrontLatch<float> &fla) {
            constexpr bool EDGE_A = decltype(edge_tag)::value;
            // luma source of B's next block: a whole body hides the latency
            const f4 lum_next = read_luma(tb + 4 - lat_out, EDGE_A || tb + 4 >= t_mid1);
            f4 lum_blk = {0.f, 0.f, 0.f, 0.f};
            if (LRING) {   // x[tb - 10 + lr_o .. + 3] out of the window (xw[j] = x[tb - 10 + j], zero outside the row)
                const float *xs = kWinX >= 8 ? xo : xw;     // the oldest samples of the window (kWinX = 0 or >= 8)
                lum_blk = lr_o == 0 ? f4{xs[0], xs[1], xs[2], xs[3]}
                        : lr_o == 1 ? f4{xs[1], xs[2], xs[3], xs[4]}
                        : lr_o == 2 ? f4{xs[2], xs[3], xs[4], xs[5]} : f4{xs[3], xs[4], xs[5], xs[6]};
            }
            const int nxt = tb + 4;
            if constexpr (PKF) {   // the next quad now (the interpolator wants x[tb + 4] in the last sub-step); same tile protocol
                if ((nxt & (kIT - 1)) == 0 && nxt < W) {
                    CM_STAMP(t0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    CM_ACC(d_other, t0);
                }
                xq = read_x(nxt);
                if constexpr (TT) {
                    if ((nxt & (kIT - 1)) == 0 && nxt < W) next_tile(nxt);
                } else
                if ((nxt & (kIT - 1)) == kIT - 4 && nxt + 4 < W && nxt + 4 < T_end) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (U8) fill_tile_u8(g, itile, xp, nxt / kIT + 1, lane); else fill_tile<kIT>(g, itile, xp, nxt / kIT + 1, lane);
                }
            }
            float me[4], mo[4];
            if constexpr (UP2) {
                Mid<float> m0, m1, m2, m3;
                fa.template step2<EDGE_A>(k, tkp, tko, tks, fla, tb, xw[11], xw[12], xw[0], xw[1], ew[PALD ? 0 : 0], ew[PALD ? 1 : 0],
                                          ew[PALD ? 10 : 0], ew[PALD ? 11 : 0], m0, m1);
                fa.template step2<EDGE_A>(k, tkp, tko, tks, fla, tb + 2, xw[13], xq.x, xw[2], xw[3], ew[PALD ? 2 : 0], ew[PALD ? 3 : 0],
                                          ew[PALD ? 12 : 0], ew[PALD ? 13 : 0], m2, m3);
                me[0] = m0.even; mo[0] = m0.odd; me[1] = m1.even; mo[1] = m1.odd;
                me[2] = m2.even; mo[2] = m2.odd; me[3] = m3.even; mo[3] = m3.odd;
            } else {
            sub_a(std::integral_constant<int, 0>(), edge_tag, fla, tb + 0, me[0], mo[0]);
            sub_a(std::integral_constant<int, 1>(), edge_tag, fla, tb + 1, me[1], mo[1]);
            sub_a(std::integral_constant<int, 2>(), edge_tag, fla, tb + 2, me[2], mo[2]);
            sub_a(std::integral_constant<int, 3>(), edge_tag, fla, tb + 3, me[3], mo[3]);
            }
#if defined(CM_EXPERIMENTS) && defined(CM_EXP_NO_WINDOW_MOVES)   /* timing experiment (results wrong): the register windows never move - the
            ceiling of what a rotating-index unroll of stage A could save (profiles/r05_headline_residue.txt) */
#pragma unroll
            for (int j = 0; j < 10; ++j) asm volatile("" : "+v"(xw[j]));
#else
            if (kWinX >= 8) {
#pragma unroll
                for (int j = 0; j + 4 < kWinX; ++j) xo[kWinX >= 8 ? j : 0] = xo[kWinX >= 8 ? j + 4 : 0];
#pragma unroll
                for (int j = 0; j < 4; ++j) xo[kWinX >= 8 ? kWinX - 4 + j : 0] = xw[j];
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) xw[j] = xw[j + 4];
#endif
            if constexpr (PKF) {
                xw[10] = xq.x; xw[11] = xq.y; xw[12] = xq.z; xw[13] = xq.w;
            } else {
                if ((nxt & (kIT - 1)) == 0 && nxt < W) {  // first read of a new tile: its fill was issued a body ago
                    CM_STAMP(t0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    CM_ACC(d_other, t0);
                }
                {
                    f4 xn = read_x(nxt);
                    xw[10] = xn.x; xw[11] = xn.y; xw[12] = xn.z; xw[13] = xn.w;
                }
                if constexpr (TT) {
                    if ((nxt & (kIT - 1)) == 0 && nxt < W) next_tile(nxt);
                } else
                if ((nxt & (kIT - 1)) == kIT - 4 && nxt + 4 < W && nxt + 4 < T_end) {  // that was the last read of this tile: refill it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (U8) fill_tile_u8(g, itile, xp, nxt / kIT + 1, lane); else fill_tile<kIT>(g, itile, xp, nxt / kIT + 1, lane);
                }
            }
#if defined(CM_EXPERIMENTS) && defined(CM_EXP_NO_WINDOW_MOVES)
            if (PALD) {
#pragma unroll
                for (int j = 0; j < 10; ++j) asm volatile("" : "+v"(ew[PALD ? j : 0]));
            }
#else
            if (PALD) {
#pragma unroll
                for (int j = 0; j < 10; ++j) ew[PALD ? j : 0] = ew[PALD ? j + 4 : 0];
            }
#endif
            lds_float *slot = ring + ((tb >> 2) & 1) * (kMid / 2) + lane * 4;
            if constexpr (LCUT) {
                // products with the phase-free carriers of the four pairs nd = tb - front_off + s, low-pass on (cos, sin)
                const f16u c4a = *(const_f16 *)(g.carrier4 + 4 * (long long)(tb - front_off));
                pf2 qe[4], qo[4];
                {
                    const pf2 m0 = {me[0], mo[0]}, m1 = {me[1], mo[1]}, m2 = {me[2], mo[2]}, m3 = {me[3], mo[3]};
                    a_lpf.template step<EDGE_A>(k, a_lpk, a_p_last, tb + 0 - front_off, pk_mul_bs<0>(m0, pf2{c4a[0], c4a[1]}),
                                                pk_mul_bs<1>(m0, pf2{c4a[2], c4a[3]}), qe[0], qo[0]);
                    a_lpf.template step<EDGE_A>(k, a_lpk, a_p_last, tb + 1 - front_off, pk_mul_bs<0>(m1, pf2{c4a[4], c4a[5]}),
                                                pk_mul_bs<1>(m1, pf2{c4a[6], c4a[7]}), qe[1], qo[1]);
                    a_lpf.template step<EDGE_A>(k, a_lpk, a_p_last, tb + 2 - front_off, pk_mul_bs<0>(m2, pf2{c4a[8], c4a[9]}),
                                                pk_mul_bs<1>(m2, pf2{c4a[10], c4a[11]}), qe[2], qo[2]);
                    a_lpf.template step<EDGE_A>(k, a_lpk, a_p_last, tb + 3 - front_off, pk_mul_bs<0>(m3, pf2{c4a[12], c4a[13]}),
                                                pk_mul_bs<1>(m3, pf2{c4a[14], c4a[15]}), qe[3], qo[3]);
                }
                *(lds_f4 *)slot = f4{qe[0].x, qe[1].x, qe[2].x, qe[3].x};
                *(lds_f4 *)(slot + 256) = f4{qe[0].y, qe[1].y, qe[2].y, qe[3].y};
                *(lds_f4 *)(slot + 512) = f4{qo[0].x, qo[1].x, qo[2].x, qo[3].x};
                *(lds_f4 *)(slot + 768) = f4{qo[0].y, qo[1].y, qo[2].y, qo[3].y};
            } else {
                *(lds_f4 *)slot = f4{me[0], me[1], me[2], me[3]};
                *(lds_f4 *)(slot + 256) = f4{mo[0], mo[1], mo[2], mo[3]};
            }
            if (LRING) {
                *(lds_f4 *)(lring + lr_w * 256 + lane * 4) = lum_blk;
                lr_w = lr_w + 1 == kLB ? 0 : lr_w + 1;
            } else if (!BSF) {
                *(lds_f4 *)(lring + ((tb >> 2) & 1) * (kLumaSlots / 2) + lane * 4) = lum_cur;
            }
            lum_cur = lum_next;
            PAIR_BARRIER(d_bar);
        };
        int tb = tb0;
        {
            FrontLatch<float> fla;
            fla.reset();
            for (; tb < t_mid0 && tb < T_end; tb += 4) body_a(tb, std::true_type(), fla);
            for (; tb < t_mid1 && tb < T_end; tb += 4) body_a(tb, std::false_type(), fla);
        }
        FrontLatch<float> fla;
        fla.reset();
        for (; tb < T_end; tb += 4) body_a(tb, std::true_type(), fla);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no tile fill may be in flight when the workgroup's LDS is released
#ifdef CM_DIAG
        if (g.diag && lane == 0 && !g.sparse) {
            unsigned long long *d = g.diag + 16ull * block;
            d[0] = cm_stamp() - d_begin; d[1] = d_bar; d[2] = d_other;
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            d[4] = hw;
        }
#endif
        return;
    }

    // ======================================= stage B ===========================================
    // packed float32 throughout (cm_stages_pk.h): pairs (cos, sin) up to the base pair, (u, v) behind it
    StageBK<S> kb;
    kb.load(k, !LCUT);
    const float *op;   // U8: strides count bytes; the pointer is carried as an opaque 64-bit value
    if (U8) op = lc.store_ok ? (const float *)((unsigned char *)g.out + frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride)
                             : nullptr;
    else op = lc.store_ok ? g.out + frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride : nullptr;
    LaneKPk lk;
    {
        int fmod = (int)((g.first_frame + frame) % g.cycle);
        LaneK<float> l1 = g.lanes[((long long)fmod * 3 + regime) * g.n_lines + lc.line];
        apply_frame_rotation(g, frame, l1);
        lk.load(l1, DEPTH, Cfg::MINAVG);
    }
    const int idx1 = ((lane + 63) & 63) * 4, idx2 = ((lane + 62) & 63) * 4;
    DetectorPk<S> det;
    DemodBackPk<S, DEPTH, Cfg::NOTCH, Cfg::MINAVG> back;
    det.reset();
    back.reset();
    pf2 base_prev = {0.f, 0.f}, b1_prev = {0.f, 0.f}, b2_prev = {0.f, 0.f};
    pf2 uv_own = {0.f, 0.f}, uv_nb = {0.f, 0.f};     // WRAP: the inner decoder's (u, v) of the step before - this lane's, the previous call's
    const bool wrap_first = regime == 0;            // comb.py:97-99: the first call of a run returns the inner result as it is
    const bool wrap_min = g.wrap_mode == 2;
    constexpr bool UVR = PairLds<Cfg>::kUv != 0;
    // (u, v) of the last SP steps.  SP <= 2 (the 13.5 MHz shapes): a window shifted by one every step (newest first).  Longer delays (the wide
    // rasters: 3 .. 9 steps) would pay SP packed moves per step for that; there the window is kUvBanks banks of four slots indexed by the
    // sub-step, a value moves one bank down when its slot is written again, and the tap SP steps back is a compile-time (bank, slot) of the
    // sub-step: (SP - 1) / 4 moves per step - none up to SP = 4.  The same values either way.
    constexpr bool UVB = !UVR && !S::RT && SP >= 3;
    constexpr int kUvBanks = UVB ? (SP - 1) / 4 + 1 : 1;
    constexpr int kUvd = UVR ? 1 : (UVB ? 4 * kUvBanks : (SP > 0 ? SP : 1));
    pf2 uvd[kUvd];   // UVR: in LDS instead
#pragma unroll
    for (int j = 0; j < kUvd; ++j) uvd[j] = pf2{0.f, 0.f};
    typedef __attribute__((address_space(3))) pf2 lds_pf2;
    lds_pf2 *uvr = (lds_pf2 *)uvring + lane;
    if constexpr (UVR) {
#pragma unroll
        for (int j = 0; j < 16; ++j) uvr[j * 64] = pf2{0.f, 0.f};
    }
    // output tile address of this lane's row: row * 16 + (column ^ quad swizzle)
    lds_float *otile = U8 ? (lds_float *)((lds_u8 *)otile_base + lane * 3 * kTile) : otile_base + lane * kTile;
    const int wpos = ((lane >> CM_TILE_SWZ) & (kTile / 4 - 1)) << 2;


    // Carriers of one body, fetched a body ahead (the tables are padded, so no index needs a clamp):
    //   c4[0 .. 15] = {C, S}(2 nd), {C, S}(2 nd + 1) for the four detector pairs nd = tb - front_off + sub
    //   c2[0 .. 7]  = {C, S}(2 n7) for the four back-end samples n7 = tb - lat_out + sub
    auto load_c4 = [&](int tb) -> f16u { return *(const_f16 *)(g.carrier4 + 4 * (long long)(tb - front_off)); };
    auto load_c2 = [&](int tb) -> f8u { return *(const_f8 *)(g.carrier2 + 2 * (long long)(tb - lat_out)); };
    const int s_flush = (lat_out + 3) & 3;
    auto maybe_flush = [&](int t) {
        const int n7 = t - lat_out;
        if (n7 >= x_lo && ((n7 & (kTile - 1)) == kTile - 1 || n7 == Wp - 1)) {      // (x_lo = 0 unless the row is cut into segments)
            CM_STAMP(t0);
            if (U8) flush_tile_u8(g, otile_base, op, n7 & ~(kTile - 1), lane);
            else flush_tile<kTile>(g, otile_base, op, n7 & ~(kTile - 1), lane);
            CM_ACC(d_flush, t0);
        }
    };
    f4 lw;
    // delay ring: the block of body tb is lr_m blocks behind A's; a luma source row other than the own one (decoders with a
    // line of delay) is the row of the previous call, i.e. of the neighbouring lane
    int lr_r = lr_m == 0 ? 0 : kLB - lr_m;
    // (calls back, never beyond the run's first call; at the bottom edge consecutive calls re-feed one row - image.py:79-81 - which makes
    // no difference one call back and all the difference two calls back)
    int luma_back = DEPTH >= 1 ? ((g.luma_prev_bits >> regime) & 1) + ((g.luma_prev_bits >> (8 + regime)) & 1) : 0;
    if (luma_back > lc.kk) luma_back = lc.kk;
    const lds_float *lr_lane = lring + ((lane + 64 - luma_back) & 63) * 4;
    // p_e, p_o: detector products of this step's pair; sc: (sn, cs) of the back-end sample
    auto sub_b = [&](auto sub_tag, auto edge_tag, pf2 &p_last, pf2 &uv_last, int tau, pf2 p_e, pf2 p_o, pf2 sc) {
        constexpr int SUB = decltype(sub_tag)::value;
        constexpr bool EDGE = decltype(edge_tag)::value;
        pf2 base;
        if constexpr (LCUT) base = det.dn.push_pair(kb.taps, p_e, p_o);      // (p_e, p_o) = stage A's low-passed pairs
        else base = det.template step<EDGE>(k, kb, p_last, tau - front_off, p_e, p_o);
        // the back end handles the PREVIOUS step's base pair: its neighbours were requested then
        const int n6 = tau - lat_front - 1 - (Cfg::WRAP ? 1 : 0), n7 = n6 - sp;
        pf2 uv = back.combine(lk, base_prev, b1_prev, b2_prev);
        if constexpr (Cfg::WRAP) {
            // second level, one step behind the first: avg / minavg of the previous call's result (asked of the neighbouring lane a step
            // ago) and this call's, comb.py:103-104; a run's first call passes its own through (avg(x, x) = minavg(x, x) = x exactly)
            const pf2 inner = uv;
            const pf2 last = wrap_first ? uv_own : uv_nb;
            if (wrap_min) uv = pf2{minavg_(last.x, uv_own.x), minavg_(last.y, uv_own.y)};
            else uv = pf2{0.5f * (last.x + uv_own.x), 0.5f * (last.y + uv_own.y)};
            uv_own = inner;
            uv_nb = pf2{lane_from(idx1, inner.x), lane_from(idx1, inner.y)};
        }
        base_prev = base;
        if (DEPTH >= 1) b1_prev = pf2{lane_from(idx1, base.x), lane_from(idx1, base.y)};
        if (DEPTH >= 2) b2_prev = pf2{lane_from(idx2, base.x), lane_from(idx2, base.y)};
        float y_src;
        if (BSF) y_src = yring[(n7 & (kYSlots - 1)) * 64 + lane];
        else y_src = SUB == 0 ? lw.x : (SUB == 1 ? lw.y : (SUB == 2 ? lw.z : lw.w));
        pf2 uv_d;
        if constexpr (UVR) {   // (u, v)[n6 - s_p] out of the LDS ring (this lane's own slots: in order within the wave)
            uvr[(n6 & 15) * 64] = uv;
            uv_d = uvr[((n6 - sp) & 15) * 64];
        } else if constexpr (UVB) {
            // before this step's write, bank m slot j holds the value of 4 m + e steps ago, e = (SUB - j) mod 4 taken from 1 .. 4
            constexpr int e = (SP - 1) % 4 + 1;
            uv_d = uvd[4 * ((SP - 1) / 4) + ((SUB - e) & 3)];
        } else {
            uv_d = SP > 0 ? uvd[SP > 0 ? SP - 1 : 0] : uv;
            if (S::RT) {   // (u, v)[n6 - s_p] out of the window: a chain of uniform selects instead of a dynamic register index
                if (sp == 0) uv_d = uv;
#pragma unroll
                for (int j = 0; j + 1 < SP; ++j)
                    if (sp == j + 1) uv_d = uvd[j];
            }
        }
        Rgb<float> o = back.template step<EDGE>(k, kb, lk, uv_last, n6, uv, uv_d, y_src, sc);
        if constexpr (UVB) {
#pragma unroll
            for (int m = kUvBanks - 1; m > 0; --m) uvd[4 * m + SUB] = uvd[4 * (m - 1) + SUB];
            uvd[SUB] = uv;
        } else if constexpr (!UVR) {
#pragma unroll
            for (int j = SP - 1; j > 0; --j) uvd[j] = uvd[j - 1];
            if (SP > 0) uvd[0] = uv;
        }
        if (!EDGE || (n7 >= 0 && n7 < W)) put_rgb<U8, kTile>(otile, wpos, n7, o);
    };
    // the first block of the ring and the first body's carriers
    PAIR_BARRIER(d_bar);
    const lds_float *slot0 = ring + ((tb0 >> 2) & 1) * (kMid / 2) + lane * 4;
    f4 me = *(const lds_f4 *)slot0;
    f4 mo = *(const lds_f4 *)(slot0 + 256);
    f4 me2 = {0.f, 0.f, 0.f, 0.f}, mo2 = me2;      // LCUT: (me, mo) = q_e (cos, sin), (me2, mo2) = q_o
    if constexpr (LCUT) {
        me2 = *(const lds_f4 *)(slot0 + 512);
        mo2 = *(const lds_f4 *)(slot0 + 768);
    }
    f16u c4;
    if constexpr (!LCUT) c4 = load_c4(tb0);
    f8u c2 = load_c2(tb0);
    auto body_b = [&](int tb, auto edge_tag, pf2 &p_last, pf2 &uv_last) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int nxt = tb + 4;
        if (LRING) {
            lw = *(const lds_f4 *)(lr_lane + lr_r * 256);
            lr_r = lr_r + 1 == kLB ? 0 : lr_r + 1;
        } else if (!BSF) {
            lw = *(const lds_f4 *)(lring + ((tb >> 2) & 1) * (kLumaSlots / 2) + lane * 4);   // left there by A
        }
        // first half: detector products and re-modulation carriers of sub-steps 0, 1
        pf2 pe0, po0, pe1, po1;
        if constexpr (LCUT) {
            pe0 = pf2{me.x, mo.x}; po0 = pf2{me2.x, mo2.x};
            pe1 = pf2{me.y, mo.y}; po1 = pf2{me2.y, mo2.y};
        } else {
            pe0 = pk_mul_bs<0>(pf2{me.x, me.y}, pf2{c4[0], c4[1]}); po0 = pk_mul_bs<0>(pf2{mo.x, mo.y}, pf2{c4[2], c4[3]});
            pe1 = pk_mul_bs<1>(pf2{me.x, me.y}, pf2{c4[4], c4[5]}); po1 = pk_mul_bs<1>(pf2{mo.x, mo.y}, pf2{c4[6], c4[7]});
        }
        pf2 sc0 = back.remod(lk, pf2{c2[0], c2[1]}), sc1 = back.remod(lk, pf2{c2[2], c2[3]});
        sub_b(std::integral_constant<int, 0>(), edge_tag, p_last, uv_last, tb + 0, pe0, po0, sc0);
        if (s_flush == 0) maybe_flush(tb + 0);
        sub_b(std::integral_constant<int, 1>(), edge_tag, p_last, uv_last, tb + 1, pe1, po1, sc1);
        if (s_flush == 1) maybe_flush(tb + 1);
        // second half; once its products are formed this block and its carriers are dead: fetch the next ones
        pf2 pe2, po2, pe3, po3;
        if constexpr (LCUT) {
            pe2 = pf2{me.z, mo.z}; po2 = pf2{me2.z, mo2.z};
            pe3 = pf2{me.w, mo.w}; po3 = pf2{me2.w, mo2.w};
        } else {
            pe2 = pk_mul_bs<0>(pf2{me.z, me.w}, pf2{c4[8], c4[9]}); po2 = pk_mul_bs<0>(pf2{mo.z, mo.w}, pf2{c4[10], c4[11]});
            pe3 = pk_mul_bs<1>(pf2{me.z, me.w}, pf2{c4[12], c4[13]}); po3 = pk_mul_bs<1>(pf2{mo.z, mo.w}, pf2{c4[14], c4[15]});
        }
        pf2 sc2 = back.remod(lk, pf2{c2[4], c2[5]}), sc3 = back.remod(lk, pf2{c2[6], c2[7]});
        if (nxt < T_end) {
            PAIR_BARRIER(d_bar);   // block nxt / 4 of the ring is complete
            const lds_float *slot = ring + ((nxt >> 2) & 1) * (kMid / 2) + lane * 4;
            me = *(const lds_f4 *)slot;
            mo = *(const lds_f4 *)(slot + 256);
            if constexpr (LCUT) {
                me2 = *(const lds_f4 *)(slot + 512);
                mo2 = *(const lds_f4 *)(slot + 768);
            } else c4 = load_c4(nxt);
            c2 = load_c2(nxt);
        }
        sub_b(std::integral_constant<int, 2>(), edge_tag, p_last, uv_last, tb + 2, pe2, po2, sc2);
        if (s_flush == 2) maybe_flush(tb + 2);
        sub_b(std::integral_constant<int, 3>(), edge_tag, p_last, uv_last, tb + 3, pe3, po3, sc3);
        if (s_flush == 3) maybe_flush(tb + 3);
    };
    // Stage B's own schedule.  Bodies whose four detector samples all lie before the row (nd = t - front_off < 0) leave
    // every state of this stage at its reset value: only the hand-over protocol runs there.  At the other end B meets its
    // first end-of-row event at nd = W - 1 (the detector's latch), front_off steps after stage A met its first one.
#ifdef CM_EXP_NO_BSKIP   /* timing experiment: stage B on stage A's schedule */
    const int t_skip = 0, t_mid0b = t_mid0, t_mid1b = t_mid1;
#else
    const int t_skip = front_off & ~3;
    int t_mid0b = (lat_out + 3) & ~3, t_mid1b = (W - 1 + front_off) & ~3;
    if (t_mid1b <= t_mid0b) t_mid0b = t_mid1b = 0;   // tiny rows: the guarded body runs everything
#ifdef CM_EXP_ALL_EDGE
    t_mid0b = t_mid1b = 0;
#endif
#endif
    int tb = tb0;
    for (; tb < t_skip; tb += 4) {
        PAIR_BARRIER(d_bar);   // t_skip < T_end: block tb / 4 + 1 of the ring exists
        const lds_float *slot = ring + (((tb + 4) >> 2) & 1) * (kMid / 2) + lane * 4;
        me = *(const lds_f4 *)slot;
        mo = *(const lds_f4 *)(slot + 256);
        if constexpr (LCUT) {
            me2 = *(const lds_f4 *)(slot + 512);
            mo2 = *(const lds_f4 *)(slot + 768);
        } else c4 = load_c4(tb + 4);
        c2 = load_c2(tb + 4);
        if (LRING) lr_r = lr_r + 1 == kLB ? 0 : lr_r + 1;
    }
    {
        pf2 p_last = {0.f, 0.f}, uv_last = {0.f, 0.f};
        for (; tb < t_mid0b && tb < T_end; tb += 4) body_b(tb, std::true_type(), p_last, uv_last);
        for (; tb < t_mid1b && tb < T_end; tb += 4) body_b(tb, std::false_type(), p_last, uv_last);
    }
    pf2 p_last = {0.f, 0.f}, uv_last = {0.f, 0.f};
    for (; tb < T_end; tb += 4) body_b(tb, std::true_type(), p_last, uv_last);
#ifdef CM_DIAG
    if (g.diag && lane == 0 && !g.sparse) {
        unsigned long long *d = g.diag + 16ull * block + 8;
        d[0] = cm_stamp() - d_begin; d[1] = d_bar; d[2] = d_flush; d[3] = cm_realtime() - d_rbegin;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        d[4] = hw;
    }
#endif
}

template <class S>
struct PassArgs {
    Geom g;
    DemodK<float, S> k;
};
// the filter-set shape of the first-line pass of a launch (usually the main pass's; SysPalSq: another shift parity)
template <class Main, class First> struct FirstSys { typedef typename First::S type; };
template <class Main> struct FirstSys<Main, NoPass> { typedef typename Main::S type; };

// One launch runs the plain first-line pass (workgroups [0, n_first)) and the main pass.
template <class Main, class First>
__global__ __launch_bounds__(64, 2) void demod_kernel(const PassArgs<typename Main::S> main_args,
                                                      const PassArgs<typename FirstSys<Main, First>::type> first_args, const int n_first) {
    constexpr int kFloats = Main::kLdsFloats > First::kLdsFloats ? Main::kLdsFloats : First::kLdsFloats;
    __shared__ __attribute__((aligned(16))) float lds_store[kFloats];
    lds_float *lds = (lds_float *)lds_store;
    const int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
    if constexpr (!std::is_same<First, NoPass>::value) {
        if (bid < n_first) {
            run_lane<First>(first_args.g, first_args.k, bid, lds);
            return;
        }
    }
    run_lane<Main>(main_args.g, main_args.k, bid - n_first, lds);
}

// Which wave of a pair plays which stage (experiment, off by default).  Stage B is the heavier one (about 290 against 168
// vector-pipe cycles per step) and the two waves of a workgroup sit on two different SIMDs of the CU; with 10 or 12 waves on
// 4 SIMDs a SIMD that happens to host two or three B waves paces every workgroup that has a wave on it.  With
// -DCM_SIMD_BALANCE=1 the pair looks at the live load of its two SIMDs (counters per (XCC, CU, SIMD) in global memory,
// weights 4 / 7) and gives stage B to the wave on the lighter one; each wave takes its weight back when it ends.
// Measured (profiles/r01_pair_notes.md section 8): 2.7 % fewer cycles per workgroup, 2 % less clock, 0.5 % less time - the
// board is power-bound, so the product keeps the waves in launch order.
#ifndef CM_SIMD_BALANCE
#define CM_SIMD_BALANCE 0
#endif
constexpr int kSimdLoadEntries = 16 * 256 * 4;   // [XCC_ID 4 bits][SE, SH, CU ids = HW_ID bits 15:8][SIMD]
constexpr unsigned kLoadA = 4, kLoadB = 7;

#ifndef CM_PAIR_WAVES_PER_SIMD
#define CM_PAIR_WAVES_PER_SIMD Main::kPairWaves
#endif
// Wave-pair variant: 128 threads, wave 0 = stage A, wave 1 = stage B of the same 64 calls.
template <class Main, class First>
__global__ __launch_bounds__(128, CM_PAIR_WAVES_PER_SIMD) void demod_pair_kernel(const PassArgs<typename Main::S> main_args,
                                                                                 const PassArgs<typename FirstSys<Main, First>::type> first_args,
                                                                                 const int n_first) {
    extern __shared__ __attribute__((aligned(16))) float lds_store[];     // pair_lds_floats() of the larger pass
    lds_float *lds = (lds_float *)lds_store;
#ifdef CM_DEV_ROLE   /* register-pressure experiments: compile one stage only (the result does not run) */
    const int role = CM_DEV_ROLE;
#else
    int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#if defined(CM_EXP_ROLE_SWAP) && CM_EXP_ROLE_SWAP == 1   /* experiment: which wave of the pair plays which stage */
    role ^= (int)blockIdx.x & 1;
#elif defined(CM_EXP_ROLE_SWAP) && CM_EXP_ROLE_SWAP == 2
    role ^= (int)(((unsigned)blockIdx.x * 2654435761u) >> 31);
#elif defined(CM_EXP_ROLE_SWAP) && CM_EXP_ROLE_SWAP == 3
    role ^= 1;
#endif
#endif
    const int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
    if constexpr (!std::is_same<First, NoPass>::value) {
        if (bid < n_first) {
            run_pair<First>(first_args.g, first_args.k, bid, lds, role);
            return;
        }
    }
#if CM_SIMD_BALANCE
    __shared__ int bal[4];
    unsigned *cnt = main_args.g.simd_load;
    unsigned my_load = 0;
    int simd = 0;
    if (cnt) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        simd = (int)(hw >> 4) & 3;
        cnt += (((xcc & 15u) << 8) | ((hw >> 8) & 0xffu)) << 2;
        if ((threadIdx.x & 63) == 0) bal[role] = simd;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int s0 = bal[0], s1 = bal[1];
            const unsigned l0 = __hip_atomic_load(cnt + s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned l1 = __hip_atomic_load(cnt + s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned keep = (l0 + kLoadA > l1 + kLoadB ? l0 + kLoadA : l1 + kLoadB);    // wave 0 = stage A
            const unsigned swap = (l0 + kLoadB > l1 + kLoadA ? l0 + kLoadB : l1 + kLoadA);    // wave 0 = stage B
            const int sw = swap < keep || (swap == keep && (blockIdx.x & 1)) ? 1 : 0;
            bal[2] = sw;
            atomicAdd(cnt + s0, sw ? kLoadB : kLoadA);
            atomicAdd(cnt + s1, sw ? kLoadA : kLoadB);
        }
        __syncthreads();
        role ^= __builtin_amdgcn_readfirstlane(bal[2]);
        my_load = role ? kLoadB : kLoadA;
    }
#endif
    run_pair<Main>(main_args.g, main_args.k, bid - n_first, lds, role);
#if CM_SIMD_BALANCE
    if (cnt && (threadIdx.x & 63) == 0) atomicSub(cnt + simd, my_load);
#endif
}

}  // namespace cm
#endif

// cm_scan_kernels.h - the QAM-family decoders with ONE WAVEFRONT PER SCAN LINE, lanes along the row: the kernel of SMALL batches.
//
// The streaming kernels (cm_kernels.h) give a scan line to a lane and walk it sample by sample: a launch lasts as long as one
// row takes however few rows there are (0.2 ms for 720 samples; 68 us with the row cut into segments).  The reference's own
// caller lives exactly there - one frame, one row at a time - so this file restates the same chain of array operations
// (ref qam.py:34-58, pal.py:71-77, 117-125, comb.py:47-59, utils.py:28-36) with the ROW spread over the 64 lanes of a wavefront:
//
//   lane l owns the samples [l C1, (l + 1) C1) of every 1x-rate signal of the row and [2 l C1, 2 (l + 1) C1) of every 2x-rate one,
//   in registers; C1 = 12 (rows up to ~740 samples), 16 (~1000), 24 (~1500), 32 (~2030).
//
//   * resample_poly up / down by 2 (41-tap half-band FIR): every lane reads its window (+- 10 / 19 samples into the neighbours'
//     chunks) from an LDS copy of the row with zero margins and evaluates its outputs directly - 20 FMAs per output, no state;
//   * lfilter + FilterFunction (utils.py:28-36): a chunked SCAN per second-order section -
//       1. every lane runs the section over its chunk from a zero state and keeps the final state (s1, s2);
//       2. the state a lane must START from is sum over the lanes before it of A^(chunk (l - 1 - j)) (s1, s2)_j, A the section's
//          2 x 2 state transition: an exclusive Hillis-Steele scan over the lanes, log2(64) = 6 steps of ds_bpermute + a 2 x 2
//          matrix (A^(chunk 2^k), float64 on the host; steps beyond the point where the power has decayed below 1e-12 are skipped);
//       3. every lane runs the section again from that state - the same recurrence, the same operation order as the streaming
//          kernels, so the only difference to a serial walk is the rounding of the carried-in state (~1e-7 of its magnitude);
//     FilterFunction's tail (shift copies of the last sample) is part of the scanned sequence, its output offset is the index
//     the chunk is written back to LDS at;
//   * the comb filter's previous lines are the neighbouring WAVES of the workgroup (NW waves = NW - depth calls + depth halo
//     waves, base pairs exchanged through LDS once per row), the first line of a run is the plain decoder on workgroups of its own.
//
// One kernel serves every filter shape: section counts, shifts, front end, comb depth, minavg and notch are run-time
// (wave-uniform) parameters of ScanK, built per plan; the lane tables, carrier tables and the row geometry (Geom, locate_call_at)
// are the streaming kernels'.  A row costs one wavefront about 10 us; a frame's ~580 calls run side by side.
#ifndef CM_SCAN_KERNELS_H
#define CM_SCAN_KERNELS_H

#include "cm_kernels.h"
#include "cm_secam_kernels.h"

namespace cm {

constexpr int kScanSec = 4;        // CM_MAX_SECTIONS
constexpr int kScanSteps = 6;      // log2(64)
constexpr int kScanMargin = 64;    // zero floats before and after every LDS row (>= the largest FilterFunction shift, >= 20)
constexpr int kScanMaxShift = 48;

struct ScanFilter {                // one cascade in the general form 1 + b1 z^-1 + b2 z^-2 over 1 - na1 z^-1 - na2 z^-2
    int32_t nsec, shift;
    float na1[kScanSec], na2[kScanSec], b1[kScanSec], b2[kScanSec];
    float m[kScanSec][kScanSteps][4];   // m[j][k] = A_j^(chunk 2^k), row-major, A = [[na1, 1], [na2, 0]] acting on (s1, s2)
    int32_t steps[kScanSec];            // scan steps that still carry anything (<= 6)
};
struct ScanK {                     // constants of one pass (device memory, one per plan and pass)
    int32_t width, pald, bsf, depth, minavg, c1;
    float taps[10], c0;            // Taps<float>
    ScanFilter ext, rem, lpf;      // 2x rate: chunk = 2 c1
    ScanFilter pre, notch;         // 1x rate: chunk = c1
    float luma_gain, notch_gain;
    float m[9];
};
typedef const __attribute__((address_space(4))) ScanK const_ScanK;
typedef const __attribute__((address_space(4))) ScanFilter const_ScanFilter;

// value of lane - d (0 for the first d lanes)
__device__ __forceinline__ float scan_up(float v, int d, int lane) {
    const float r = lane_from(((lane - d) & 63) * 4, v);
    return lane >= d ? r : 0.f;
}

// One cascade over a chunk per lane, in place: the sequence is the concatenation of the lanes' chunks, from a zero state.
template <int CN>
__device__ __forceinline__ void scan_iir(float (&v)[CN], const_ScanFilter &f, int lane) {
    const int nsec = f.nsec;
    for (int j = 0; j < nsec; ++j) {
        const float na1 = f.na1[j], na2 = f.na2[j], b1 = f.b1[j], b2 = f.b2[j];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            const float x = v[i], y = x + s1;
            s1 = fmaf_(na1, y, fmaf_(b1, x, s2));
            s2 = fmaf_(na2, y, b2 * x);
        }
        float e1 = scan_up(s1, 1, lane), e2 = scan_up(s2, 1, lane);
        const int steps = f.steps[j];
        for (int k = 0; k < steps; ++k) {
            const float t1 = scan_up(e1, 1 << k, lane), t2 = scan_up(e2, 1 << k, lane);
            e1 = fmaf_(f.m[j][k][0], t1, fmaf_(f.m[j][k][1], t2, e1));
            e2 = fmaf_(f.m[j][k][2], t1, fmaf_(f.m[j][k][3], t2, e2));
        }
        s1 = e1;
        s2 = e2;
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            const float x = v[i], y = x + s1;
            s1 = fmaf_(na1, y, fmaf_(b1, x, s2));
            s2 = fmaf_(na2, y, b2 * x);
            v[i] = y;
        }
    }
}
// the same for two signals that share the filter (the two detector channels, (u, v))
__device__ __forceinline__ f2 fma2(float a, f2 x, f2 c) { return __builtin_elementwise_fma(f2{a, a}, x, c); }
template <int CN>
__device__ __forceinline__ void scan_iir2(f2 (&v)[CN], const_ScanFilter &f, int lane) {
    const int nsec = f.nsec;
    for (int j = 0; j < nsec; ++j) {
        const float na1 = f.na1[j], na2 = f.na2[j], b1 = f.b1[j], b2 = f.b2[j];
        f2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            const f2 x = v[i], y = x + s1;
            s1 = fma2(na1, y, fma2(b1, x, s2));
            s2 = fma2(na2, y, b2 * x);
        }
        f2 e1 = {scan_up(s1.x, 1, lane), scan_up(s1.y, 1, lane)}, e2 = {scan_up(s2.x, 1, lane), scan_up(s2.y, 1, lane)};
        const int steps = f.steps[j];
        for (int k = 0; k < steps; ++k) {
            const int d = 1 << k;
            const f2 t1 = {scan_up(e1.x, d, lane), scan_up(e1.y, d, lane)}, t2 = {scan_up(e2.x, d, lane), scan_up(e2.y, d, lane)};
            e1 = fma2(f.m[j][k][0], t1, fma2(f.m[j][k][1], t2, e1));
            e2 = fma2(f.m[j][k][2], t1, fma2(f.m[j][k][3], t2, e2));
        }
        s1 = e1;
        s2 = e2;
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            const f2 x = v[i], y = x + s1;
            s1 = fma2(na1, y, fma2(b1, x, s2));
            s2 = fma2(na2, y, b2 * x);
            v[i] = y;
        }
    }
}

struct ScanTaps {
    float c[10], c0;
    __device__ __forceinline__ float tap(int k) const { return k < 0 || k > 19 ? 0.f : c[k < 10 ? k : 19 - k]; }
};

// The 20-tap half of the half-band FIR over a chunk, two taps per instruction: out[i] = sum_k c_k w[OFF + i - k], w[j] = the
// window's sample j (aligned pairs wp[q] = (w[2 q], w[2 q + 1]) as they come out of LDS).  The products are summed in two lanes -
// a pair (w[j], w[j + 1]) with j even meets the taps (k, k - 1), k = OFF + i - j, which runs over the even k for an even OFF + i
// (11 pairs, taps -1 and 20 being zero) and over the odd k otherwise (10 pairs) - and the lanes are added at the end.
template <int C1, int OFF, int STRIDE, int FIRST, int NW2, int NOUT>    // results to out[FIRST + STRIDE i]
__device__ __forceinline__ void scan_fir20(const f2 (&wp)[NW2], const ScanTaps &tp, float (&out)[NOUT]) {
    if constexpr (C1 > 16) {     // the long chunks have no registers for the tap pairs: one tap per instruction, taps in SGPRs
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                const int j = OFF + i - k;
                acc = fmaf_(tp.tap(k), (j & 1) ? wp[j >> 1].y : wp[j >> 1].x, acc);
            }
            out[FIRST + STRIDE * i] = acc;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        const int par = (OFF + i) & 1;
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int k = par; k <= 20; k += 2) {          // j = OFF + i - k is even
            const int j = OFF + i - k;
            const f2 taps = {tp.tap(k), tp.tap(k - 1)};
            acc = __builtin_elementwise_fma(wp[j >> 1], taps, acc);
        }
        out[FIRST + STRIDE * i] = acc.x + acc.y;
    }
}
// the window w[0 .. 4 Q) = src[first .. first + 4 Q), first a multiple of 4, as aligned pairs
template <int Q>
__device__ __forceinline__ void scan_window(const lds_float *src, int first, f2 (&wp)[2 * Q]) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const f4 t = *(const lds_f4 *)(src + first + 4 * q);
        wp[2 * q] = f2{t.x, t.y};
        wp[2 * q + 1] = f2{t.z, t.w};
    }
}

// resample_poly(x, 2, 1) over this lane's chunk: even[i] = c0 x[n0 + i], odd[i] = sum_k c_k x[n0 + i + 10 - k]
// (HalfbandChain::push, cm_stages.h), interleaved into out[2 i], out[2 i + 1].  src: the 1x-rate row in LDS, zero outside [0, W).
template <int C1>
__device__ __forceinline__ void scan_up2(const lds_float *src, int n0, const ScanTaps &tp, float (&out)[2 * C1]) {
    f2 wp[(C1 + 24) / 2];     // x[n0 - 12 .. n0 + C1 + 12)
    scan_window<(C1 + 24) / 4>(src, n0 - 12, wp);
    scan_fir20<C1, 22, 2, 1>(wp, tp, out);
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        const f2 w = wp[(12 + i) >> 1];
        out[2 * i] = tp.c0 * ((i & 1) ? w.y : w.x);
    }
}
// the odd output of the same interpolator at one sample n, by every lane (FilterFunction pads with the last sample: n = W - 1)
// (one tap per instruction whatever the chunks do: a packed version of this scalar-addressed sum measured garbage on the device,
// and the last ulp of the padding value is nobody's business)
__device__ __forceinline__ float scan_up2_odd_at(const lds_float *src, int n, const ScanTaps &tp) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 20; ++k) acc = fmaf_(tp.tap(k), src[n + 10 - k], acc);
    return acc;
}
// 2 resample_poly(z, 1, 2) over this lane's chunk: out[i] = c0 z[2 n] + sum_k c_k z[2 n + 19 - 2 k] = c0 ev[n] + sum_k c_k od[n + 9 - k],
// n = n0 + i (HalfbandChain::push_pair).  The 2x-rate rows live in LDS as their even and odd samples (ev, od), zero outside [0, W).
template <int C1>
__device__ __forceinline__ void scan_dn2(const lds_float *ev, const lds_float *od, int n0, const ScanTaps &tp, float (&out)[C1]) {
    f2 wp[(C1 + 24) / 2];     // od[n0 - 12 .. n0 + C1 + 12)
    scan_window<(C1 + 24) / 4>(od, n0 - 12, wp);
    scan_fir20<C1, 21, 1, 0>(wp, tp, out);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const f4 t = *(const lds_f4 *)(ev + n0 + 4 * q);
        out[4 * q] = fmaf_(tp.c0, t.x, out[4 * q]);
        out[4 * q + 1] = fmaf_(tp.c0, t.y, out[4 * q + 1]);
        out[4 * q + 2] = fmaf_(tp.c0, t.z, out[4 * q + 2]);
        out[4 * q + 3] = fmaf_(tp.c0, t.w, out[4 * q + 3]);
    }
}

// FilterFunction (utils.py:28-36) around a cascade: v holds in[t0 .. t0 + CN) (anything beyond the row), the sequence is
// in[0 .. len) followed by copies of `last` (= in[len - 1]); the filtered chunk goes back to LDS `shift` samples earlier - the
// first `shift` outputs land in the margin before sample 0, what lies beyond the row is zeroed again (the decimators read zeros there).
template <int CN>
__device__ __forceinline__ void scan_pad(float (&v)[CN], int t0, int len, float last) {
    if (t0 + CN > len) {
#pragma unroll
        for (int i = 0; i < CN; ++i) v[i] = t0 + i >= len ? last : v[i];
    }
}
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(3))) f2u lds_f2u;
// The rows in LDS are exchanged between the lanes of ONE wavefront: it runs in lockstep and its LDS operations execute in order, so no
// hardware barrier is needed.  The compiler, however, reasons per thread: a load from an address THIS lane has not stored to may be
// hoisted above the lane's stores or merged with an earlier load of the same address (measured: niir_demod_scan_kernel, which fills
// its x row a second time - the window loads of the second pass came back with the first pass's values for the neighbouring lanes'
// samples).  Every writer of a row ends with this fence (no instruction: a compiler-level ordering point).
__device__ __forceinline__ void scan_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// a 1x-rate chunk: dst[n0 + i - shift] = v[i]
template <int CN>
__device__ __forceinline__ void scan_put(lds_float *dst, const float (&v)[CN], int n0, int shift) {
#pragma unroll
    for (int i = 0; i < CN; i += 2) *(lds_f2u *)(dst + n0 - shift + i) = f2u{v[i], v[i + 1]};
    scan_fence();
}
// a 2x-rate chunk (v[i] = sample 2 n0 + i) into the row's even / odd halves: sample m = 2 n0 + i - shift
template <int C1>
__device__ __forceinline__ void scan_put2(lds_float *ev, lds_float *od, const float (&v)[2 * C1], int n0, int shift) {
    const int h = shift >> 1;
    // even shift: v[2 j] -> ev[n0 - h + j], v[2 j + 1] -> od[n0 - h + j];  odd: v[2 j] -> od[n0 - h - 1 + j], v[2 j + 1] -> ev[n0 - h + j]
    lds_float *a = (shift & 1) ? od + (n0 - h - 1) : ev + (n0 - h);
    lds_float *b = (shift & 1) ? ev + (n0 - h) : od + (n0 - h);
#pragma unroll
    for (int j = 0; j < C1; j += 2) {
        *(lds_f2u *)(a + j) = f2u{v[2 * j], v[2 * j + 2]};
        *(lds_f2u *)(b + j) = f2u{v[2 * j + 1], v[2 * j + 3]};
    }
    scan_fence();
}
// zero what lies before sample 0 and from sample len on (1x units) of both halves
__device__ __forceinline__ void scan_trim2(lds_float *ev, lds_float *od, int len, int lane) {
    ev[lane - kScanMargin] = 0.f;
    od[lane - kScanMargin] = 0.f;
    ev[len + lane] = 0.f;
    od[len + lane] = 0.f;
    scan_fence();
}

// gf where first, else gm, field by field (scalar selects: a reference picked at run time would send both kernel arguments
// through scratch memory)
__device__ __forceinline__ Geom select_geom(const Geom &gm, const Geom &gf, bool first) {
    Geom g;
#define CM_PICK(f) g.f = first ? gf.f : gm.f
    CM_PICK(in); CM_PICK(out); CM_PICK(lanes); CM_PICK(carrier4); CM_PICK(carrier2);
    CM_PICK(in_frame_stride); CM_PICK(in_plane_stride); CM_PICK(in_row_stride);
    CM_PICK(out_frame_stride); CM_PICK(out_plane_stride); CM_PICK(out_row_stride);
    CM_PICK(total_calls); CM_PICK(first_frame); CM_PICK(cycle); CM_PICK(n_lines);
    CM_PICK(frame_rot); CM_PICK(rot_first); CM_PICK(rot_cycle);
    CM_PICK(W); CM_PICK(H); CM_PICK(Wp); CM_PICK(calls_per_frame); CM_PICK(calls_run0); CM_PICK(runs_per_frame);
    CM_PICK(first_line[0]); CM_PICK(first_line[1]); CM_PICK(k0); CM_PICK(delay); CM_PICK(rows_mode); CM_PICK(luma_prev_bits);
    CM_PICK(sparse); CM_PICK(seg_len); CM_PICK(seg_blocks); CM_PICK(seg_warm); CM_PICK(in_calls); CM_PICK(out_calls);
    CM_PICK(skip_first); CM_PICK(keep_calls); CM_PICK(diag); CM_PICK(simd_load); CM_PICK(blk_tiles);
#undef CM_PICK
    return g;
}

// ---- the ImageModem byte boundary (image.py:7-8, 20-25, 43-45, 62-71) at the edges of the scan kernels ----------------------
// U8: `row` points at bytes (Geom's strides count bytes then) - composite rows hold one byte per sample, picture rows interleaved
// R, G, B; the same conversions as the streaming kernels' (decode_bytes, read_tile3_u8, put_rgb, composite_byte).
typedef unsigned scan_u32u __attribute__((aligned(1)));
template <bool U8>
__device__ __forceinline__ const float *scan_row(const float *base, long long frame, long long frame_stride, long long row, long long row_stride) {
    if (U8) return (const float *)((const unsigned char *)base + frame * frame_stride + row * row_stride);
    return base + frame * frame_stride + row * row_stride;
}
// composite samples n .. n + 3 of a row
template <bool U8>
__device__ __forceinline__ f4 scan_load4(const float *row, int n) {
    if (U8) return decode_bytes(*(const scan_u32u *)((const unsigned char *)row + n));
    return *(const f4 *)(row + n);
}
// (r, g, b) of pixels n .. n + 3
template <bool U8>
__device__ __forceinline__ void scan_load_rgb4(const float *row, long long plane, int n, f4 out[3]) {
    if (U8) {
        const scan_u32u *p = (const scan_u32u *)((const unsigned char *)row + 3 * n);
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];      // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        const float s = 1.0f / 255.0f;
        out[0] = f4{(float)(w0 & 0xffu) * s, (float)(w0 >> 24) * s, (float)((w1 >> 16) & 0xffu) * s, (float)((w2 >> 8) & 0xffu) * s};
        out[1] = f4{(float)((w0 >> 8) & 0xffu) * s, (float)(w1 & 0xffu) * s, (float)(w1 >> 24) * s, (float)((w2 >> 16) & 0xffu) * s};
        out[2] = f4{(float)((w0 >> 16) & 0xffu) * s, (float)((w1 >> 8) & 0xffu) * s, (float)(w2 & 0xffu) * s, (float)(w2 >> 24) * s};
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) out[p] = *(const f4 *)(row + p * plane + n);
    }
}
// (r, g, b) of pixel n
template <bool U8>
__device__ __forceinline__ void scan_load_rgb1(const float *row, long long plane, int n, float &r, float &g, float &b) {
    if (U8) {
        const unsigned char *p = (const unsigned char *)row + 3 * n;
        const float s = 1.0f / 255.0f;
        r = (float)p[0] * s; g = (float)p[1] * s; b = (float)p[2] * s;
    } else {
        r = row[n]; g = row[plane + n]; b = row[2 * plane + n];
    }
}
// four composite samples out
template <bool U8>
__device__ __forceinline__ void scan_store4(float *row, int n, const f4 &o) {
    if (U8) {
        const unsigned w = (unsigned)composite_byte(o.x) | (unsigned)composite_byte(o.y) << 8 | (unsigned)composite_byte(o.z) << 16 |
                           (unsigned)composite_byte(o.w) << 24;
        *(scan_u32u *)((unsigned char *)row + n) = w;
    } else {
        *(f4 *)(row + n) = o;
    }
}
// four interleaved RGB pixels out: bytes = uint8(rint(255 * clip(x, 0, 1)))
__device__ __forceinline__ void scan_store_rgb4_u8(float *row, int n, const f4 &r, const f4 &g, const f4 &b) {
    auto pk = [](float a, float b2, float c, float d) {
        unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(255.f * a, 0, 0);
        w = __builtin_amdgcn_cvt_pk_u8_f32(255.f * b2, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(255.f * c, 2, w);
        return __builtin_amdgcn_cvt_pk_u8_f32(255.f * d, 3, w);
    };
    scan_u32u *p = (scan_u32u *)((unsigned char *)row + 3 * n);
    p[0] = pk(r.x, g.x, b.x, r.y);
    p[1] = pk(g.y, b.y, r.z, g.z);
    p[2] = pk(b.z, r.w, g.w, b.w);
}

// One launch: workgroups [0, n_first) run the plain first-line pass (NW sparse calls each), the others the main pass
// (NW - depth calls behind depth halo waves).  Dynamic LDS: NW * scan_wave_floats<C1>() floats.
template <int C1> constexpr int scan_wave_floats() { return 5 * (64 * C1 + 2 * kScanMargin); }      // x and the two halves of P and Q

// Registers: the chunks of 24 / 32 samples need more than 256 per lane; the compiler takes 256 VGPRs + 42 / 104 AGPRs and parks values in the
// AGPRs (one wave per SIMD - the LDS rows allow one workgroup per CU at these chunks anyway).  Round 3 capped the kernel at 256 registers
// (amdgpu_waves_per_eu(2, 2): 92 / 460 B of scratch per lane) because its AGPR builds had given wrong, run-to-run different results; round 4 went
// back with a register poison (tests/poison.py: NaNs in every VGPR / AGPR / LDS byte before the launch) and found the uncapped build of the
// CURRENT source exact, bit-identical under every poison and green on the whole GPU suite: the fault was the LDS ordering hazard found later in
// round 3 - lanes of a wavefront exchanging rows without a compiler-level fence, fixed by scan_fence() behind every row writer - which another
// register allocation had merely exposed.  Without the cap: 1280 x 576 84 -> 73 us, 1920 x 576 156 -> 125 us per frame (profiles/r04_scan_notes.txt).
template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void demod_scan_kernel(const Geom gm, const Geom gf, const ScanK *km, const ScanK *kf, int n_first) {
    constexpr int C2 = 2 * C1, N1 = 64 * C1, MG = kScanMargin;
    constexpr int kX = N1 + 2 * MG;       // one 1x-rate row with its margins; a 2x-rate row is two of them (even / odd samples)
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool is_first = (int)blockIdx.x < n_first;
    const Geom g = select_geom(gm, gf, is_first);
    const_ScanK &k = *(const_ScanK *)(is_first ? kf : km);
    const int depth = g.sparse ? 0 : k.depth;     // a sparse pass runs call 0 of every run: no line before it
    const long long c = g.sparse ? (long long)blockIdx.x * NW + w : (long long)((int)blockIdx.x - n_first) * (NW - depth) - depth + w;
    const LaneCall lc = locate_call_at(g, c, w >= depth);
    const bool alive = c >= 0 && c < g.total_calls;     // halo waves included: their base pairs are read by the waves behind them

    lds_float *wave = (lds_float *)scan_lds + w * scan_wave_floats<C1>();
    lds_float *X = wave + MG, *PE = X + kX, *PO = PE + kX, *QE = PO + kX, *QO = QE + kX;
    lds_float *BS = QE, *BC = QO;                         // base pairs (Ps, Pc)[n]: in the place of Q once it has been read
    const int W = g.W, L = 2 * W;
    const int n0 = lane * C1, t0 = 2 * n0;
    ScanTaps tp;
#pragma unroll
    for (int i = 0; i < 10; ++i) tp.c[i] = k.taps[i];
    tp.c0 = k.c0;
#ifdef CM_DIAG   /* cycle stamps of the stages of one wave (tools/scan_diag.py) */
    unsigned long long st[16];
    int sti = 0;
    const unsigned long long rt0 = cm_realtime();
#define CM_SCAN_STAMP() st[sti++] = cm_stamp()
#else
#define CM_SCAN_STAMP()
#endif
    CM_SCAN_STAMP();

    // ---- the row: registers + LDS copy with zero margins ---------------------------------------------------------------
    float u[C1], v[C1];          // later: the combined chroma of this lane's samples
    LaneK<float> lk;             // the call's constants: asked for here, needed behind the front end
    {
        const int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = g.lanes[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    {
        const float *xp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, g.Wp);
        X[lane - MG] = 0.f;
        X[N1 + lane] = 0.f;
        PE[lane - MG] = 0.f;
        PO[lane - MG] = 0.f;
        QE[lane - MG] = 0.f;
        QO[lane - MG] = 0.f;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            const int n = n0 + 4 * q;
            f4 t = {0.f, 0.f, 0.f, 0.f};
            if (alive && n < g.Wp) t = scan_load4<U8>(xp, n);
            if (n + 3 >= W) {
                if (n >= W) t.x = 0.f;
                if (n + 1 >= W) t.y = 0.f;
                if (n + 2 >= W) t.z = 0.f;
                if (n + 3 >= W) t.w = 0.f;
            }
            *(lds_f4 *)(X + n) = t;
        }
        scan_fence();
    }
    CM_SCAN_STAMP();   // row in LDS
    // ---- front end: the pair the product detectors multiply, then the detectors -----------------------------------------
    // kLean (chunks of 24 / 32 samples): the same operations ordered for registers - the band-stop luma before the band-pass input
    // exists, the two detector channels one after the other, carriers loaded where they are used (the packed order needs > 256 VGPRs)
    constexpr bool kLean = C1 > 16;
    f2 pq[kLean ? 1 : C2];      // (ps, pc)[t]: detector products, then their low-passed versions
    {
        const float a_last = scan_up2_odd_at(X, W - 1, tp);
        if (kLean && k.bsf) {
            float r[C2];
            scan_up2<C1>(X, n0, tp, r);
            scan_pad<C2>(r, t0, L, a_last);
            scan_iir<C2>(r, k.rem, lane);
            scan_put2<C1>(QE, QO, r, n0, k.rem.shift);
            scan_trim2(QE, QO, W, lane);
            scan_dn2<C1>(QE, QO, n0, tp, v);
        }
        float a[C2];
        scan_up2<C1>(X, n0, tp, a);
        if (!kLean && k.bsf) {  // luma = dn2(band-stop(up2 x)) * gain (qam.py:57): through Q, kept in v[] until the back end
            float r[C2];
#pragma unroll
            for (int i = 0; i < C2; ++i) r[i] = a[i];
            scan_pad<C2>(r, t0, L, a_last);
            scan_iir<C2>(r, k.rem, lane);
            scan_put2<C1>(QE, QO, r, n0, k.rem.shift);
            scan_trim2(QE, QO, W, lane);
            scan_dn2<C1>(QE, QO, n0, tp, v);
        }
        CM_SCAN_STAMP();   // up2 (+ band-stop luma)
        // the detector carriers of this lane's samples, asked for ahead of the band-pass that hides their latency
        f4 car4[kLean ? 1 : C1];
        if constexpr (!kLean) {
#pragma unroll
            for (int i = 0; i < C1; ++i) car4[i] = *(const f4 *)(g.carrier4 + 4 * (n0 + i < W ? n0 + i : W - 1));
        }
        const f4 car_last = *(const f4 *)(g.carrier4 + 4 * (W - 1));
        scan_pad<C2>(a, t0, L, a_last);
        scan_iir<C2>(a, k.ext, lane);
        CM_SCAN_STAMP();   // band-pass
        scan_put2<C1>(PE, PO, a, n0, k.ext.shift);
        scan_trim2(PE, PO, W, lane);
        float m[C2];            // the 2x-rate signal the detectors see: b (QAM front) or up2(dn2(b)) (PAL-D front)
        float m_last;
        if (k.pald) {
            float e[C1];
            scan_dn2<C1>(PE, PO, n0, tp, e);
            lds_float *E = QE;          // 1x-rate row in the place of Q's first half (free here)
            E[lane - MG] = 0.f;
            E[N1 + lane] = 0.f;
#pragma unroll
            for (int q = 0; q < C1 / 4; ++q) {
                f4 t;
                t.x = n0 + 4 * q < W ? e[4 * q] : 0.f;
                t.y = n0 + 4 * q + 1 < W ? e[4 * q + 1] : 0.f;
                t.z = n0 + 4 * q + 2 < W ? e[4 * q + 2] : 0.f;
                t.w = n0 + 4 * q + 3 < W ? e[4 * q + 3] : 0.f;
                *(lds_f4 *)(E + n0 + 4 * q) = t;
            }
            scan_fence();
            scan_up2<C1>(E, n0, tp, m);
            m_last = scan_up2_odd_at(E, W - 1, tp);
        } else {
#pragma unroll
            for (int q = 0; q < C1 / 4; ++q) {
                const f4 te = *(const lds_f4 *)(PE + n0 + 4 * q), to = *(const lds_f4 *)(PO + n0 + 4 * q);
                m[8 * q] = te.x; m[8 * q + 1] = to.x; m[8 * q + 2] = te.y; m[8 * q + 3] = to.y;
                m[8 * q + 4] = te.z; m[8 * q + 5] = to.z; m[8 * q + 6] = te.w; m[8 * q + 7] = to.w;
            }
            m_last = PO[W - 1];
        }
        CM_SCAN_STAMP();   // dn2 / up2 of e
        // product detectors against the phase-free carriers (Detector::step): car = {C[2 n], S[2 n], C[2 n + 1], S[2 n + 1]}
        if constexpr (!kLean) {
#pragma unroll
            for (int i = 0; i < C1; ++i) {
                const f4 car = car4[i];
                pq[2 * i] = f2{m[2 * i] * car.y, m[2 * i] * car.x};
                pq[2 * i + 1] = f2{m[2 * i + 1] * car.w, m[2 * i + 1] * car.z};
            }
            if (t0 + C2 > L) {
                const f2 last = {m_last * car_last.w, m_last * car_last.z};
#pragma unroll
                for (int i = 0; i < C2; ++i) pq[i] = t0 + i >= L ? last : pq[i];
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {          // 0: against sin (-> P), 1: against cos (-> Q)
                float d[C2];
#pragma unroll
                for (int i = 0; i < C1; ++i) {
                    const f4 car = *(const f4 *)(g.carrier4 + 4 * (n0 + i < W ? n0 + i : W - 1));
                    d[2 * i] = m[2 * i] * (ch ? car.x : car.y);
                    d[2 * i + 1] = m[2 * i + 1] * (ch ? car.z : car.w);
                }
                scan_pad<C2>(d, t0, L, m_last * (ch ? car_last.z : car_last.w));
                scan_iir<C2>(d, k.lpf, lane);
                scan_put2<C1>(ch ? QE : PE, ch ? QO : PO, d, n0, k.lpf.shift);
                scan_trim2(ch ? QE : PE, ch ? QO : PO, W, lane);
            }
        }
    }
    CM_SCAN_STAMP();   // detector products
    if constexpr (!kLean) {
        scan_iir2<C2>(pq, k.lpf, lane);
        float s[C2];
#pragma unroll
        for (int i = 0; i < C2; ++i) s[i] = pq[i].x;
        scan_put2<C1>(PE, PO, s, n0, k.lpf.shift);
        scan_trim2(PE, PO, W, lane);
#pragma unroll
        for (int i = 0; i < C2; ++i) s[i] = pq[i].y;
        scan_put2<C1>(QE, QO, s, n0, k.lpf.shift);
        scan_trim2(QE, QO, W, lane);
    }
    CM_SCAN_STAMP();   // low-pass + put
    float bs[C1], bc[C1];       // this line's base pair
    scan_dn2<C1>(PE, PO, n0, tp, bs);
    scan_dn2<C1>(QE, QO, n0, tp, bc);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        *(lds_f4 *)(BS + n0 + 4 * q) = f4{bs[4 * q], bs[4 * q + 1], bs[4 * q + 2], bs[4 * q + 3]};
        *(lds_f4 *)(BC + n0 + 4 * q) = f4{bc[4 * q], bc[4 * q + 1], bc[4 * q + 2], bc[4 * q + 3]};
    }
    CM_SCAN_STAMP();   // dn2 x 2, base pairs out
    __syncthreads();
    if (w < depth || !alive) return;

    // ---- comb combination (DemodBack::combine) --------------------------------------------------------------------------
    apply_frame_rotation(g, lc.frame, lk);
    CM_SCAN_STAMP();   // barrier + lane constants
    float y[C1];
    if (k.bsf) {
#pragma unroll
        for (int i = 0; i < C1; ++i) y[i] = v[i] * k.luma_gain;
    }
    const int minavg = k.minavg;
    auto combine_at = [&](const lds_float *b0s, const lds_float *b0c, int n, float &uo, float &vo) {
        // the base pairs of this call and of the depth calls before it (the waves before this one) at sample n
        const int wf = scan_wave_floats<C1>();
        float u1 = fmaf_(lk.cu[0][0], b0s[n], lk.cu[0][1] * b0c[n]);
        float v1 = fmaf_(lk.cv[0][0], b0s[n], lk.cv[0][1] * b0c[n]);
        float u2 = 0.f, v2 = 0.f;
        if (minavg) {
            u2 = fmaf_(lk.cu2[0][0], b0s[n], lk.cu2[0][1] * b0c[n]);
            v2 = fmaf_(lk.cv2[0][0], b0s[n], lk.cv2[0][1] * b0c[n]);
        }
#pragma unroll
        for (int j = 1; j < 3; ++j) {
            if (j > depth) continue;
            const float ps = b0s[n - j * wf], pc = b0c[n - j * wf];
            u1 = fmaf_(lk.cu[j][0], ps, fmaf_(lk.cu[j][1], pc, u1));
            v1 = fmaf_(lk.cv[j][0], ps, fmaf_(lk.cv[j][1], pc, v1));
            if (minavg) {
                u2 = fmaf_(lk.cu2[j][0], ps, fmaf_(lk.cu2[j][1], pc, u2));
                v2 = fmaf_(lk.cv2[j][0], ps, fmaf_(lk.cv2[j][1], pc, v2));
            }
        }
        uo = minavg ? minavg_(u1, u2) : u1;
        vo = minavg ? minavg_(v1, v2) : v1;
    };
    // (the lines beyond the plan's depth enter as zeros: no branch per sample)
    auto combine_chunk = [&](auto minavg_tag) __attribute__((always_inline)) {
        constexpr bool MA = decltype(minavg_tag)::value;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            f4 ps[3], pc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ps[j] = pc[j] = f4{0.f, 0.f, 0.f, 0.f};
                if (j <= depth) {
                    ps[j] = *(const lds_f4 *)(BS - j * scan_wave_floats<C1>() + n0 + 4 * q);
                    pc[j] = *(const lds_f4 *)(BC - j * scan_wave_floats<C1>() + n0 + 4 * q);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u1 = fmaf_(lk.cu[0][0], ps[0][e], lk.cu[0][1] * pc[0][e]);
                float v1 = fmaf_(lk.cv[0][0], ps[0][e], lk.cv[0][1] * pc[0][e]);
                float u2 = 0.f, v2 = 0.f;
                if (MA) {
                    u2 = fmaf_(lk.cu2[0][0], ps[0][e], lk.cu2[0][1] * pc[0][e]);
                    v2 = fmaf_(lk.cv2[0][0], ps[0][e], lk.cv2[0][1] * pc[0][e]);
                }
#pragma unroll
                for (int j = 1; j < 3; ++j) {
                    u1 = fmaf_(lk.cu[j][0], ps[j][e], fmaf_(lk.cu[j][1], pc[j][e], u1));
                    v1 = fmaf_(lk.cv[j][0], ps[j][e], fmaf_(lk.cv[j][1], pc[j][e], v1));
                    if (MA) {
                        u2 = fmaf_(lk.cu2[j][0], ps[j][e], fmaf_(lk.cu2[j][1], pc[j][e], u2));
                        v2 = fmaf_(lk.cv2[j][0], ps[j][e], fmaf_(lk.cv2[j][1], pc[j][e], v2));
                    }
                }
                u[4 * q + e] = MA ? minavg_(u1, u2) : u1;
                v[4 * q + e] = MA ? minavg_(v1, v2) : v1;
            }
        }
    };
    if (minavg) combine_chunk(std::true_type());
    else combine_chunk(std::false_type());
    CM_SCAN_STAMP();   // combination
    // ---- back end (DemodBack::step): pre-correction low-pass of (u, v), re-modulation, notch, matrix ----------------------
    {
        // re-modulation carriers and the luma source row, asked for ahead of the pre-correction filter
        f2 car2[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) car2[i] = *(const f2 *)(g.carrier2 + 2 * (n0 + i < W ? n0 + i : W - 1));
        if (!k.bsf) {      // luma source: this call's row or the previous call's (comb.py:102), straight from memory
            const int luma_row = ((g.luma_prev_bits >> lc.regime) & 1) ? lc.prev_row : lc.src_row;
            const float *lp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, luma_row, g.Wp);
#pragma unroll
            for (int q = 0; q < C1 / 4; ++q) {
                f4 t = {0.f, 0.f, 0.f, 0.f};
                if (n0 + 4 * q < g.Wp) t = scan_load4<U8>(lp, n0 + 4 * q);
                y[4 * q] = t.x; y[4 * q + 1] = t.y; y[4 * q + 2] = t.z; y[4 * q + 3] = t.w;
            }
        }
        f2 wuv[C1];
        float u_last, v_last;
        combine_at(BS, BC, W - 1, u_last, v_last);
#pragma unroll
        for (int i = 0; i < C1; ++i) wuv[i] = n0 + i >= W ? f2{u_last, v_last} : f2{u[i], v[i]};
        scan_iir2<C1>(wuv, k.pre, lane);
        // the filtered pair of sample n7 is output n7 + s_p of the cascade: through P (two 1x-rate rows)
        lds_float *PU = PE, *PV = PO;
        {
            float s[C1];
#pragma unroll
            for (int i = 0; i < C1; ++i) s[i] = wuv[i].x;
            scan_put<C1>(PU, s, n0, k.pre.shift);
#pragma unroll
            for (int i = 0; i < C1; ++i) s[i] = wuv[i].y;
            scan_put<C1>(PV, s, n0, k.pre.shift);
        }
        CM_SCAN_STAMP();   // pre-correction low-pass + put
        const bool strip = lk.sph != 0.f || lk.cph != 0.f;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            const f4 tu = *(const lds_f4 *)(PU + n0 + 4 * q), tv = *(const lds_f4 *)(PV + n0 + 4 * q);
            const float wu[4] = {tu.x, tu.y, tu.z, tu.w}, wv[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q + e;
                const f2 car = car2[i];
                const float sn = fmaf_(lk.sph, car.x, lk.cph * car.y);
                const float cs = fmaf_(lk.vcph, car.x, -(lk.vsph * car.y));
                y[i] = y[i] - fmaf_(sn, wu[e], cs * wv[e]);
            }
        }
        if (k.notch_gain != 0.f) {     // comb.py:54-55: luma[0 .. W) from a zero state, where the line re-modulates
            float yn[C1];
#pragma unroll
            for (int i = 0; i < C1; ++i) yn[i] = n0 + i < W ? y[i] : 0.f;
            scan_iir<C1>(yn, k.notch, lane);
            if (strip) {
#pragma unroll
                for (int i = 0; i < C1; ++i) y[i] = yn[i] * k.notch_gain;
            }
        }
    }
    CM_SCAN_STAMP();   // re-modulation, notch
    if (!lc.store_ok) return;
    if (U8) {      // interleaved bytes: the three planes of a quad of pixels together
        float *ob = (float *)scan_row<true>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride);
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            f4 o[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float m0 = k.m[3 * p], m1 = k.m[3 * p + 1], m2 = k.m[3 * p + 2];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[p][e] = fmaf_(m0, y[4 * q + e], fmaf_(m1, u[4 * q + e], m2 * v[4 * q + e]));
            }
            if (n0 + 4 * q < g.Wp) scan_store_rgb4_u8(ob, n0 + 4 * q, o[0], o[1], o[2]);
        }
        return;
    }
    float *op = g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride + n0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float *pp = op + p * g.out_plane_stride;
        const float m0 = k.m[3 * p], m1 = k.m[3 * p + 1], m2 = k.m[3 * p + 2];
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            f4 o;
            o.x = fmaf_(m0, y[4 * q], fmaf_(m1, u[4 * q], m2 * v[4 * q]));
            o.y = fmaf_(m0, y[4 * q + 1], fmaf_(m1, u[4 * q + 1], m2 * v[4 * q + 1]));
            o.z = fmaf_(m0, y[4 * q + 2], fmaf_(m1, u[4 * q + 2], m2 * v[4 * q + 2]));
            o.w = fmaf_(m0, y[4 * q + 3], fmaf_(m1, u[4 * q + 3], m2 * v[4 * q + 3]));
            if (n0 + 4 * q < g.Wp) *(f4 *)(pp + 4 * q) = o;
        }
    }
#ifdef CM_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CM_SCAN_STAMP();
    if (g.diag && lane == 0 && (int)blockIdx.x < 512) {
        unsigned long long *d = g.diag + ((long long)blockIdx.x * NW + w) * 16;
        for (int i = 0; i < sti && i < 14; ++i) d[i] = st[i];
        d[14] = cm_realtime() - rt0;
        d[15] = sti;
    }
#endif
}

// =============================================================================================================================
// The QAM modulators (ref qam.py:20-32 behind pal.py:48-52 / ntsc.py:43-45, comb.py:141-152) the same way: one wavefront per call,
// NW independent calls per workgroup.  (y, u, v) = e (r, g, b) of this lane's samples straight from memory - with the previous
// call's row mixed in for ColorAveragingModem (row weights of ModLaneK) -, the pre-correction low-pass of (u, v) as one packed
// scan, its output offset through LDS, composite[n] = y[n] + sin(phi + 2 n cps) F(u)[n] + (+-cos) F(v)[n] (QamModCore::step).
// =============================================================================================================================
struct ScanModK {
    int32_t width, depth, c1, pad;
    float e[9];
    ScanFilter pre;                // 1x rate: chunk = c1
};
typedef const __attribute__((address_space(4))) ScanModK const_ScanModK;
template <int C1> constexpr int scan_mod_wave_floats() { return 2 * (64 * C1 + 2 * kScanMargin); }

template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void qam_mod_scan_kernel(const Geom g, const ScanModK *km) {
    constexpr int N1 = 64 * C1, MG = kScanMargin;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanModK &k = *(const_ScanModK *)km;
    const long long c = (long long)blockIdx.x * NW + w;
    if (c >= g.total_calls) return;                       // (no barrier in this kernel)
    const LaneCall lc = locate_call_at(g, c, true);
    const LaneCall lp = locate_call_at(g, c > 0 ? c - 1 : 0, true);      // the previous call of the list (line averaging)
    lds_float *PU = (lds_float *)scan_lds + w * scan_mod_wave_floats<C1>() + MG, *PV = PU + N1 + 2 * MG;
    const int W = g.W, n0 = lane * C1;
    ModLaneK<float> lk;
    {
        const int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const ModLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
        float rc, rs;
        if (frame_turn(g, lc.frame, rc, rs)) {
            turn(lk.cph, lk.sph, rc, rs);
            turn(lk.vcph, lk.vsph, rc, rs);
        }
    }
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, row_stride);
    const float *rq = scan_row<U8>(g.in, lp.frame, g.in_frame_stride, lp.src_row, row_stride);
    const int depth = k.depth;
    f2 car2[C1];
#pragma unroll
    for (int i = 0; i < C1; ++i) car2[i] = *(const f2 *)(g.carrier2 + 2 * (n0 + i < W ? n0 + i : W - 1));
    // (y, u, v) of one sample of the call (QamModCore's caller, cm_mod_kernels.h: the same operation order)
    auto yuv_of = [&](float r, float gg, float b, float rr, float gr, float br, float &y, float &u, float &v) {
        y = fmaf_(k.e[0], r, fmaf_(k.e[1], gg, k.e[2] * b));
        u = fmaf_(k.e[3], r, fmaf_(k.e[4], gg, k.e[5] * b));
        v = fmaf_(k.e[6], r, fmaf_(k.e[7], gg, k.e[8] * b));
        if (depth >= 1) {
            const float yp = fmaf_(k.e[0], rr, fmaf_(k.e[1], gr, k.e[2] * br));
            const float up = fmaf_(k.e[3], rr, fmaf_(k.e[4], gr, k.e[5] * br));
            const float vp = fmaf_(k.e[6], rr, fmaf_(k.e[7], gr, k.e[8] * br));
            y = fmaf_(lk.wy0, y, lk.wy1 * yp);
            u = fmaf_(lk.wc0, u, lk.wc1 * up);
            v = fmaf_(lk.wc0, v, lk.wc1 * vp);
        }
    };
    float y[C1];
    f2 uv[C1];
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const int n = n0 + 4 * q;
        f4 a[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = b[p] = f4{0.f, 0.f, 0.f, 0.f};
        if (n < g.Wp) {
            scan_load_rgb4<U8>(rp, g.in_plane_stride, n, a);
            if (depth >= 1) scan_load_rgb4<U8>(rq, g.in_plane_stride, n, b);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float yy, uu, vv;
            yuv_of(a[0][e], a[1][e], a[2][e], b[0][e], b[1][e], b[2][e], yy, uu, vv);
            y[4 * q + e] = yy;
            uv[4 * q + e] = f2{uu, vv};
        }
    }
    {   // FilterFunction pads with the last sample (utils.py:31-33): (u, v)[W - 1], by every lane
        float yl, ul, vl;
        float ar, ag, ab, br = 0.f, bg = 0.f, bb = 0.f;
        scan_load_rgb1<U8>(rp, g.in_plane_stride, W - 1, ar, ag, ab);
        if (depth >= 1) scan_load_rgb1<U8>(rq, g.in_plane_stride, W - 1, br, bg, bb);
        yuv_of(ar, ag, ab, br, bg, bb, yl, ul, vl);
        if (n0 + C1 > W) {
#pragma unroll
            for (int i = 0; i < C1; ++i) uv[i] = n0 + i >= W ? f2{ul, vl} : uv[i];
        }
    }
    scan_iir2<C1>(uv, k.pre, lane);
    {
        float s[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) s[i] = uv[i].x;
        scan_put<C1>(PU, s, n0, k.pre.shift);
#pragma unroll
        for (int i = 0; i < C1; ++i) s[i] = uv[i].y;
        scan_put<C1>(PV, s, n0, k.pre.shift);
    }
    if (!lc.store_ok) return;
    float *op = (float *)scan_row<U8>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        if (n0 + 4 * q >= g.Wp) continue;
        const f4 tu = *(const lds_f4 *)(PU + n0 + 4 * q), tv = *(const lds_f4 *)(PV + n0 + 4 * q);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f2 car = car2[4 * q + e];
            const float sn = fmaf_(lk.sph, car.x, lk.cph * car.y);
            const float cs = fmaf_(lk.vcph, car.x, -(lk.vsph * car.y));
            o[e] = y[4 * q + e] + fmaf_(sn, tu[e], cs * tv[e]);
        }
        scan_store4<U8>(op, n0 + 4 * q, o);
    }
}

// =============================================================================================================================
// The back end of the wrapped PAL combs (SimpleCombModem / Simple3DCombModem around PalDModem / Pal3DModem, ref comb.py:96-113;
// comb_wrap_back_kernel, cm_wrap_kernels.h) with one wavefront per call: the components of this call and of the call before it
// straight from the scratch of the inner decoder, (u, v) = avg / minavg, the backend modulator's pre-correction low-pass as one
// packed scan (its constants: the backend plan's ScanModK), re-modulation at line - 2 own_delay, notch, matrix.
// =============================================================================================================================
struct ScanWrapArgs {              // by value: a few scalars + the wrapper's notch (one section, shift 0) in the scan form
    int32_t own_delay, minavg, strip, notch_steps;
    float na1, na2, b1, b2, notch_gain;      // notch_gain 0: no notch
    float nm[kScanSteps][4];
    float m[9];
};

template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void wrap_back_scan_kernel(const Geom g, const ScanModK *km, const ScanWrapArgs a) {
    constexpr int N1 = 64 * C1, MG = kScanMargin;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanModK &k = *(const_ScanModK *)km;
    const long long c = (long long)blockIdx.x * NW + w;
    if (c >= g.total_calls) return;                       // (no barrier in this kernel)
    const LaneCall lc = locate_call_at(g, c, true);
    lds_float *PU = (lds_float *)scan_lds + w * scan_mod_wave_floats<C1>() + MG, *PV = PU + N1 + 2 * MG;
    const int W = g.W, n0 = lane * C1;
    const bool first = lc.kk == 0;                        // call 0 of a run: (y, u, v) = curr, not stripped (comb.py:97-99)
    const bool strip = a.strip != 0 && !first;
    const float strip_f = strip ? 1.f : 0.f;
    const bool take_prev_y = a.own_delay != 0 && !first;
    ModLaneK<float> lk;
    {
        int lm = lc.line - 2 * a.own_delay;               // the line the wrapper re-modulates at; unused where first
        if (lm < 0) lm &= 1;
        const int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const ModLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lm];
        float rc, rs;
        if (frame_turn(g, lc.frame, rc, rs)) {
            turn(lk.cph, lk.sph, rc, rs);
            turn(lk.vcph, lk.vsph, rc, rs);
        }
    }
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = g.in + lc.frame * g.in_frame_stride + (long long)lc.src_row * row_stride;
    const float *rq = g.in + lc.frame * g.in_frame_stride + (long long)lc.prev_row * row_stride;
    f2 car2[C1];
#pragma unroll
    for (int i = 0; i < C1; ++i) car2[i] = *(const f2 *)(g.carrier2 + 2 * (n0 + i < W ? n0 + i : W - 1));
    auto mix = [&](float cy, float cu, float cv, float py, float pu, float pv, float &ys, float &u, float &v) {
        u = a.minavg == 1 ? minavg_(pu, cu) : 0.5f * (pu + cu);                // comb.py:102-104
        v = a.minavg == 1 ? minavg_(pv, cv) : 0.5f * (pv + cv);
        if (first || a.minavg == 2) { u = cu; v = cv; }          // (2: the caller has averaged the component buffer itself - avg= callables)
        ys = take_prev_y ? py : cy;
    };
    float y[C1], ud[C1], vd[C1];
    f2 uv[C1];
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const int n = n0 + 4 * q;
        f4 cur[3], prv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) cur[p] = prv[p] = f4{0.f, 0.f, 0.f, 0.f};
        if (n < g.Wp) {
            scan_load_rgb4<false>(rp, g.in_plane_stride, n, cur);
            scan_load_rgb4<false>(rq, g.in_plane_stride, n, prv);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ys, u, v;
            mix(cur[0][e], cur[1][e], cur[2][e], prv[0][e], prv[1][e], prv[2][e], ys, u, v);
            y[4 * q + e] = ys;
            ud[4 * q + e] = u;
            vd[4 * q + e] = v;
            uv[4 * q + e] = f2{u, v};
        }
    }
    {   // FilterFunction pads with the last sample (utils.py:31-33): (u, v)[W - 1], by every lane
        float cy, cu, cv, py, pu, pv, ys, ul, vl;
        scan_load_rgb1<false>(rp, g.in_plane_stride, W - 1, cy, cu, cv);
        scan_load_rgb1<false>(rq, g.in_plane_stride, W - 1, py, pu, pv);
        mix(cy, cu, cv, py, pu, pv, ys, ul, vl);
        if (n0 + C1 > W) {
#pragma unroll
            for (int i = 0; i < C1; ++i) uv[i] = n0 + i >= W ? f2{ul, vl} : uv[i];
        }
    }
    scan_iir2<C1>(uv, k.pre, lane);
    {
        float s[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) s[i] = uv[i].x;
        scan_put<C1>(PU, s, n0, k.pre.shift);
#pragma unroll
        for (int i = 0; i < C1; ++i) s[i] = uv[i].y;
        scan_put<C1>(PV, s, n0, k.pre.shift);
    }
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const f4 tu = *(const lds_f4 *)(PU + n0 + 4 * q), tv = *(const lds_f4 *)(PV + n0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            const f2 car = car2[i];
            const float sn = fmaf_(lk.sph, car.x, lk.cph * car.y);
            const float cs = fmaf_(lk.vcph, car.x, -(lk.vsph * car.y));
            y[i] = fmaf_(-strip_f, fmaf_(sn, tu[e], cs * tv[e]), y[i]);       // comb.py:105-107
        }
    }
    if (a.notch_gain != 0.f) {     // comb.py:108-110: luma[0 .. W) from a zero state (one section, shift 0), where the wrapper strips
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const float x = n0 + i < W ? y[i] : 0.f, yy = x + s1;
            s1 = fmaf_(a.na1, yy, fmaf_(a.b1, x, s2));
            s2 = fmaf_(a.na2, yy, a.b2 * x);
        }
        float e1 = scan_up(s1, 1, lane), e2 = scan_up(s2, 1, lane);
#pragma unroll
        for (int kk = 0; kk < kScanSteps; ++kk) {
            if (kk < a.notch_steps) {
                const float t1 = scan_up(e1, 1 << kk, lane), t2 = scan_up(e2, 1 << kk, lane);
                e1 = fmaf_(a.nm[kk][0], t1, fmaf_(a.nm[kk][1], t2, e1));
                e2 = fmaf_(a.nm[kk][2], t1, fmaf_(a.nm[kk][3], t2, e2));
            }
        }
        s1 = e1;
        s2 = e2;
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const float x = n0 + i < W ? y[i] : 0.f, yy = x + s1;
            s1 = fmaf_(a.na1, yy, fmaf_(a.b1, x, s2));
            s2 = fmaf_(a.na2, yy, a.b2 * x);
            if (strip) y[i] = yy * a.notch_gain;
        }
    }
    if (!lc.store_ok) return;
    float *op = (float *)scan_row<U8>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride);
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        f4 o[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[p][e] = fmaf_(a.m[3 * p], y[4 * q + e], fmaf_(a.m[3 * p + 1], ud[4 * q + e], a.m[3 * p + 2] * vd[4 * q + e]));
        }
        if (n0 + 4 * q >= g.Wp) continue;
        if (U8) scan_store_rgb4_u8(op, n0 + 4 * q, o[0], o[1], o[2]);
        else {
#pragma unroll
            for (int p = 0; p < 3; ++p) *(f4 *)(op + p * g.out_plane_stride + n0 + 4 * q) = o[p];
        }
    }
}

// =============================================================================================================================
// The SECAM modulator (ref secam.py:240-276; SecamMod::step, cm_stages.h) with one wavefront per call.  Its colour-difference
// path runs in float64 on the device (the phase is an integral over the whole line), so the scans here carry doubles:
//   d -> pre-correction low-pass (two sections, FilterFunction shift s_p) -> LF pre-emphasis (one section, from sample 0) ->
//   f = fsc + fdev x, clipped -> bell pre-emphasis G(f) -> phase[n] = start - arg G[0] + pi sum_{i=1..n} f[i] (numpy.cumsum: a
//   prefix sum over the lanes) -> composite[n] = luma[n] + Re(G e^{j phase}).
// =============================================================================================================================
struct ScanFilterD {               // ScanFilter in float64
    int32_t nsec, shift;
    double na1[kScanSec], na2[kScanSec], b1[kScanSec], b2[kScanSec];
    double m[kScanSec][kScanSteps][4];
    int32_t steps[kScanSec];
};
struct ScanSecamModK {
    int32_t width, depth, c1, pad;
    ScanFilterD pre_lp, lf_pre;    // 1x rate: chunk = c1
    double gain, f_min, f_max, f0, pi, two_pi;
    float m0, kn, kd, pad2;
    float e[9];
};
typedef const __attribute__((address_space(4))) ScanSecamModK const_ScanSecamModK;
typedef const __attribute__((address_space(4))) ScanFilterD const_ScanFilterD;

__device__ __forceinline__ double scan_up_d(double v, int d, int lane) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int idx = ((lane - d) & 63) * 4;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)(unsigned)(b >> 32));
    const double r = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    return lane >= d ? r : 0.0;
}
template <int CN>
__device__ __forceinline__ void scan_iir_d(double (&v)[CN], const_ScanFilterD &f, int lane) {
    const int nsec = f.nsec;
    for (int j = 0; j < nsec; ++j) {
        const double na1 = f.na1[j], na2 = f.na2[j], b1 = f.b1[j], b2 = f.b2[j];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < CN; ++i) {           // iir_gen (cm_stages.h) in float64
            const double x = v[i], y = x + s1;
            s1 = fmaf_(na1, y, fmaf_(b1, x, s2));
            s2 = fmaf_(na2, y, b2 * x);
        }
        double e1 = scan_up_d(s1, 1, lane), e2 = scan_up_d(s2, 1, lane);
        const int steps = f.steps[j];
        for (int k = 0; k < steps; ++k) {
            const double t1 = scan_up_d(e1, 1 << k, lane), t2 = scan_up_d(e2, 1 << k, lane);
            e1 = fmaf_(f.m[j][k][0], t1, fmaf_(f.m[j][k][1], t2, e1));
            e2 = fmaf_(f.m[j][k][2], t1, fmaf_(f.m[j][k][3], t2, e2));
        }
        s1 = e1;
        s2 = e2;
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            const double x = v[i], y = x + s1;
            s1 = fmaf_(na1, y, fmaf_(b1, x, s2));
            s2 = fmaf_(na2, y, b2 * x);
            v[i] = y;
        }
    }
}
template <int C1> constexpr int scan_secam_mod_wave_floats() { return 2 * (64 * C1 + 2 * kScanMargin); }    // one row of doubles

template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void secam_mod_scan_kernel(const Geom g, const ScanSecamModK *km) {
    constexpr int N1 = 64 * C1, MG = kScanMargin;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    typedef __attribute__((address_space(3))) double lds_double;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanSecamModK &k = *(const_ScanSecamModK *)km;
    const long long c = (long long)blockIdx.x * NW + w;
    if (c >= g.total_calls) return;                       // (no barrier in this kernel)
    const LaneCall lc = locate_call_at(g, c, true);
    const LaneCall lp = locate_call_at(g, c > 0 ? c - 1 : 0, true);
    lds_double *PW = (lds_double *)((lds_float *)scan_lds + w * scan_secam_mod_wave_floats<C1>()) + MG;
    const int W = g.W, n0 = lane * C1;
    SecamModLaneK<float, double> lk;
    {
        const int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamModLaneK<float, double> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    const long long row_stride = g.in_row_stride ? g.in_row_stride : g.W;
    const float *rp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, row_stride);
    const float *rq = scan_row<U8>(g.in, lp.frame, g.in_frame_stride, lp.src_row, row_stride);
    const int depth = k.depth;
    const bool own_db = lk.own_is_db != 0.f;
    // (luma, d) of one sample (secam_mod_kernel's body, cm_secam_kernels.h: the same operation order)
    auto yd_of = [&](float r, float gg, float b, float rr, float gr, float br, float &y, float &d) {
        y = fmaf_(k.e[0], r, fmaf_(k.e[1], gg, k.e[2] * b));
        const float dr = fmaf_(k.e[3], r, fmaf_(k.e[4], gg, k.e[5] * b));
        const float db = fmaf_(k.e[6], r, fmaf_(k.e[7], gg, k.e[8] * b));
        d = own_db ? db : dr;
        if (depth >= 1) {
            const float yp = fmaf_(k.e[0], rr, fmaf_(k.e[1], gr, k.e[2] * br));
            const float drp = fmaf_(k.e[3], rr, fmaf_(k.e[4], gr, k.e[5] * br));
            const float dbp = fmaf_(k.e[6], rr, fmaf_(k.e[7], gr, k.e[8] * br));
            y = fmaf_(lk.wy0, y, lk.wy1 * yp);
            d = fmaf_(lk.wc0, d, lk.wc1 * (own_db ? dbp : drp));
        }
    };
    float y[C1];
    double dd[C1];
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const int n = n0 + 4 * q;
        f4 a[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = b[p] = f4{0.f, 0.f, 0.f, 0.f};
        if (n < g.Wp) {
            scan_load_rgb4<U8>(rp, g.in_plane_stride, n, a);
            if (depth >= 1) scan_load_rgb4<U8>(rq, g.in_plane_stride, n, b);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float yy, d;
            yd_of(a[0][e], a[1][e], a[2][e], b[0][e], b[1][e], b[2][e], yy, d);
            y[4 * q + e] = yy;
            dd[4 * q + e] = (double)d;
        }
    }
    {   // FilterFunction pads with the last sample (utils.py:31-33): d[W - 1], by every lane
        float yl, dl;
        float ar, ag, ab, br = 0.f, bg = 0.f, bb = 0.f;
        scan_load_rgb1<U8>(rp, g.in_plane_stride, W - 1, ar, ag, ab);
        if (depth >= 1) scan_load_rgb1<U8>(rq, g.in_plane_stride, W - 1, br, bg, bb);
        yd_of(ar, ag, ab, br, bg, bb, yl, dl);
        if (n0 + C1 > W) {
#pragma unroll
            for (int i = 0; i < C1; ++i) dd[i] = n0 + i >= W ? (double)dl : dd[i];
        }
    }
    scan_iir_d<C1>(dd, k.pre_lp, lane);
    // the filtered sample of n7 is output n7 + s_p of the cascade: through LDS
    PW[lane - MG] = 0.0;
#pragma unroll
    for (int i = 0; i < C1; ++i) PW[n0 - k.pre_lp.shift + i] = dd[i];
    scan_fence();
    double x[C1];
#pragma unroll
    for (int i = 0; i < C1; ++i) x[i] = PW[n0 + i];
    scan_iir_d<C1>(x, k.lf_pre, lane);                    // from sample 0 of the row, no shift (secam.py:175-177)
    // frequency, bell pre-emphasis, phase increments
    float re[C1], im[C1];
    double ph[C1];
    const double fg = lk.fdev * k.gain;
    const float f0 = (float)k.f0;
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        double f = fmaf_(fg, x[i], lk.fsc);                                   // secam.py:266 / 271
        f = f < k.f_min ? k.f_min : (f > k.f_max ? k.f_max : f);               // secam.py:272
        secam_bell_gain<float>((float)(f - k.f0), (float)f, f0, k.m0, k.kn, k.kd, re[i], im[i]);      // (cm_stages.h: the streaming encoder's)
        ph[i] = k.pi * f;
        if (n0 + i == 0) ph[i] = (double)lk.start_phase - (double)atan2f(im[i], re[i]);   // secam.py:244
    }
    // numpy.cumsum: prefix sum within the chunk, exclusive sum of the chunk totals over the lanes before
    {
        double run = 0.0;
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            run += ph[i];
            ph[i] = run;
        }
        double tot = scan_up_d(run, 1, lane);
#pragma unroll
        for (int kk = 0; kk < kScanSteps; ++kk) tot += scan_up_d(tot, 1 << kk, lane);
#pragma unroll
        for (int i = 0; i < C1; ++i) ph[i] += tot;
    }
    if (!lc.store_ok) return;
    float *op = (float *)scan_row<U8>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride);
    const double inv_two_pi = 1.0 / k.two_pi;
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            const double acc = ph[i] - k.two_pi * __builtin_floor(ph[i] * inv_two_pi);   // the running sum wrapped to [0, 2 pi)
            float sn, cs;
            sincos_((float)acc, sn, cs);                                                  // (cm_stages.h: the streaming encoder's)
            o[e] = y[i] + (re[i] * cs - im[i] * sn);                                      // secam.py:246, 276
        }
        if (n0 + 4 * q < g.Wp) scan_store4<U8>(op, n0 + 4 * q, o);
    }
}

// =============================================================================================================================
// The SECAM decoder (ref secam.py:278-304 with the FmDecoder of secam.py:127-149; SecamDemod, cm_stages.h) with one wavefront
// per call, the plans whose float32 discriminator has its margin (rows below 1280 samples).  Array operations of a row:
//   cc[m] (the row behind its mirrored pre-roll, Lc = W + P samples) -> band-pass (FilterFunction shift s_b) -> bell, both in
//   FLOAT64 (the streaming kernels carry them in float64 at the row ends, where float32 loses the sub-carrier's phase: here the
//   whole row) -> up2 -> x FM reference -> low-pass of (I, Q) (packed scan, FilterFunction shift) -> phase step of consecutive
//   samples (atan2 of cross and dot product) -> dn2 -> + dc table, clip, scale -> de-emphasis (scan, from sample 0 of the row) ;
//   luma = band-stop of the row (scan, FilterFunction shift s_y) ; (dr, db) = this call's and the previous call's c.
// NW waves = NW - 1 calls behind one halo wave.
// =============================================================================================================================
struct ScanSecamK {
    int32_t width, preroll, c1, has_bell;
    float taps[10], c0, two_over_pi;
    ScanFilterD bpf, bell;         // chunk = c1 (the chroma stream m)
    ScanFilter lpf;                // 2x rate: chunk = 2 c1
    ScanFilter ybs, deemph;        // 1x rate: chunk = c1
    float luma_gain, pad;
    float m[9];
};
typedef const __attribute__((address_space(4))) ScanSecamK const_ScanSecamK;
template <int C1> constexpr int scan_secam_wave_floats() { return 7 * (64 * C1 + 2 * kScanMargin); }

template <int C1, int NW, bool U8 = false>
__global__ __launch_bounds__(64 * NW) void secam_demod_scan_kernel(const Geom g, const ScanSecamK *km) {
    constexpr int C2 = 2 * C1, N1 = 64 * C1, MG = kScanMargin, kRow = N1 + 2 * MG;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    typedef __attribute__((address_space(3))) double lds_double;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const_ScanSecamK &k = *(const_ScanSecamK *)km;
    const long long c = (long long)blockIdx.x * (NW - 1) - 1 + w;
    const LaneCall lc = locate_call_at(g, c, w >= 1);
    const bool alive = c >= 0 && c < g.total_calls;
    lds_float *wave = (lds_float *)scan_lds + w * scan_secam_wave_floats<C1>();
    lds_float *X = wave + MG, *PE = X + kRow, *PO = PE + kRow, *QE = PO + kRow, *QO = QE + kRow, *COWN = QO + kRow, *YROW = COWN + kRow;
    lds_double *BD = (lds_double *)(PE - MG) + MG;       // the band-passed stream in float64: in the place of PE and PO
    lds_float *CH = QE;                                    // ch[m1], float: in the place of QE
    const int W = g.W, P = k.preroll, Lc = W + P;
    const int n0 = lane * C1;
    ScanTaps tp;
#pragma unroll
    for (int i = 0; i < 10; ++i) tp.c[i] = k.taps[i];
    tp.c0 = k.c0;
    SecamDemodLaneK<float> lk;
    {
        const int fmod = (int)((g.first_frame + lc.frame) % g.cycle);
        lk = ((const SecamDemodLaneK<float> *)g.lanes)[((long long)fmod * 3 + lc.regime) * g.n_lines + lc.line];
    }
    // ---- the row ------------------------------------------------------------------------------------------------------------
    float xr[C1];
    {
        const float *xp = scan_row<U8>(g.in, lc.frame, g.in_frame_stride, lc.src_row, g.Wp);
        X[lane - MG] = 0.f;
        X[N1 + lane] = 0.f;
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            const int n = n0 + 4 * q;
            f4 t = {0.f, 0.f, 0.f, 0.f};
            if (alive && n < g.Wp) t = scan_load4<U8>(xp, n);
            if (n + 3 >= W) {
                if (n >= W) t.x = 0.f;
                if (n + 1 >= W) t.y = 0.f;
                if (n + 2 >= W) t.z = 0.f;
                if (n + 3 >= W) t.w = 0.f;
            }
            *(lds_f4 *)(X + n) = t;
            xr[4 * q] = t.x; xr[4 * q + 1] = t.y; xr[4 * q + 2] = t.z; xr[4 * q + 3] = t.w;
        }
        scan_fence();
    }
    const float x_last = X[W - 1];
    // ---- luma: band-stop of the row with FilterFunction's tail, output s_y samples earlier (SecamDemod::luma_step) -----------
    {
        scan_pad<C1>(xr, n0, W, x_last);
        scan_iir<C1>(xr, k.ybs, lane);
        YROW[lane - MG] = 0.f;
        scan_put<C1>(YROW, xr, n0, k.ybs.shift);
    }
    // ---- chroma stream cc[m]: band-pass + bell in float64 --------------------------------------------------------------------
    float ch[C1];
    {
        double b[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const int m = n0 + i;
            const int idx = m < P ? P - m : (m < Lc ? m - P : W - 1);      // secam.py:283-284: the mirrored start; the tail repeats cc[Lc - 1]
            b[i] = (double)X[idx];
        }
        scan_iir_d<C1>(b, k.bpf, lane);
        BD[lane - MG] = 0.0;
#pragma unroll
        for (int i = 0; i < C1; ++i) BD[n0 - k.bpf.shift + i] = b[i];
        scan_fence();
#pragma unroll
        for (int i = 0; i < C1; ++i) b[i] = BD[n0 + i];
        if (k.has_bell) scan_iir_d<C1>(b, k.bell, lane);   // the bell sees the band-pass output from its sample 0 on
#pragma unroll
        for (int i = 0; i < C1; ++i) ch[i] = n0 + i < Lc ? (float)b[i] : 0.f;
    }
    CH[lane - MG] = 0.f;
    CH[N1 + lane] = 0.f;
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) *(lds_f4 *)(CH + n0 + 4 * q) = f4{ch[4 * q], ch[4 * q + 1], ch[4 * q + 2], ch[4 * q + 3]};
    scan_fence();
    // ---- up2, products with the FM reference, low-pass of (I, Q) ---------------------------------------------------------------
    f2 iq[C2];
    {
        float a[C2];
        scan_up2<C1>(CH, n0, tp, a);
        const float a_last = scan_up2_odd_at(CH, Lc - 1, tp);
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const f4 car = *(const f4 *)(g.carrier4 + 4 * (n0 + i < Lc ? n0 + i : Lc - 1));
            iq[2 * i] = f2{a[2 * i] * car.x, -(a[2 * i] * car.y)};             // data_up = cos part - j sin part (secam.py:143)
            iq[2 * i + 1] = f2{a[2 * i + 1] * car.z, -(a[2 * i + 1] * car.w)};
        }
        if (2 * n0 + C2 > 2 * Lc) {
            const f4 car = *(const f4 *)(g.carrier4 + 4 * (Lc - 1));
            const f2 last = {a_last * car.z, -(a_last * car.w)};
#pragma unroll
            for (int i = 0; i < C2; ++i) iq[i] = 2 * n0 + i >= 2 * Lc ? last : iq[i];
        }
    }
    scan_iir2<C2>(iq, k.lpf, lane);
    {
        float s[C2];
#pragma unroll
        for (int i = 0; i < C2; ++i) s[i] = iq[i].x;
        PE[lane - MG] = 0.f;
        PO[lane - MG] = 0.f;
        scan_put2<C1>(PE, PO, s, n0, k.lpf.shift);
#pragma unroll
        for (int i = 0; i < C2; ++i) s[i] = iq[i].y;
        QO[lane - MG] = 0.f;
        scan_put2<C1>(QE, QO, s, n0, k.lpf.shift);
    }
    // ---- phase steps of consecutive (I, Q) samples (SecamDemod::phase_step), as the deviation from the discriminator centre -----
    float f[C2];
    {
        f2 prev = {PO[n0 - 1], QO[n0 - 1]};              // sample 2 n0 - 1
#pragma unroll
        for (int q = 0; q < C1 / 4; ++q) {
            const f4 ie = *(const lds_f4 *)(PE + n0 + 4 * q), io = *(const lds_f4 *)(PO + n0 + 4 * q);
            const f4 qe = *(const lds_f4 *)(QE + n0 + 4 * q), qo = *(const lds_f4 *)(QO + n0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f2 s0 = {ie[e], qe[e]}, s1 = {io[e], qo[e]};
                const int m3 = n0 + 4 * q + e;
                float d_e = SecamDemod<float>::phase_step(prev.x, prev.y, s0.x, s0.y);
                float d_o = SecamDemod<float>::phase_step(s0.x, s0.y, s1.x, s1.y);
                if (m3 == 0) d_e = 0.f;                  // secam.py:147: the first step is 0
                if (m3 >= Lc) d_e = d_o = 0.f;
                f[8 * q + 2 * e] = d_e * k.two_over_pi;
                f[8 * q + 2 * e + 1] = d_o * k.two_over_pi;
                prev = s1;
            }
        }
    }
    scan_put2<C1>(PE, PO, f, n0, 0);
    scan_trim2(PE, PO, Lc, lane);
    float g2[C1];
    scan_dn2<C1>(PE, PO, n0, tp, g2);
    // ---- frequency -> colour difference, de-emphasis from sample 0 of the row (n = m4 - P) ----------------------------------------
    {
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const int m4 = n0 + i, n = m4 - P;
            const float dc = g.carrier2[m4 < Lc ? m4 : Lc - 1];
            float v2 = (g2[i] + dc) + lk.off2;                                   // 2 (f - fsc)
            v2 = v2 < lk.lo ? lk.lo : (v2 > lk.hi ? lk.hi : v2);                 // secam.py:290
            g2[i] = n >= 0 && n < W ? v2 * lk.scale : 0.f;
        }
        scan_iir<C1>(g2, k.deemph, lane);                                       // secam.py:291-296
        COWN[lane - MG] = 0.f;
        scan_put<C1>(COWN, g2, n0, P);
    }
    __syncthreads();
    if (w < 1 || !alive || !lc.store_ok) return;
    // ---- finish (SecamDemod::finish): this call's and the previous call's colour difference, matrix ---------------------------------
    const lds_float *CPREV = COWN - scan_secam_wave_floats<C1>();
    float *op = U8 ? (float *)scan_row<true>(g.out, lc.frame, g.out_frame_stride, lc.out_row, g.out_row_stride)
                   : g.out + lc.frame * g.out_frame_stride + (long long)lc.out_row * g.out_row_stride + n0;
    const bool own_db = lk.own_is_db != 0.f;
#pragma unroll
    for (int q = 0; q < C1 / 4; ++q) {
        const f4 yv = *(const lds_f4 *)(YROW + n0 + 4 * q), own = *(const lds_f4 *)(COWN + n0 + 4 * q), pv = *(const lds_f4 *)(CPREV + n0 + 4 * q);
        f4 o[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float luma = yv[e] * k.luma_gain, prev = pv[e] * lk.w_prev;
            const float dr = own_db ? prev : own[e], db = own_db ? own[e] : prev;   // secam.py:297-300
#pragma unroll
            for (int p = 0; p < 3; ++p) o[p][e] = fmaf_(k.m[3 * p], luma, fmaf_(k.m[3 * p + 1], dr, k.m[3 * p + 2] * db));
        }
        if (n0 + 4 * q < g.Wp) {
            if (U8) scan_store_rgb4_u8(op, n0 + 4 * q, o[0], o[1], o[2]);
            else {
#pragma unroll
                for (int p = 0; p < 3; ++p) *(f4 *)(op + p * g.out_plane_stride + 4 * q) = o[p];
            }
        }
    }
}

}  // namespace cm
#endif

// cm_mac_kernels.h - D2-MAC style time-multiplex modem (ref color_modem/color/mac.py) on gfx950.
//
// Unlike the QAM / SECAM paths this one has no recursive filter: a line is two half-band FIR passes (the 41-tap Kaiser
// filter of scipy's resample_poly) and a fixed re-arrangement of samples, so nothing runs
// sequentially along a row and the layout is the opposite of the IIR kernels': ONE WORKGROUP WALKS A SEGMENT OF
// CONSECUTIVE ROWS OF ONE FIELD, its 256 threads spread along the row (every global access is a run of consecutive
// floats), the half-rate chroma of the row staged in LDS between the two passes.  Bandwidth-bound: 4 * (3 * 720 + 1080)
// = 12960 algorithmic bytes per row either way, about 25 FMAs per sample.
//
//   line (1080 samples, mac.py:56-69): [15..17] chroma ramp-in, [18..368] chroma[5..355], [369..371] cross-fade,
//   [372..1070] luma[11..709], [1071..1073] luma ramp-out, 0.5 elsewhere;  chroma = dn2(dr or db) + 0.5 (360 samples)
//
// A "call" is one Modem.modulate() / demodulate() of the reference; calls of a field form a run (image.py:47-55, 75-83).
//   decoder  (mac.py:84-125)  call k uses the line of call k and, for the other colour-difference signal, the
//            interpolated chroma of call k - 1 (zeros on the first call of a run)
//   encoder  plain: call k uses its own row;  inside ColorAveragingModem (comb.py:141-152, modulation_delay 1): luma of
//            call k - 1, chroma = mean of calls k - 1 and k, colour-difference signal chosen for line - 2
#ifndef CM_MAC_KERNELS_H
#define CM_MAC_KERNELS_H

#include <hip/hip_runtime.h>

namespace cm {

constexpr int kMacLuma = 720, kMacChroma = 360, kMacLine = 1080;
constexpr int kMacThreads = 256;
constexpr int kMacSegment = 8;        // rows of one field per workgroup

struct MacArgs {
    const float *in;
    float *out;
    int n_frames, H;            // frames mode: H rows per frame; rows mode: n_frames = 1, H = number of submitted rows
    int rows_mode;              // 1: row i of the buffers is call k0 + i of one run (line first_line + 2 i)
    int first_line;
    long long first_frame;      // frame number of frame 0 (rows mode: the frame of the run)
    int averaging;              // encoder inside ColorAveragingModem
    int line_shift, even_first, odd_first;   // LineConfig (line.py:49-65)
    float c0;                   // centre tap: 2 h[20] (decoder, up2) or h[20] (encoder, dn2)
    float taps[20];             // odd taps 2 h[2 j + 1] resp. h[2 j + 1], j = 0 .. 19
    float m[9];                 // decoder: (r, g, b) = m . (luma, dr, db);  encoder: (luma, dr, db) = m . (r, g, b)
};

// line.py:54-65: analog line parity against frame parity
__device__ __forceinline__ bool mac_alternate(const MacArgs &a, long long frame, int line) {
    const int adj = line + a.line_shift;
    const int analog = ((adj & 1) ? a.odd_first : a.even_first) + (adj >> 1);
    return (analog & 1) == (int)(frame & 1);
}

// Which rows of the buffers a workgroup walks: field `fld` of frame `f`, calls [k_begin, k_end) of that field's run.
struct MacSegment {
    int f, fld, k_begin, k_end, n_field;
};
__device__ __forceinline__ MacSegment mac_segment(const MacArgs &a) {
    MacSegment s;
    if (a.rows_mode) {
        s.f = 0;
        s.fld = 0;
        s.n_field = a.H;
        s.k_begin = xcd_block((int)blockIdx.x, (int)gridDim.x) * kMacSegment;
    } else {
        const int half = (a.H + 1) >> 1;
        const int segs = (half + kMacSegment - 1) / kMacSegment;
        const int b = xcd_block((int)blockIdx.x, (int)gridDim.x);
        const int seg = b % segs, ff = b / segs;
        s.f = ff >> 1;
        s.fld = ff & 1;
        s.n_field = (a.H - s.fld + 1) >> 1;
        s.k_begin = seg * kMacSegment;
    }
    s.k_end = s.k_begin + kMacSegment < s.n_field ? s.k_begin + kMacSegment : s.n_field;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------
// decoder: composite [F][H][1080] (rows mode [n][1080]) -> rgb [F][3][H][720] (rows mode [n][3][720])
// ---------------------------------------------------------------------------------------------------------------------
// chroma[i] of one line, mac.py:101-118 (sample 1 keeps 0.5: mac.py:113 assigns chroma[0:1]); p: the line, in LDS
template <class P>
__device__ __forceinline__ float mac_line_chroma(P p, int i) {
    if (i >= 5 && i <= 355) return p[13 + i];
    switch (i) {
        case 0:
        case 2: return 8.f * p[15] - 3.5f;
        case 1: return 0.5f;
        case 3: return 2.f * p[16] - 0.5f;
        case 4: return (p[17] - 0.0625f) / 0.875f;
        case 356: return (p[369] - 0.125f * p[372]) / 0.875f;
        case 357: return 2.f * p[370] - p[372];
        default: return 8.f * p[371] - 7.f * p[372];     // 358, 359
    }
}
// luma[n] of one line, mac.py:96-111
template <class P>
__device__ __forceinline__ float mac_line_luma(P p, int n) {
    if (n >= 11 && n <= 709) return p[361 + n];
    if (n < 8) n = 8;
    if (n > 712) n = 712;
    switch (n) {
        case 8: return 8.f * p[369] - 7.f * p[368];
        case 9: return 2.f * p[370] - p[368];
        case 10: return (p[371] - 0.125f * p[368]) / 0.875f;
        case 710: return (p[1071] - 0.0625f) / 0.875f;
        case 711: return 2.f * p[1072] - 0.5f;
        default: return 8.f * p[1073] - 3.5f;            // 712
    }
}

// Per row: the line arrives in LDS (16-byte loads, issued a row ahead and parked in registers), its chroma is laid out
// with the zero extension of resample_poly, and every thread below 180 turns 21 chroma samples into four interpolated
// ones (two of them 20-tap sums over the same window), adds the previous call's four from LDS and stores 3 x 16 bytes.
__global__ __launch_bounds__(kMacThreads) void mac_demod_kernel(const MacArgs a) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float lin[kMacLine];          // the line in work
    __shared__ __attribute__((aligned(16))) float ch[kMacChroma + 24];    // its chroma, 10 zeros before, 14 behind
    __shared__ __attribute__((aligned(16))) float up[2][kMacLuma];        // resample_poly(chroma, 2, 1) - 0.5 of this call and the one before
    const MacSegment s = mac_segment(a);
    if (s.k_begin >= s.k_end) return;
    const int t = threadIdx.x;
    const long long frame = a.first_frame + s.f;
    auto row_of = [&](int k) -> int { return a.rows_mode ? k : s.fld + 2 * k; };
    auto in_row = [&](int k) -> const f4 * { return (const f4 *)(a.in + ((long long)s.f * a.H + row_of(k)) * kMacLine); };
    constexpr int kQuads = kMacLine / 4;   // 270 16-byte pieces per line: thread t takes t and t + 256
    auto fetch = [&](int k, f4 &r0, f4 &r1) {
        const f4 *p = in_row(k);
        r0 = p[t];
        if (t + kMacThreads < kQuads) r1 = p[t + kMacThreads];
    };
    auto park = [&](const f4 &r0, const f4 &r1) {
        ((f4 *)lin)[t] = r0;
        if (t + kMacThreads < kQuads) ((f4 *)lin)[t + kMacThreads] = r1;
    };
    // the first call of a run has no previous chroma (zeros); a segment that starts inside a run walks the row before it
    // without writing anything
    const int k_first = s.k_begin > 0 ? s.k_begin - 1 : s.k_begin;
    f4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
    fetch(k_first, r0, r1);
    int cur = 0;
    bool have_prev = false;
    for (int k = k_first; k < s.k_end; ++k) {
        park(r0, r1);
        if (k + 1 < s.k_end) fetch(k + 1, r0, r1);      // the next line travels while this one is worked on
        __syncthreads();
        for (int i = t; i < kMacChroma + 24; i += kMacThreads)
            ch[i] = (i < 10 || i >= kMacChroma + 10) ? 0.f : mac_line_chroma(lin, i - 10);
        __syncthreads();
        if (t < kMacLuma / 4) {
            // outputs n = 4 t .. 4 t + 3 = samples 2 i, 2 i + 1, 2 i + 2, 2 i + 3 of up2(chroma), i = 2 t:
            //   y[2 i] = 2 h[20] c[i];  y[2 i + 1] = sum_j 2 h[2 j + 1] c[i + 10 - j]   (c = ch[. + 10])
            const int i = 2 * t;
            float w[24];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 v = *(const f2 *)(ch + i + 2 * q);       // ch[i .. i + 23] = c[i - 10 .. i + 13]; i is even: 8-byte pieces
                w[2 * q] = v.x; w[2 * q + 1] = v.y;
            }
            float o1 = 0.f, o3 = 0.f;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                o1 = __builtin_fmaf(a.taps[j], w[20 - j], o1);     // c[i + 10 - j] = ch[i + 20 - j]
                o3 = __builtin_fmaf(a.taps[j], w[21 - j], o3);     // c[i + 11 - j]
            }
            const f4 own = {a.c0 * w[10] - 0.5f, o1 - 0.5f, a.c0 * w[11] - 0.5f, o3 - 0.5f};
            f4 other = {0.f, 0.f, 0.f, 0.f};
            if (have_prev) other = *(const f4 *)(up[cur ^ 1] + 4 * t);
            *(f4 *)(up[cur] + 4 * t) = own;
            if (k >= s.k_begin) {
                const int line = a.rows_mode ? a.first_line + 2 * k : row_of(k);
                const bool alt = mac_alternate(a, frame, line);
                const f4 dr = alt ? other : own, db = alt ? own : other;   // mac.py:115-120
                f4 luma;
                luma.x = mac_line_luma(lin, 4 * t);
                luma.y = mac_line_luma(lin, 4 * t + 1);
                luma.z = mac_line_luma(lin, 4 * t + 2);
                luma.w = mac_line_luma(lin, 4 * t + 3);
                float *o = a.rows_mode ? a.out + (long long)k * 3 * kMacLuma
                                       : a.out + (((long long)s.f * 3) * a.H + row_of(k)) * kMacLuma;
                const long long plane = a.rows_mode ? kMacLuma : (long long)a.H * kMacLuma;
                *(f4 *)(o + 4 * t) = a.m[0] * luma + a.m[1] * dr + a.m[2] * db;
                *(f4 *)(o + plane + 4 * t) = a.m[3] * luma + a.m[4] * dr + a.m[5] * db;
                *(f4 *)(o + 2 * plane + 4 * t) = a.m[6] * luma + a.m[7] * dr + a.m[8] * db;
            }
        }
        have_prev = true;
        cur ^= 1;
        __syncthreads();      // lin / ch are free for the next line
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// encoder: rgb [F][3][H][720] (rows mode [n][3][720]) -> composite [F][H][1080] (rows mode [n][1080])
// ---------------------------------------------------------------------------------------------------------------------
// Per output row: the new input row of the call arrives as 3 x 16 bytes per thread (issued a row ahead, parked in
// registers), is turned into (luma, dr, db) on its way into LDS, where the previous call's components still sit; the
// colour-difference signal of the line (mean of the two calls inside ColorAveragingModem) is laid out with the zero
// extension of resample_poly, and every thread below 270 forms four samples of the line and stores 16 bytes.
__global__ __launch_bounds__(kMacThreads) void mac_mod_kernel(const MacArgs a) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float comp[2][3][kMacLuma];   // (luma, dr, db) of this call's row and the previous one's
    // colour-difference signal c of the line by parity: ce[i] = c[2 i]; co[m + 10] = c[2 m + 1], 10 zeros on either side
    // (the zero extension of resample_poly); c3 = the 360 chroma samples of the line
    __shared__ float ce[kMacChroma], co[kMacChroma + 20], c3[kMacChroma];
    const MacSegment s = mac_segment(a);
    if (s.k_begin >= s.k_end) return;
    const int t = threadIdx.x;
    const long long frame = a.first_frame + s.f;
    const long long plane = a.rows_mode ? kMacLuma : (long long)a.H * kMacLuma;
    auto in_row = [&](int row) -> const float * {
        return a.rows_mode ? a.in + (long long)row * 3 * kMacLuma : a.in + (((long long)s.f * 3) * a.H + row) * kMacLuma;
    };
    // rows of the two calls that meet in output row k (comb.py:141-152, image.py:47-55): a = the previous call's input
    // (luma, half of the chroma), b = this call's
    auto rows_of = [&](int k, int &row_a, int &row_b) {
        if (a.rows_mode) {
            row_b = k;
            row_a = a.averaging && k > 0 ? k - 1 : k;
        } else {
            row_a = row_b = s.fld + 2 * k;
            if (a.averaging) {     // modulation_delay 1: this call's input is row y + 2, stepped back into the image
                row_b += 2;
                while (row_b >= a.H) row_b -= 2;
            }
        }
    };
    constexpr int kQuads = kMacLuma / 4;   // 180 16-byte pieces per plane
    auto fetch = [&](int row, f4 &r, f4 &g, f4 &b) {
        if (t < kQuads) {
            const float *p = in_row(row) + 4 * t;
            r = *(const f4 *)p;
            g = *(const f4 *)(p + plane);
            b = *(const f4 *)(p + 2 * plane);
        }
    };
    auto park = [&](int slot, const f4 &r, const f4 &g, const f4 &b) {
        if (t < kQuads) {
            *(f4 *)(comp[slot][0] + 4 * t) = a.m[0] * r + a.m[1] * g + a.m[2] * b;
            *(f4 *)(comp[slot][1] + 4 * t) = a.m[3] * r + a.m[4] * g + a.m[5] * b;
            *(f4 *)(comp[slot][2] + 4 * t) = a.m[6] * r + a.m[7] * g + a.m[8] * b;
        }
    };
    f4 r = {0.f, 0.f, 0.f, 0.f}, g = r, b = r;
    int row_a, row_b;
    rows_of(s.k_begin, row_a, row_b);
    int cur = 0;
    if (row_a != row_b) {      // a segment inside a run: the previous call's row comes first
        fetch(row_a, r, g, b);
        park(1, r, g, b);
    }
    fetch(row_b, r, g, b);
    for (int k = s.k_begin; k < s.k_end; ++k) {
        rows_of(k, row_a, row_b);
        park(cur, r, g, b);
        if (k + 1 < s.k_end) {
            int na, nb;
            rows_of(k + 1, na, nb);
            fetch(nb, r, g, b);       // the next call's row travels while this line is formed
        }
        const int sa = row_a != row_b ? cur ^ 1 : cur;
        const int line = a.rows_mode ? a.first_line + 2 * k - (a.averaging ? 2 : 0) : s.fld + 2 * k;   // comb.py:152: line - 2
        const int sel = mac_alternate(a, frame, line) ? 2 : 1;     // mac.py:44-47
        __syncthreads();
        for (int i = t; i < kMacLuma; i += kMacThreads) {
            const float c = sa != cur ? 0.5f * (comp[cur][sel][i] + comp[sa][sel][i]) : comp[cur][sel][i];   // comb.py:147-148
            if (i & 1) co[(i >> 1) + 10] = c; else ce[i >> 1] = c;
        }
        if (t < 10) co[t] = co[kMacChroma + 10 + t] = 0.f;
        __syncthreads();
        // chroma[i] = resample_poly(c, 1, 2)[i] + 0.5 = h[20] c[2 i] + sum_j h[2 j + 1] c[2 (i + 9 - j) + 1] + 0.5
        for (int i = t; i < kMacChroma; i += kMacThreads) {
            float v = a.c0 * ce[i];
#pragma unroll
            for (int j = 0; j < 20; ++j) v = __builtin_fmaf(a.taps[j], co[i + 19 - j], v);
            c3[i] = v + 0.5f;
        }
        __syncthreads();
        const float *lum = comp[sa][0];
        auto chroma = [&](int i) -> float { return c3[i]; };
        auto sample = [&](int n) -> float {      // mac.py:56-69
            if (n >= 18 && n <= 368) return chroma(n - 13);
            if (n >= 372 && n <= 1070) return lum[n - 361];
            switch (n) {
                case 15: return 0.4375f + 0.125f * chroma(2);
                case 16: return 0.25f + 0.5f * chroma(3);
                case 17: return 0.0625f + 0.875f * chroma(4);
                case 369: return 0.875f * chroma(356) + 0.125f * lum[8];
                case 370: return 0.5f * chroma(357) + 0.5f * lum[9];
                case 371: return 0.125f * chroma(358) + 0.875f * lum[10];
                case 1071: return 0.0625f + 0.875f * lum[710];
                case 1072: return 0.25f + 0.5f * lum[711];
                case 1073: return 0.4375f + 0.125f * lum[712];
                default: return 0.5f;
            }
        };
        const int out_row = a.rows_mode ? k : s.fld + 2 * k;
        float *o = a.out + ((long long)s.f * a.H + out_row) * kMacLine;
        for (int q = t; q < kMacLine / 4; q += kMacThreads) {
            f4 v;
            v.x = sample(4 * q);
            v.y = sample(4 * q + 1);
            v.z = sample(4 * q + 2);
            v.w = sample(4 * q + 3);
            *(f4 *)(o + 4 * q) = v;
        }
        cur ^= 1;
        // the next park() overwrites the older slot; ce / co / c3 are rewritten behind the next barriers
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The cases that resample (mac.py:49-55, 71-74, 88-91): rows of a length other than 720 samples, lines of a length other
// than 1080 (MacVariant.D2MAC_7MHZ: 720).  scipy's resample_poly(x, up, down) written out (zero extension):
//     y[n] = sum_j up h[j] xu[n down + half_len - j],  xu = x with up - 1 zeros after every sample,
//     h = firwin(2 half_len + 1, 1 / max(up, down), ('kaiser', 5.0)),  half_len = 10 max(up, down)
// i.e. about 21 products per output whatever the ratio.  One workgroup per call, everything of the call in LDS; these are
// plain kernels for the uncommon shapes, the tuned pair above serves 720 <-> 1080.
// ---------------------------------------------------------------------------------------------------------------------
struct MacFir {
    const float *h;              // up * firwin(...), 2 half_len + 1 taps, device memory; null: up == down (a copy)
    int up, down, half_len;
    int stage;                   // 1: the kernel copies the taps into LDS first (the host sets it while they fit)
};
struct MacGenArgs {
    MacArgs a;
    int W, CW;                   // samples per rgb row / per transmitted line
    MacFir luma_in, chroma_in, line_out, line_in;   // W -> 720, W -> 360, 1080 -> CW, CW -> 1080
};
// Taps read in the product loop come from LDS where they fit (the loop's loads depend on each other through the
// address arithmetic only, but hipcc does not batch them: from global memory their latency is the kernel's time).
__device__ __forceinline__ float *mac_stage_taps(MacFir &f, float *lds, int t) {
    if (!f.h || !f.stage) return lds;
    const int n = 2 * f.half_len + 1;
    for (int i = t; i < n; i += kMacThreads) lds[i] = f.h[i];
    f.h = lds;
    return lds + n;
}
__device__ __forceinline__ float mac_resample_at(const float *src, int n_in, const MacFir &f, int n) {
    if (!f.h) return src[n];
    const int t0 = n * f.down + f.half_len;
    int j = t0 % f.up, i = t0 / f.up;
    if (i > n_in - 1) {          // taps that would meet samples behind the row
        const int skip = i - (n_in - 1);
        j += skip * f.up;
        i -= skip;
    }
    float acc = 0.f;
    for (; j <= 2 * f.half_len && i >= 0; j += f.up, --i) acc = __builtin_fmaf(f.h[j], src[i], acc);
    return acc;
}

// U8: the ImageModem byte boundary fused in (image.py:7-8, 24-25, 62): lines are uint8 [..][CW], entering as
// (5 (byte / 255) - 1) / 3; the output is interleaved uint8 RGB [..][720][3] = rint(255 clip(x, 0, 1)).
template <bool U8>
__global__ __launch_bounds__(kMacThreads) void mac_demod_generic_kernel(const MacGenArgs ga) {
    const MacArgs &a = ga.a;
    extern __shared__ __attribute__((aligned(16))) float mac_lds[];
    float *lin = mac_lds;                       // [1080] the line at its own rate
    float *ch = lin + kMacLine;                 // [360 + 24]
    float *up = ch + kMacChroma + 24;           // [2][720] interpolated chroma of this call and of the one before
    float *raw = up + 2 * kMacLuma;             // [CW] the transmitted line
    const MacSegment s = mac_segment(a);        // a segment of consecutive calls of one field, like the tuned kernel
    if (s.k_begin >= s.k_end) return;
    const int t = threadIdx.x;
    MacFir line_in = ga.line_in;
    mac_stage_taps(line_in, raw + ga.CW, t);
    const long long frame = a.first_frame + s.f;
    auto row_of = [&](int k) -> int { return a.rows_mode ? k : s.fld + 2 * k; };
    const int k_first = s.k_begin > 0 ? s.k_begin - 1 : s.k_begin;   // a segment inside a run walks the row before it silently
    int cur = 0;
    bool have_prev = false;
    for (int k = k_first; k < s.k_end; ++k) {
        const int row = row_of(k);
        const long long row_index = (long long)s.f * a.H + row;
        const float *p = a.in + row_index * ga.CW;
        const unsigned char *p8 = (const unsigned char *)a.in + row_index * ga.CW;
        __syncthreads();
        for (int i = t; i < ga.CW; i += kMacThreads)
            raw[i] = U8 ? __builtin_fmaf((float)p8[i], 5.0f / (255.0f * 3.0f), -1.0f / 3.0f) : p[i];
        __syncthreads();
        for (int n = t; n < kMacLine; n += kMacThreads) lin[n] = mac_resample_at(raw, ga.CW, line_in, n);    // mac.py:88-91
        __syncthreads();
        for (int i = t; i < kMacChroma + 24; i += kMacThreads)
            ch[i] = (i < 10 || i >= kMacChroma + 10) ? 0.f : mac_line_chroma(lin, i - 10);
        __syncthreads();
        for (int n = t; n < kMacLuma; n += kMacThreads) {
            const int i = n >> 1;
            float v;
            if (n & 1) {
                v = 0.f;
                for (int j = 0; j < 20; ++j) v = __builtin_fmaf(a.taps[j], ch[i + 20 - j], v);
            } else {
                v = a.c0 * ch[i + 10];
            }
            up[cur * kMacLuma + n] = v - 0.5f;
        }
        __syncthreads();
        if (k >= s.k_begin) {
            const int line = a.rows_mode ? a.first_line + 2 * k : row;
            const bool alt = mac_alternate(a, frame, line);
            float *o = a.rows_mode ? a.out + (long long)row * 3 * kMacLuma : a.out + (((long long)s.f * 3) * a.H + row) * kMacLuma;
            unsigned char *o8 = (unsigned char *)a.out + ((long long)s.f * a.H + row) * 3 * kMacLuma;
            const long long plane = a.rows_mode ? kMacLuma : (long long)a.H * kMacLuma;
            for (int n = t; n < kMacLuma; n += kMacThreads) {
                const float luma = mac_line_luma(lin, n);
                const float own = up[cur * kMacLuma + n], other = have_prev ? up[(cur ^ 1) * kMacLuma + n] : 0.f;
                const float dr = alt ? other : own, db = alt ? own : other;
                const float r = __builtin_fmaf(a.m[0], luma, __builtin_fmaf(a.m[1], dr, a.m[2] * db));
                const float g = __builtin_fmaf(a.m[3], luma, __builtin_fmaf(a.m[4], dr, a.m[5] * db));
                const float b = __builtin_fmaf(a.m[6], luma, __builtin_fmaf(a.m[7], dr, a.m[8] * db));
                if (U8) {
                    // v_cvt_pk_u8_f32 = uint8(rint(255 clip(x, 0, 1))) exactly (tools/ubench_cvt_u8.hip)
                    o8[3 * n] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(255.f * r, 0, 0);
                    o8[3 * n + 1] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(255.f * g, 0, 0);
                    o8[3 * n + 2] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(255.f * b, 0, 0);
                } else {
                    o[n] = r;
                    o[plane + n] = g;
                    o[2 * plane + n] = b;
                }
            }
        }
        have_prev = true;
        cur ^= 1;
    }
}

// U8: interleaved uint8 RGB rows [..][W][3] enter as byte / 255 (image.py:43-45); the line leaves as
// uint8 = rint(255 clip(0.6 x + 0.2, 0, 1)) (image.py:20-21, 7-8).
template <bool U8>
__global__ __launch_bounds__(kMacThreads) void mac_mod_generic_kernel(const MacGenArgs ga) {
    const MacArgs &a = ga.a;
    extern __shared__ __attribute__((aligned(16))) float mac_lds[];
    const int W = ga.W;
    float *lin = mac_lds;                       // [1080]
    float *lum = lin + kMacLine;                // [720]
    float *c3 = lum + kMacLuma;                 // [360]
    float *cavg = c3 + kMacChroma;              // [W] colour-difference signal of the line at the rows' rate
    float *comp = cavg + W;                     // [2][3][W] (luma, dr, db) of the previous call's row and of this call's
    const int t = threadIdx.x;
    MacFir luma_in = ga.luma_in, chroma_in = ga.chroma_in, line_out = ga.line_out;
    mac_stage_taps(line_out, mac_stage_taps(chroma_in, mac_stage_taps(luma_in, comp + (a.averaging ? 6 : 3) * W, t), t), t);   // the second row slot exists with averaging only
    int f, out_row, row_a, row_b, line;
    if (a.rows_mode) {
        f = 0; out_row = xcd_block((int)blockIdx.x, (int)gridDim.x); row_b = out_row; row_a = a.averaging && out_row > 0 ? out_row - 1 : out_row;
        line = a.first_line + 2 * out_row - (a.averaging ? 2 : 0);
    } else {
        const int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
        f = bid / a.H; out_row = bid % a.H; row_a = row_b = out_row; line = out_row;
        if (a.averaging) {
            row_b += 2;
            while (row_b >= a.H) row_b -= 2;
        }
    }
    const long long frame = a.first_frame + f;
    const long long plane = a.rows_mode ? W : (long long)a.H * W;
    for (int slot = 0; slot < (row_a != row_b ? 2 : 1); ++slot) {
        const int row = slot ? row_b : row_a;
        const float *p = a.rows_mode ? a.in + (long long)row * 3 * W : a.in + (((long long)f * 3) * a.H + row) * W;
        const unsigned char *p8 = (const unsigned char *)a.in + ((long long)f * a.H + row) * 3 * W;
        for (int i = t; i < W; i += kMacThreads) {
            float r, g, b;
            if (U8) { r = (float)p8[3 * i] / 255.0f; g = (float)p8[3 * i + 1] / 255.0f; b = (float)p8[3 * i + 2] / 255.0f; }
            else { r = p[i]; g = p[plane + i]; b = p[2 * plane + i]; }
            comp[(slot * 3 + 0) * W + i] = __builtin_fmaf(a.m[0], r, __builtin_fmaf(a.m[1], g, a.m[2] * b));
            comp[(slot * 3 + 1) * W + i] = __builtin_fmaf(a.m[3], r, __builtin_fmaf(a.m[4], g, a.m[5] * b));
            comp[(slot * 3 + 2) * W + i] = __builtin_fmaf(a.m[6], r, __builtin_fmaf(a.m[7], g, a.m[8] * b));
        }
    }
    __syncthreads();
    const int sel = mac_alternate(a, frame, line) ? 2 : 1;
    for (int i = t; i < W; i += kMacThreads)
        cavg[i] = row_a != row_b ? 0.5f * (comp[(3 + sel) * W + i] + comp[sel * W + i]) : comp[sel * W + i];
    __syncthreads();
    for (int n = t; n < kMacLuma; n += kMacThreads) lum[n] = mac_resample_at(comp, W, luma_in, n);                 // mac.py:49-52
    for (int i = t; i < kMacChroma; i += kMacThreads) c3[i] = mac_resample_at(cavg, W, chroma_in, i) + 0.5f;       // mac.py:53-55, 57
    __syncthreads();
    for (int n = t; n < kMacLine; n += kMacThreads) {        // mac.py:56-69
        float v = 0.5f;
        if (n >= 18 && n <= 368) v = c3[n - 13];
        else if (n >= 372 && n <= 1070) v = lum[n - 361];
        else if (n == 15) v = 0.4375f + 0.125f * c3[2];
        else if (n == 16) v = 0.25f + 0.5f * c3[3];
        else if (n == 17) v = 0.0625f + 0.875f * c3[4];
        else if (n == 369) v = 0.875f * c3[356] + 0.125f * lum[8];
        else if (n == 370) v = 0.5f * c3[357] + 0.5f * lum[9];
        else if (n == 371) v = 0.125f * c3[358] + 0.875f * lum[10];
        else if (n == 1071) v = 0.0625f + 0.875f * lum[710];
        else if (n == 1072) v = 0.25f + 0.5f * lum[711];
        else if (n == 1073) v = 0.4375f + 0.125f * lum[712];
        lin[n] = v;
    }
    __syncthreads();
    float *o = a.out + ((long long)f * a.H + out_row) * ga.CW;
    unsigned char *o8 = (unsigned char *)a.out + ((long long)f * a.H + out_row) * ga.CW;
    for (int m = t; m < ga.CW; m += kMacThreads) {
        const float v = mac_resample_at(lin, kMacLine, line_out, m);              // mac.py:71-74
        if (U8) o8[m] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(255.f * __builtin_fmaf(0.6f, v, 0.2f), 0, 0);
        else o[m] = v;
    }
}

}  // namespace cm
#endif

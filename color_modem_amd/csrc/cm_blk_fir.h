// cm_blk_fir.h - one 20-tap half-band FIR chain of the time-blocked decoder on the matrix pipe (gfx950).
//
// A lane owns a scan line and advances it 16 samples at a time.  y[t] = sum_j g[j] x[t - j], j = 0 .. 19, of the 16 new
// outputs of 64 lines is a Toeplitz product on v_mfma_f32_16x16x32_f16: time on M (16 outputs), lines on N (four groups of
// 16), the 32-sample window [block start - 16, block start + 16) on K; the three taps that reach further back (6 products)
// stay on the vector pipe.  Data and taps are split into two float16 pieces each (hi.hi + hi.lo + lo.hi with float32
// accumulation: 2.5e-7 of full scale, the same as a float32 fmaf chain - profiles/r02_ubench_mfma_f16_fir.txt).
//
// The data operand goes through LDS: a lane writes the 16 new samples of its line as 4 x 16 bytes (hi[0..7], hi[8..15],
// lo[0..7], lo[8..15]) into a 4 KiB *slot* [64 lines][64 bytes]; operand lane l of line group G reads line 16 G + (l & 15),
// k-slice l >> 4: slices 0, 1 from the slot the chain wrote one block earlier (its history), slices 2, 3 from the new one -
// no register history and no cross-lane moves on the way in.  The 16-byte chunks of a line are XOR-swizzled with
// (-(line >> 2)) & 3 so that the sixteen lanes of a ds_read_b128 group hit sixteen different 16-byte bank slots.
// The results (lane = line 16 G + (l & 15) of group G, times 4 (l >> 4) + r) come back to "all 16 outputs of my own line"
// by a 4 x 4 transpose between register index and lane quarter: v_permlane32_swap + v_permlane16_swap.
// Slots rotate: run n of the kernel (5 chains per block) writes slot n mod 6 and finds its history in slot (n + 1) mod 6.
#ifndef CM_BLK_FIR_H
#define CM_BLK_FIR_H

namespace cm {

typedef _Float16 blk_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 blk_h2 __attribute__((ext_vector_type(2)));
typedef float blk_f2 __attribute__((ext_vector_type(2)));
typedef float blk_f4 __attribute__((ext_vector_type(4)));
typedef unsigned blk_u4 __attribute__((ext_vector_type(4)));

constexpr int kBlk = 16;                 // samples per block
constexpr int kBlkSlots = 6;
constexpr int kBlkSlotBytes = 64 * 64;   // [64 lines][hi 32 bytes | lo 32 bytes]
constexpr float kBlkScale = 1.f;         // taps enter the matrix pipe times this (the un-normalised sections of the recursive filters already lift the streams by 10 - 1000: a larger factor overflows float16 at the detector low-pass)

// Toeplitz operand of the taps: lane l holds row i = l & 15 (output tb + i), k = 8 (l >> 4) + j (window sample tb - 16 + k):
// tap index i + 16 - k where that is within 0 .. 19.  Two float16 pieces.
struct BlkTiles {
    blk_h8 hi, lo;
};
// tap index of element j of lane l, or -1
inline __host__ __device__ int blk_tile_tap(int l, int j) {
    const int kk = (l & 15) + 16 - (8 * (l >> 4) + j);
    return kk >= 0 && kk < 20 ? kk : -1;
}

#ifdef __HIPCC__
typedef __attribute__((address_space(3))) unsigned char blk_lds_byte;
typedef __attribute__((address_space(3))) blk_u4 blk_lds_u4;
typedef __attribute__((address_space(3))) blk_h8 blk_lds_h8;

// two samples -> packed float16 high and low pieces
__device__ __forceinline__ void blk_split2(float x0, float x1, unsigned &hi, unsigned &lo) {
    const blk_h2 ph = __builtin_convertvector((blk_f2){x0, x1}, blk_h2);
    const blk_f2 back = __builtin_convertvector(ph, blk_f2);
    const blk_h2 pl = __builtin_convertvector((blk_f2){x0 - back[0], x1 - back[1]}, blk_h2);
    hi = __builtin_bit_cast(unsigned, ph);
    lo = __builtin_bit_cast(unsigned, pl);
}

// per-lane byte offsets inside a slot
struct BlkAddr {
    int wr;        // own line, chunk 0; chunk c: wr ^ (c << 4)
    int rd;        // line lane & 15, the hi chunk of k-slice (lane >> 4) & 1; lo: rd ^ 32; line group G: + 1024 G
    bool hist;     // this lane's k-slice lies in the history slot
    __device__ __forceinline__ void init(int lane) {
        wr = 64 * lane + 16 * ((-(lane >> 2)) & 3);
        const int l15 = lane & 15, kk = lane >> 4;
        rd = 64 * l15 + 16 * ((kk & 1) ^ ((-(l15 >> 2)) & 3));
        hist = kk < 2;
    }
};

// slot rotation of the kernel: cur = byte offset of the slot this run writes, hist = of the slot with the chain's history
struct BlkSlots {
    int cur, hist;
    __device__ __forceinline__ void init() { cur = 0; hist = kBlkSlotBytes; }
    __device__ __forceinline__ void next() {
        cur = hist;
        hist = hist + kBlkSlotBytes == kBlkSlots * kBlkSlotBytes ? 0 : hist + kBlkSlotBytes;
    }
};

// State of one chain between blocks: samples 13 .. 15 of the last two blocks (the products behind the window).
struct BlkFir {
    float p[3], q[3];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = q[i] = 0.f;
    }
    // in[16] -> out[16] = kBlkScale * (FIR of the stream); g17 .. g19: the last three taps times kBlkScale.
    // ops: the slots; the caller advances `sl` after the call.
    __device__ __forceinline__ void run(const float (&in)[kBlk], float (&out)[kBlk], const BlkTiles &tl, float g17, float g18, float g19,
                                        blk_lds_byte *ops, const BlkAddr &ad, const BlkSlots &sl) {
        unsigned hi[8], lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) blk_split2(in[2 * i], in[2 * i + 1], hi[i], lo[i]);
        blk_lds_byte *w = ops + sl.cur;
        *(blk_lds_u4 *)(w + ad.wr) = (blk_u4){hi[0], hi[1], hi[2], hi[3]};
        *(blk_lds_u4 *)(w + (ad.wr ^ 16)) = (blk_u4){hi[4], hi[5], hi[6], hi[7]};
        *(blk_lds_u4 *)(w + (ad.wr ^ 32)) = (blk_u4){lo[0], lo[1], lo[2], lo[3]};
        *(blk_lds_u4 *)(w + (ad.wr ^ 48)) = (blk_u4){lo[4], lo[5], lo[6], lo[7]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int ro = (ad.hist ? sl.hist : sl.cur) + ad.rd;
        blk_h8 bh[4], bl[4];
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            bh[G] = *(const blk_lds_h8 *)(ops + ro + 1024 * G);
            bl[G] = *(const blk_lds_h8 *)(ops + (ro ^ 32) + 1024 * G);
        }
        __builtin_amdgcn_sched_barrier(0);      // all eight loads in flight before the first product waits
        blk_f4 acc[4];
#pragma unroll
        for (int G = 0; G < 4; ++G) acc[G] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl.lo, bh[G], (blk_f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int G = 0; G < 4; ++G) acc[G] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl.hi, bl[G], acc[G], 0, 0, 0);
#pragma unroll
        for (int G = 0; G < 4; ++G) acc[G] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl.hi, bh[G], acc[G], 0, 0, 0);
        // acc[G][r] at lane (a', b) = line 16 G + b, time 4 a' + r  ->  out[4 a' + r] at lane (G, b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // (clang folds __builtin_bit_cast of a vector ELEMENT expression to element 0: go through scalars)
            const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r];
            const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a2), false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1), __float_as_uint(a3), false, false);
            const auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
            const auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
            out[r] = __uint_as_float(t01[0]);
            out[4 + r] = __uint_as_float(t01[1]);
            out[8 + r] = __uint_as_float(t23[0]);
            out[12 + r] = __uint_as_float(t23[1]);
        }
        out[0] = __builtin_fmaf(g17, p[2], __builtin_fmaf(g18, p[1], __builtin_fmaf(g19, p[0], out[0])));
        out[1] = __builtin_fmaf(g18, p[2], __builtin_fmaf(g19, p[1], out[1]));
        out[2] = __builtin_fmaf(g19, p[2], out[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) { p[i] = q[i]; q[i] = in[13 + i]; }
    }
};
#endif  // __HIPCC__

}  // namespace cm
#endif

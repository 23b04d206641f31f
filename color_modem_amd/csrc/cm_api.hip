// cm_api.hip - C ABI of libcolor_modem_hip.so (include/color_modem_hip.h): plan management and
// kernel launches.  There is deliberately no host fallback in this file.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

// CM_PART: this file compiles as one translation unit (0, the default) or as four that __graft_entry__.build() compiles side by side and
// links into the one library - 1: the QAM / SECAM / MAC / wrapped-comb entry points and their streaming kernels (decoder instances of the
// PAL-BG filter shapes), 2: the cm_am_* entry points (Proto-SECAM / NIIR, streaming and scan kernels), 3: the row-parallel scan kernels of
// part 1's families behind five launch functions (cm_host::scan_launch_*), 4: the decoder instances of every other filter-set shape (NTSC /
// PAL-M/N, NTSC-I, NTSC-A, the 640 / 704 / 768 sample rasters, the run-time shape) behind cm_host::select_other_shapes.  The helpers at the
// top are in every part; the process-wide state (last error, pointer check) lives in part 1.
#ifndef CM_PART
#define CM_PART 0
#endif
#define CM_MAIN_PART (CM_PART == 0 || CM_PART == 1)
#define CM_AM_PART (CM_PART == 0 || CM_PART == 2)
#define CM_SCAN_PART (CM_PART == 0 || CM_PART == 3)
#define CM_SHAPES_PART (CM_PART == 0 || CM_PART == 4)
// 5 .. 7 (round 6): the tuned decoder instances of the wide rasters (cm_shapes_wide.h, generated) behind cm_host::select_wide_* - 5: the PAL-D front end,
// 6: the QAM front end on the PAL shapes, 7: the NTSC shapes
#define CM_WIDE_PALD_PART (CM_PART == 0 || CM_PART == 5)
#define CM_WIDE_PAL_QAM_PART (CM_PART == 0 || CM_PART == 6)
#define CM_WIDE_NTSC_PART (CM_PART == 0 || CM_PART == 7)
#define CM_WIDE_PART (CM_WIDE_PALD_PART || CM_WIDE_PAL_QAM_PART || CM_WIDE_NTSC_PART)
#define CM_DEMOD_PART (CM_MAIN_PART || CM_SHAPES_PART || CM_WIDE_PART)      /* parts that launch demod_pair_kernel instances */

#include "../../include/color_modem_hip.h"

#ifdef CM_HOST_DRY_RUN
// Sanitizer build of the HOST code (tests/test_host_sanitize.py; round 6): `hipcc -fsanitize=address,undefined -DCM_HOST_DRY_RUN` makes a library whose plan constructors - descriptor validation, instance selection, every table builder and
// coefficient conversion of cm_plan.h / cm_am_plan.h - run on a box WITHOUT a GPU: "device" tables are host allocations (so that ASan
// watches every byte the constructors write), a device count of one is reported, and the compute entry points are never reached by that
// test.  Nothing of this is compiled into the product library: without the macro a plan constructor answers CM_ERR_NO_DEVICE there.
#include <cstdlib>
namespace cm_dry {
inline hipError_t malloc_(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t free_(void *p) { std::free(p); return hipSuccess; }
inline hipError_t memcpy_(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t memset_(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t get_device_(int *d) { *d = 0; return hipSuccess; }
}  // namespace cm_dry
#define hipMalloc(p, n) cm_dry::malloc_((void **)(p), (n))
#define hipFree(p) cm_dry::free_((void *)(p))
#define hipMemcpy(d, s, n, k) cm_dry::memcpy_((void *)(d), (const void *)(s), (n), (k))
#define hipMemset(d, v, n) cm_dry::memset_((void *)(d), (v), (n))
#define hipGetDevice(d) cm_dry::get_device_(d)
#endif

#include "cm_kernels.h"
#include "cm_mod_kernels.h"
#include "cm_secam_kernels.h"
#if CM_MAIN_PART
#include "cm_mac_kernels.h"
#endif
#include "cm_plan.h"
#include "cm_shapes_wide.h"
#if CM_AM_PART
#include "cm_am_kernels.h"
#endif
#if CM_MAIN_PART
#include "cm_wrap_kernels.h"
#endif
#include "cm_scan_kernels.h"
#if CM_AM_PART
#include "cm_am_scan_kernels.h"
#endif
#if (CM_MAIN_PART || CM_SHAPES_PART) && defined(CM_EXPERIMENTS)
#include "cm_blk_kernels.h"      // the time-blocked decoder with the FIRs on the matrix pipe (round 2's experiment)
#endif
#if CM_AM_PART
#include "cm_am_plan.h"
#endif

constexpr int kModAnyShift = 12;   // luma delay window of the run-time-shape modulators (pre-correction shift <= 12)

#ifndef CM_PAIR
#define CM_PAIR 1   /* 1: wave-pair kernels (demod_pair_kernel) where they fit; 0: one wave per 64 calls (demod_kernel) everywhere */
#endif

using namespace cm;

namespace cm_host {      // process-wide state shared by the parts
#if CM_PART >= 2
extern thread_local std::string g_error;
extern bool g_pointer_check;
#else
thread_local std::string g_error;
bool g_pointer_check = true;
#endif
// the scan kernels of the QAM / SECAM families (CM_PART 3): c1 = samples per lane, the constants are device pointers of the plan
int scan_launch_demod(int c1, bool u8, int device, const ScanK *km, const ScanK *kf, int depth, const Geom &gm, const Geom &gf, bool with_first,
                      hipStream_t stream);
int scan_launch_qam_mod(int c1, bool u8, int device, const ScanModK *k, const Geom &g, hipStream_t stream);
int scan_launch_secam_mod(int c1, bool u8, int device, const ScanSecamModK *k, const Geom &g, hipStream_t stream);
int scan_launch_secam_demod(int c1, bool u8, int device, const ScanSecamK *k, const Geom &g, hipStream_t stream);
int scan_launch_wrap_back(int c1, bool u8, int device, const ScanModK *k, const ScanWrapArgs &a, const Geom &g, hipStream_t stream);
// the decoder instances of every filter-set shape but PAL-BG's (CM_PART 4)
bool select_other_shapes(cm_plan *p, const cm_plan_desc &d, std::string &err);
// the tuned instances of the wide rasters (CM_PART 5 .. 7): 1 = selected, 0 = failed (err), -1 = no tuned instance for this plan (the run-time shape takes it)
int select_wide_pald(cm_plan *p, const cm_plan_desc &d, std::string &err);
int select_wide_pal_qam(cm_plan *p, const cm_plan_desc &d, std::string &err);
int select_wide_ntsc(cm_plan *p, const cm_plan_desc &d, std::string &err);
}  // namespace cm_host
using cm_host::g_error;
using cm_host::g_pointer_check;

namespace {

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
// Dynamic LDS beyond 64 KB has to be allowed per kernel and per device: done once for each (kernel, device) of the process.
int allow_dynamic_lds(const void *kernel, int device, size_t bytes, const char *what) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, device})) return CM_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(CM_ERR_LAUNCH, std::string("hipFuncSetAttribute(max dynamic LDS) failed for ") + what);
    done.insert({kernel, device});
    return CM_OK;
}
#ifdef CM_HOST_DRY_RUN   /* the host sanitizer build never launches: no kernel instance is referenced, so none is compiled (a build of seconds) */
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) ((void)0)
#define allow_dynamic_lds(...) CM_OK
#endif
// A plan's tables live on the device that was current in cm_*_plan_create.  Every compute entry point checks that this
// device is still the current one and that both image buffers are device memory of it: a plan used under another current
// device, or fed another GPU's pointers, would otherwise fault inside the kernel (or run over peer access) instead of
// returning an error.  Rejected: device memory of another GPU, pageable host memory, pointers the runtime cannot classify
// (a kernel fault takes more than the process down on a shared node).  Pinned / mapped host memory and managed memory are
// device-accessible and pass.  cm_set_pointer_check(0) drops the two hipPointerGetAttributes calls for callers whose
// allocator the runtime does not know (a few microseconds per call less, too); -DCM_NO_POINTER_CHECK compiles them out.
int check_device(int plan_device, const void *a, const void *b) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(CM_ERR_NO_DEVICE, "hipGetDevice failed");
    if (cur != plan_device)
        return fail(CM_ERR_INVALID, "the plan belongs to HIP device " + std::to_string(plan_device) + ", the current device is " +
                                        std::to_string(cur));
#ifndef CM_NO_POINTER_CHECK
    const void *ptrs[2] = {a, b};
    for (const void *ptr : ptrs) {
        if (!ptr || !g_pointer_check) continue;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
            (void)hipGetLastError();
            return fail(CM_ERR_INVALID, "an image buffer is not memory the HIP runtime knows as device-accessible (the ABI takes device "
                                        "pointers; cm_set_pointer_check(0) skips this check)");
        }
        if (at.type == hipMemoryTypeDevice && at.device != plan_device)
            return fail(CM_ERR_INVALID, "an image buffer lives on HIP device " + std::to_string(at.device) + ", the plan on device " +
                                            std::to_string(plan_device));
        if (at.type == hipMemoryTypeUnregistered)
            return fail(CM_ERR_INVALID, "an image buffer is pageable host memory (the ABI takes device pointers)");
    }
#endif
    return CM_OK;
}
#define HIP_TRY(expr, code)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(code, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#if CM_DEMOD_PART
// One launch = first-line workgroups [0, n_first) followed by the main pass's workgroups.
typedef int (*LaunchFn)(const Geom &gm, const void *km, const Geom &gf, const void *kf, int n_first, int n_main,
                        hipStream_t);

template <class Main, class First>
int launch_demod(const Geom &gm, const void *km, const Geom &gf, const void *kf, int n_first, int n_main,
                 hipStream_t stream) {
    typedef typename Main::S S;
    typedef typename FirstSys<Main, First>::type SF;
    PassArgs<S> am;
    PassArgs<SF> af;
    am.g = gm;
    am.k = *static_cast<const DemodK<float, S> *>(km);
    af.g = gf;
    if (kf) af.k = *static_cast<const DemodK<float, SF> *>(kf);
    else std::memcpy(&af.k, &am.k, sizeof af.k < sizeof am.k ? sizeof af.k : sizeof am.k);   // not run: n_first = 0
    // PassCfg::kUsePair: the wave pair for every instance unless the build asks for the earlier selection
    if constexpr (CM_PAIR != 0 && Main::kUsePair) {
        int floats = pair_lds_floats<Main>(am.k);
        if constexpr (!std::is_same<First, NoPass>::value) {
            const int ff = pair_lds_floats<First>(af.k);
            if (ff > floats) floats = ff;
        }
#ifdef CM_EXPERIMENTS
        if (const char *pad = std::getenv("CM_EXP_LDS_PAD_KIB")) floats += 256 * std::atoi(pad);   // fewer workgroups per CU (occupancy study)
#endif
        hipLaunchKernelGGL((demod_pair_kernel<Main, First>), dim3(n_first + n_main), dim3(128), sizeof(float) * (size_t)floats, stream, am, af, n_first);
    }
    else
        hipLaunchKernelGGL((demod_kernel<Main, First>), dim3(n_first + n_main), dim3(64), 0, stream, am, af, n_first);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("demod_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}

#ifdef CM_EXPERIMENTS
// The blocked decoder (cm_blk_kernels.h) for the main pass; the plain first-line workgroups stay on the wave-pair kernel
// (launched with an empty main pass).
template <class Main, class First, int QE, int QL>
int launch_demod_blk(const Geom &gm, const void *km, const Geom &gf, const void *kf, int n_first, int n_main, hipStream_t stream) {
    typedef typename Main::S S;
    PassArgs<S> am, af;
    am.g = gm;
    am.k = *static_cast<const DemodK<float, S> *>(km);
    af.g = gf;
    af.k = kf ? *static_cast<const DemodK<float, S> *>(kf) : am.k;
    if (n_first > 0) {
        int floats = pair_lds_floats<Main>(am.k);
        if constexpr (!std::is_same<First, NoPass>::value) {
            const int ff = pair_lds_floats<First>(af.k);
            if (ff > floats) floats = ff;
        }
        hipLaunchKernelGGL((demod_pair_kernel<Main, First>), dim3(n_first), dim3(128), sizeof(float) * (size_t)floats, stream, am, af, n_first);
    }
    if (n_main > 0)
        hipLaunchKernelGGL((demod_blk_kernel<S, QE, QL>), dim3(n_main), dim3(64), 0, stream, am, (const BlkTiles *)gm.blk_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("demod_blk_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}

// Toeplitz operand of y[t] = sum_j g[j] x[t - j] (g = the 20 odd taps of 2 h, symmetric) times kBlkScale, split into two
// float16 pieces; layout: cm_blk_fir.h: BlkTiles
inline bool build_blk_tiles(const cm_plan_desc &d, void **out) {
    std::vector<_Float16> t(64 * 16);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int kk = blk_tile_tap(l, j);
            float v = 0.f;
            if (kk >= 0) {
                const int i = kk < 10 ? kk : 19 - kk;              // tap(I) = c[I < 10 ? I : 19 - I], c[i] = 2 h[2 i + 1]
                v = (float)(2.0 * d.resample_fir[2 * i + 1]) * kBlkScale;
            }
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            t[(size_t)l * 16 + j] = hi;             // BlkTiles::hi
            t[(size_t)l * 16 + 8 + j] = lo;         // BlkTiles::lo
        }
    if (hipMalloc(out, t.size() * sizeof(_Float16)) != hipSuccess) return false;
    return hipMemcpy(*out, t.data(), t.size() * sizeof(_Float16), hipMemcpyHostToDevice) == hipSuccess;
}

#endif  // CM_EXPERIMENTS

struct Pass {
    std::vector<unsigned char> k;  // DemodK<float, S> blob
    LaneK<float> *lanes = nullptr; // device
    int cycle = 0, n_lines = 0, luma_prev_bits = 0;
    int wrap_mode = 0;             // cm_lane_table::wrap_mode (two-level comb: PassCfg::WRAP instances)
    int depth = 0;                 // halo lanes of the kernel instance
    std::string name;
};
#endif  // CM_DEMOD_PART

}  // namespace

#if CM_DEMOD_PART
typedef int (*ModLaunchFn)(const Geom &g, const void *k, int blocks, hipStream_t);

// calls up to which the decoders' scan kernels beat the streaming kernels (profiles/r03_batch_curve.txt)
#ifndef CM_SCAN_MAX_CALLS
#define CM_SCAN_MAX_CALLS 6000
#endif

struct cm_plan {
    cm_plan_desc desc;
    int device = 0;
    float *carrier4 = nullptr, *carrier2 = nullptr;             // entry 0 of the padded tables
    float *carrier4_base = nullptr, *carrier2_base = nullptr;   // the allocations
    float *frame_rot = nullptr;   // {cos, sin} per frame of the rotation cycle (long sub-carrier cycles), else null
    unsigned *simd_load = nullptr; // wave-pair kernels: live load per (XCC, CU, SIMD), kSimdLoadEntries counters (cm_kernels.h)
    void *blk_tiles = nullptr;     // blocked decoder (cm_blk_kernels.h): Toeplitz tiles of the half-band FIR, [64 lanes] BlkTiles
    int rot_cycle = 0;
    LaunchFn fn = nullptr, fn_u8 = nullptr;
    bool has_first = false;
    int seg_warm = 1 << 30;        // samples a row segment enters the stream early (segment_warmup)
    // small batches: one wavefront per scan line (cm_scan_kernels.h); null / 0 where the plan's shape does not fit it
    ScanK *scan_main = nullptr, *scan_first = nullptr;
    ScanModK *scan_mod = nullptr;  // the QAM modulator's (qam_mod_scan_kernel)
    int scan_mod_c1 = 0;
    ScanSecamModK *scan_smod = nullptr;   // the SECAM modulator's (secam_mod_scan_kernel)
    int scan_smod_c1 = 0;
    ScanSecamK *scan_sdem = nullptr;      // the SECAM decoder's (secam_demod_scan_kernel)
    int scan_sdem_c1 = 0;
    int scan_c1 = 0, scan_depth = 0;
    mutable std::atomic<int> small_batch{CM_SMALL_BATCH_AUTO};   // cm_plan_set_small_batch (the one field that changes after creation: atomic)
    bool pair = false;             // wave-pair kernel (two wavefronts per 64 calls)
    Pass main, first;
    // modulator
    ModLaunchFn mod_fn = nullptr, mod_fn_u8 = nullptr;
    std::vector<unsigned char> mod_k;
    ModLaneK<float> *mod_lanes = nullptr;
    int mod_cycle = 0, mod_n_lines = 0, mod_depth = 0, mod_shape = 0;   // mod_shape: 1 = (1 section, shift 2), 2 = (2, 4), 0 = run-time shape
    std::string mod_name, demod_error;
    // SECAM
    bool secam = false;
    SecamDemodK<float> sd_k;
    SecamBp64 sd_e64;              // band-pass + bell of the guarded bodies in float64 (cm_stages.h)
    SecamDemodLaneK<float> *sd_lanes = nullptr;
    float *fm_ref = nullptr;      // SECAM discriminator reference {cos, sin} pairs
    double *fm_ref64 = nullptr;   // the same in float64, for the float64 front end (sd_f64)
    SecamDemodK<double> sd_k64;
    bool sd_f64 = false;          // decoder shapes whose float32 margin is thin: stage A of the wave pair in float64
    bool sd_pair = false;         // float rows run on secam_demod_pair_kernel
    float *fm_dc = nullptr;       // SECAM: decimator response to the constant fc beyond 2 fc (cm_plan.h: build_fm_dc)
    int sd_cycle = 0, sd_n_lines = 0;
    SecamModK<float, double> sm_k;
    SecamModLaneK<float, double> *sm_lanes = nullptr;
};

#endif  // CM_DEMOD_PART
namespace {
#if CM_DEMOD_PART

template <class S>
bool make_pass(const cm_plan_desc &d, bool pald, bool bsf, const cm_lane_table &tb, Pass &pass, std::string &err, bool pair, int depth = 0) {
    DemodK<float, S> k;
    DemodScales sc;
    if (!build_demod_k<float, S>(d, pald, bsf, k, sc, err)) return false;
    {   // capacities the kernels assume (cm_kernels.h): carrier padding, band-stop luma ring of the wave pair
        const int lat_front = pald ? 10 + k.q_e + 9 + 10 + k.q_l + 9 : 10 + k.q_e + k.q_l + 9;
        const bool wrap = tb.wrap_mode != 0;      // PassCfg::WRAP: one more step of output latency
        const int lat_out = lat_front + 1 + k.s_p + (wrap ? 1 : 0);
        if (lat_out + 8 > kCarrierPad) { err = "pipeline latency beyond the carrier table padding"; return false; }
        const int luma_lag = lat_out - (10 + k.q_r + 9);   // steps between the band-stop luma sample and its use
        if (bsf && (pair ? luma_lag + 12 > luma_ring_slots<S>() : luma_lag > 15)) { err = "band-stop luma delay beyond its LDS ring"; return false; }
        const bool lcut = CM_QAM_LPF_IN_A != 0 && !pald && !bsf && depth >= 2 && !S::RT;     // PassCfg::kLcutCfg
        const int ring_max = pald ? luma_delay_max_latency<S, 1>()
                           : (lcut ? (wrap ? luma_delay_max_latency<S, 0, true, 1>() : luma_delay_max_latency<S, 0, true>()) : luma_delay_max_latency<S, 0>());
        if (wrap && !lcut && !S::RT) { err = "the two-level comb is built on the depth-2 QAM instances"; return false; }
        const int ring_win = pald ? ring_window<S, 1>() : (lcut ? ring_window<S, 0, true>() : ring_window<S, 0>());
        if (CM_LUMA_RING && !S::NORING && !bsf && pair && lat_out > ring_max) { err = "pipeline latency beyond the luma delay ring"; return false; }
        if (CM_LUMA_RING && !S::NORING && !bsf && pair && lat_out < 10 + ring_win) { err = "pipeline latency below the luma window"; return false; }
    }
    pass.k.resize(sizeof(k));
    std::memcpy(pass.k.data(), &k, sizeof(k));
    const size_t n = (size_t)tb.frame_cycle * 3 * tb.n_lines;
    std::vector<LaneK<float>> host(n);
    for (size_t i = 0; i < n; ++i) host[i] = convert_lane<float>(tb.table + i * CM_LANE_DOUBLES, sc);
    if (hipMalloc((void **)&pass.lanes, n * sizeof(LaneK<float>)) != hipSuccess ||
        hipMemcpy(pass.lanes, host.data(), n * sizeof(LaneK<float>), hipMemcpyHostToDevice) != hipSuccess) {
        err = "device allocation / upload of the lane table failed";
        return false;
    }
    pass.cycle = tb.frame_cycle;
    pass.n_lines = tb.n_lines;
    pass.luma_prev_bits = tb.luma_from_prev;
    pass.wrap_mode = tb.wrap_mode;
    return true;
}

template <class S, class SF = S>
bool make_passes(cm_plan *p, const cm_plan_desc &d, bool pald, bool bsf, bool first, std::string &err) {
    if (!make_pass<S>(d, pald, bsf, d.demod_main, p->main, err, p->pair, p->main.depth)) return false;
    if (first && !make_pass<SF>(d, false, true, d.demod_first, p->first, err, p->pair)) return false;
    p->has_first = first;
    return true;
}


// The blocked decoder (cm_blk_kernels.h: round 2's experiment with the FIRs on the matrix pipe) replaces the wave pair
// for the PAL-D front end of an even-shift tuned shape - in -DCM_EXPERIMENTS builds only, when CM_BLK is set in the environment at plan creation.
#ifdef CM_EXPERIMENTS
template <class S, class First>
bool maybe_select_blk(cm_plan *p, const cm_plan_desc &d) {
    if constexpr (!S::ODD_E && !S::ODD_L && !S::RT) {
        const char *env = getenv("CM_BLK");
        if (!env || !*env || *env == '0') return false;
        if (d.width % 4) return false;
        const int q_e = pair_delay(d.extract2x.shift), q_l = pair_delay(d.pald_lp.shift);
        if (q_e != 2 || q_l != 3) return false;          // the PAL-BG delays the instance is compiled for
        if (!build_blk_tiles(d, &p->blk_tiles)) return false;
        p->fn = launch_demod_blk<PassCfg<S, FRONT_PALD, false, 1, 16>, First, 2, 3>;
        return true;
    }
    return false;
}
#else
template <class S, class First>
bool maybe_select_blk(cm_plan *, const cm_plan_desc &) { return false; }
#endif

// Kernel instances of one filter-set shape S.  HAS_PALD / HAS_D1: whether the PAL-D front end and the one-line
// comb behind the QAM front end (NTSC comb) exist for this shape.  The notch variants are float-only.
template <class S, bool HAS_PALD, bool HAS_D1>
bool select_for_shape(cm_plan *p, const cm_plan_desc &d, const char *sys, std::string &err) {
    const bool pald = d.pipeline == CM_PIPE_PAL_D;
    const bool bsf = d.main_luma_bandstop != 0;
    const bool first = d.first_is_plain != 0;
    const bool notch = d.notch.n_sections != 0;
    const bool minavg = d.chroma_average == CM_AVG_MIN;
    const int depth = d.depth;
    typedef PassCfg<S, FRONT_QAM, true, 0, 8> First;
    typedef PassCfg<S, FRONT_QAM, true, 0, 16, true> FirstU8;      // byte tiles are small: no need for 8-sample tiles
    std::string what;
    p->fn = nullptr;
    p->fn_u8 = nullptr;
#ifdef CM_DEV_PALD_ONLY   /* development builds: the headline instance only (compiles in seconds) */
    if constexpr (HAS_PALD) {
        if (pald && !notch && !minavg && depth == 1 && first) {
#ifndef CM_DEV_TILE
#define CM_DEV_TILE 16
#endif
            p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, CM_DEV_TILE>, First>;
            p->main.depth = 1;
            p->pair = CM_PAIR != 0;
            p->main.name = std::string(CM_PAIR ? "demod_pair_kernel<" : "demod_kernel<") + sys + ": pal-d front, depth 1 | plain first line>";
            if (maybe_select_blk<S, First>(p, d)) p->main.name = std::string("demod_blk_kernel<") + sys + ": pal-d front, depth 1, FIRs on the matrix pipe | plain first line>";
            return make_passes<S>(p, d, pald, bsf, first, err);
        }
    }
    err = "development build: PAL-D only";
    return false;
#else
    const int wrap = d.demod_main.wrap_mode;
    if (wrap) {
        // SimpleCombModem / Simple3DCombModem around Pal3DModem as a two-level comb (cm_lane_table::wrap_mode): Pal3DModem's tables, the
        // wrapper's average of consecutive calls in stage B, three halo lanes
        if constexpr (HAS_PALD) {
            if (pald || bsf || first || depth != 3 || d.skip_calls || (wrap != 1 && wrap != 2)) {
                err = "a two-level comb (wrap_mode) takes the QAM pipeline, depth 3 (two table lines + the wrapper's), no plain first line";
                return false;
            }
            if (minavg) {
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, true, true, true>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, true, true, true>, NoPass>;
            } else if (notch) {
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, true, false, true>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, true, false, true>, NoPass>;
            } else {
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, false, false, true>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, false, false, true>, NoPass>;
            }
            p->main.depth = 3;
            what = std::string("qam front, depth 2") + (minavg ? ", minavg" : "") + (wrap == 2 ? " | minavg" : " | avg") + " of consecutive calls (two-level comb)";
        } else {
            err = std::string("no two-level comb instance for the ") + sys + " filter shapes";
            return false;
        }
    } else if (pald && depth == 2 && !first) {
        // SimpleCombModem / Simple3DCombModem around PalDModem, the calls k >= 2 of every run (comb.py:96-113 over pal.py:79-127: both
        // chroma estimates come from the PAL-D front end there, two lines of history; cm_comb_wrap_demodulate_frames_fused supplies
        // the calls k < 2, which mix in the plain first-line decode)
        if constexpr (HAS_PALD) {
            if (d.skip_calls != 2) { err = "the PAL-D front end with two lines of history serves the fused wrapped combs (skip_calls = 2)"; return false; }
            if (minavg) {
                p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, false, true, true>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, true, true, true>, NoPass>;
            } else if (notch) {
                p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, false, true>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, true, true>, NoPass>;
            } else {
                p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, true>, NoPass>;
            }
            p->main.depth = 2; what = minavg ? "pal-d front, depth 2, minavg (wrapped comb, calls k >= 2)" : "pal-d front, depth 2 (wrapped comb, calls k >= 2)";
        } else {
            err = std::string("no PAL-D front end for the ") + sys + " filter shapes";
            return false;
        }
    } else if (minavg) {
        // comb.py:13-15 behind SimpleCombModem / Pal3DModem: one instance per shape (depth 2, notch switchable)
        if (pald || bsf || first) { err = "minavg is built behind the QAM front end (SimpleCombModem, Pal3DModem)"; return false; }
        p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, true, true>, NoPass>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, true, true>, NoPass>;
        p->main.depth = 2; what = "qam front, depth 2, minavg";
    } else if (pald) {
        if constexpr (HAS_PALD) {
            if (depth != 1 || !first) { err = "PAL-D front end is built with one line of history and a plain first line"; return false; }
            if (notch) {
                p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, false, true>, First>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true, true>, FirstU8>;
            } else {
                p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16>, First>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true>, FirstU8>;
            }
            p->main.depth = 1; what = "pal-d front, depth 1 | plain first line";
            if (!notch && maybe_select_blk<S, First>(p, d)) what = "pal-d front, depth 1, FIRs on the matrix pipe (demod_blk_kernel) | plain first line";
        } else {
            err = std::string("no PAL-D front end for the ") + sys + " filter shapes";
            return false;
        }
    } else if (bsf) {
        if (depth != 0 || first || notch) { err = "band-stop luma is built for plain decoders only"; return false; }
        p->fn = launch_demod<PassCfg<S, FRONT_QAM, true, 0, 16>, NoPass>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, true, 0, 16, true>, NoPass>;
        p->main.depth = 0; what = "qam front + band-stop, depth 0";
    } else if (first) {
        if constexpr (HAS_D1) {
            if (depth != 1) { err = "a comb with a plain first line is built with one line of history"; return false; }
            if (notch) {
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, false, true>, First>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, true, true>, FirstU8>;
            } else {
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16>, First>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, true>, FirstU8>;
            }
            p->main.depth = 1; what = "qam front, depth 1 | plain first line";
        } else {
            err = std::string("no kernel instance with a plain first line behind the QAM front end for the ") + sys + " filter shapes";
            return false;
        }
    } else {
        if (notch) {
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, true>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, true>, NoPass>;
        } else {
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true>, NoPass>;
        }
        p->main.depth = 2; what = "qam front, depth 2";
    }
#ifdef CM_ONE_WAVE_SELECT
    const bool pair = CM_PAIR != 0 && ((p->main.depth < 2 && !notch && !minavg && S::NE < 4 && S::NP < 2) || (pald && notch));   // PassCfg::kUsePair
#else
    const bool pair = CM_PAIR != 0;   // PassCfg::kUsePair
#endif
    p->pair = pair;
    p->main.name = std::string(pair ? "demod_pair_kernel<" : "demod_kernel<") + sys + ": " + what + (notch ? " + notch>" : ">");
    return make_passes<S>(p, d, pald, bsf, first, err);
#endif
}

#endif  // CM_DEMOD_PART
#if CM_SHAPES_PART
// Run-time shape (SysAny): any sampling rate whose filters fit 4 / 3 / 3 / 2 sections and a pre-correction shift <= 12.
// The fused byte boundary exists where the tuned shapes have it (not with notch / minavg).
bool select_any(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    typedef SysAny S;
    const bool pald = d.pipeline == CM_PIPE_PAL_D;
    const bool bsf = d.main_luma_bandstop != 0;
    const bool first = d.first_is_plain != 0;
    const bool notch = d.notch.n_sections != 0;
    const bool minavg = d.chroma_average == CM_AVG_MIN;
    const int depth = d.depth;
    typedef PassCfg<S, FRONT_QAM, true, 0, 8> First;
    typedef PassCfg<S, FRONT_QAM, true, 0, 16, true> FirstU8;
    p->fn = nullptr;
    p->fn_u8 = nullptr;
    std::string what;
    if (d.demod_main.wrap_mode) {
        // the two-level comb around Pal3DModem (select_for_shape) at the other sampling rates: comb.avg / comb.minavg of the wrapper over Pal3DModem's
        // plain average - its own minavg and the notch stay on the composition there, like the fused plans around PalDModem
        if (pald || bsf || first || depth != 3 || d.skip_calls) { err = "a two-level comb (wrap_mode) takes the QAM pipeline, depth 3, no plain first line"; return false; }
        if (minavg || notch) { err = "the run-time shape runs the two-level comb without the inner minavg / the notch (those: the composition)"; return false; }
        p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, false, false, true>, NoPass>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, false, false, true>, NoPass>;
        p->main.depth = 3; what = std::string("qam front, depth 2 | ") + (d.demod_main.wrap_mode == 2 ? "minavg" : "avg") + " of consecutive calls (two-level comb)";
    } else if (pald && depth == 2 && !first) {
        // the fused wrapped combs (select_for_shape) at the other sampling rates: the comb.avg form only - minavg / notch stay on the composition
        if (d.skip_calls != 2) { err = "the PAL-D front end with two lines of history serves the fused wrapped combs (skip_calls = 2)"; return false; }
        if (minavg || notch) { err = "the run-time shape fuses the plain average only (minavg / notch: the composition)"; return false; }
        p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16>, NoPass>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, true>, NoPass>;
        p->main.depth = 2; what = "pal-d front, depth 2 (wrapped comb, calls k >= 2)";
    } else if (minavg) {
        if (pald || bsf || first) { err = "minavg is built behind the QAM front end (SimpleCombModem, Pal3DModem)"; return false; }
        p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, true, true>, NoPass>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, true, true>, NoPass>;
        p->main.depth = 2; what = "qam front, depth 2, minavg";
    } else if (pald) {
        if (depth != 1 || !first) { err = "PAL-D front end is built with one line of history and a plain first line"; return false; }
        if (notch) {
            p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, false, true>, First>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true, true>, FirstU8>;
        } else {
            p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16>, First>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true>, FirstU8>;
        }
        p->main.depth = 1; what = "pal-d front, depth 1 | plain first line";
    } else if (bsf) {
        if (depth != 0 || first || notch) { err = "band-stop luma is built for plain decoders only"; return false; }
        p->fn = launch_demod<PassCfg<S, FRONT_QAM, true, 0, 16>, NoPass>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, true, 0, 16, true>, NoPass>;
        p->main.depth = 0; what = "qam front + band-stop, depth 0";
    } else if (first) {
        if (depth != 1) { err = "a comb with a plain first line is built with one line of history"; return false; }
        if (notch) {
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, false, true>, First>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, true, true>, FirstU8>;
        } else {
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16>, First>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, true>, FirstU8>;
        }
        p->main.depth = 1; what = "qam front, depth 1 | plain first line";
    } else {
        if (notch) {
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, true>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, true>, NoPass>;
        } else {
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true>, NoPass>;
        }
        p->main.depth = 2; what = "qam front, depth 2";
    }
    p->pair = CM_PAIR != 0;   // PassCfg::kUsePair: the run-time shape does not fit one wave's registers
    p->main.name = std::string(p->pair ? "demod_pair_kernel" : "demod_kernel") + "<run-time shape: " + what + (notch ? " + notch>" : ">");
    return make_passes<S>(p, d, pald, bsf, first, err);
}

// PalDModem on the 768-sample PAL raster (SysPalSq | SysPalSqFirst): the headline decoder's instances for the square-pixel
// image size (round 3; on the run-time shape it ran at 113 Gpixel/s against 171 at 720 wide)
bool select_pald_sq(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    typedef SysPalSq S;
    typedef PassCfg<SysPalSqFirst, FRONT_QAM, true, 0, 8> First;
    typedef PassCfg<SysPalSqFirst, FRONT_QAM, true, 0, 16, true> FirstU8;
    const bool notch = d.notch.n_sections != 0;
    p->fn = nullptr;
    p->fn_u8 = nullptr;
    if (notch) {
        p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, false, true>, First>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true, true>, FirstU8>;
    } else {
        p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16>, First>;
        p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true>, FirstU8>;
    }
    p->main.depth = 1;
    p->pair = CM_PAIR != 0;
    p->main.name = std::string("demod_pair_kernel<pal at 768 samples per line: pal-d front, depth 1 | plain first line") + (notch ? " + notch>" : ">");
    return make_passes<S, SysPalSqFirst>(p, d, true, false, true, err);
}

}  // namespace
namespace cm_host {
bool select_other_shapes(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    const bool pald = d.pipeline == CM_PIPE_PAL_D;
    const bool bsf = d.main_luma_bandstop != 0;
    const bool first = d.first_is_plain != 0;
    SysSignature want = signature_wanted(d, pald);
    const SysSignature want_first = signature_wanted(d, false);   // the plain first-line pass runs the QAM front + band-stop
    auto match = [&](SysSignature have) {
        if (first && !same_signature(want_first, have)) return false;   // one launch, one shape for both passes
        if (!bsf && !first) { have.nr = want.nr; have.odd_r = want.odd_r; }
        return same_signature(want, have);
    };
#ifndef CM_DEV_PALD_ONLY
    if (pald && first && d.depth == 1 && d.chroma_average != CM_AVG_MIN && same_signature(want, signature_of<SysPalSq>()) &&
        same_signature(want_first, signature_of<SysPalSqFirst>()))
        return select_pald_sq(p, d, err);
    if (match(signature_of<SysNtsc>())) return select_for_shape<SysNtsc, true, true>(p, d, "ntsc (pal-m/n)", err);
    if (!pald && match(signature_of<SysNtscI>())) return select_for_shape<SysNtscI, false, true>(p, d, "ntsc-i", err);
    if (!pald && match(signature_of<SysNtscSq>())) return select_for_shape<SysNtscSq, false, true>(p, d, "ntsc at 640 / 704 samples per line", err);
    if (!pald && match(signature_of<SysNtscA>())) return select_for_shape<SysNtscA, false, true>(p, d, "ntsc-a", err);
    {   // the tuned shapes of the wide rasters (CM_PART 5 .. 7); -1: none of them serves this plan
        int r = select_wide_pald(p, d, err);
        if (r < 0) r = select_wide_pal_qam(p, d, err);
        if (r < 0) r = select_wide_ntsc(p, d, err);
        if (r >= 0) return r == 1;
    }
#endif
    if (fits_any(want) && (!first || fits_any(want_first))) return select_any(p, d, err);
    char buf[256];
    snprintf(buf, sizeof buf,
             "no kernel instance for this filter set (sections extract/remove/detect/pre = %d/%d/%d/%d, shift parities %d/%d/%d, "
             "pre shift %d); built: the filter shapes of PAL-BG, NTSC-M (= PAL-M/N, NTSC-N/3.61), NTSC-I/4.43 and NTSC-A at 13.5 MHz",
             want.ne, want.nr, want.nl, want.np, want.odd_e, want.odd_l, want.odd_r, want.sp);
    err = buf;
    return false;
}
}  // namespace cm_host
namespace {
#endif  // CM_SHAPES_PART
#if CM_WIDE_PART
// ---- the tuned shapes of the wide rasters (round 6; cm_shapes_wide.h, written by tools/gen_wide_shapes.py) -----------------------
// Every image width has its own sampling rate and with it its own filter orders and FilterFunction shift parities (ref line.py:49-55,
// utils.py:44-64).  Until round 6 only 640 / 704 / 720 / 768 samples per line had kernel instances with these as compile-time constants and
// every other width ran on the run-time shape (SysAny: padded sections, run-time parities, 41 KiB of LDS, 2 waves per SIMD: 65 - 80 % of
// the tuned speed).  The instances here cover the plain stacks of the common wide rasters - PalDModem, Pal3DModem / the two-line combs,
// PalSModem, NtscModem, NtscCombModem, Simple3DCombModem(NtscCombModem) and the fused comb wrappers around PalDModem / Pal3DModem - floats
// and bytes; notch / minavg stay on the run-time shape there.
enum WideKind { WIDE_PALD, WIDE_PAL_QAM, WIDE_NTSC };
template <class S, class SF, WideKind KIND>
int select_wide(cm_plan *p, const cm_plan_desc &d, const char *sys, std::string &err) {
    const bool pald = d.pipeline == CM_PIPE_PAL_D;
    const bool bsf = d.main_luma_bandstop != 0;
    const bool first = d.first_is_plain != 0;
    const int depth = d.depth, wrap = d.demod_main.wrap_mode;
    if (d.notch.n_sections != 0 || d.chroma_average == CM_AVG_MIN) return -1;
    typedef PassCfg<SF, FRONT_QAM, true, 0, 8> First;
    typedef PassCfg<SF, FRONT_QAM, true, 0, 16, true> FirstU8;
    std::string what;
    p->fn = nullptr;
    p->fn_u8 = nullptr;
    if constexpr (KIND == WIDE_PALD) {
        if (!pald || wrap) return -1;
        if (depth == 1 && first && !d.skip_calls) {
            p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16>, First>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 1, 16, true>, FirstU8>;
            p->main.depth = 1; what = "pal-d front, depth 1 | plain first line";
        } else if (depth == 2 && !first && d.skip_calls == 2) {
            p->fn = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_PALD, false, 2, 16, true>, NoPass>;
            p->main.depth = 2; what = "pal-d front, depth 2 (wrapped comb, calls k >= 2)";
        } else return -1;
    } else {
        if (pald || d.skip_calls) return -1;
        if (wrap) {
            if constexpr (KIND == WIDE_PAL_QAM) {
                if (bsf || first || depth != 3 || (wrap != 1 && wrap != 2)) return -1;
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, false, false, false, true>, NoPass>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true, false, false, true>, NoPass>;
                p->main.depth = 3; what = std::string("qam front, depth 2 | ") + (wrap == 2 ? "minavg" : "avg") + " of consecutive calls (two-level comb)";
            } else return -1;
        } else if (bsf) {
            if (depth != 0 || first) return -1;
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, true, 0, 16>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, true, 0, 16, true>, NoPass>;
            p->main.depth = 0; what = "qam front + band-stop, depth 0";
        } else if (first) {
            if constexpr (KIND == WIDE_NTSC) {
                if (depth != 1) return -1;
                p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16>, First>;
                p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 1, 16, true>, FirstU8>;
                p->main.depth = 1; what = "qam front, depth 1 | plain first line";
            } else return -1;
        } else {
            if (depth > 2) return -1;
            p->fn = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16>, NoPass>;
            p->fn_u8 = launch_demod<PassCfg<S, FRONT_QAM, false, 2, 16, true>, NoPass>;
            p->main.depth = 2; what = "qam front, depth 2";
        }
    }
    p->pair = true;
    p->main.name = std::string("demod_pair_kernel<") + sys + ": " + what + ">";
    return make_passes<S, SF>(p, d, pald, bsf, first, err) ? 1 : 0;
}
// does the plan ask for exactly this shape?  (the passes that ignore the band-stop - no band-stop luma, no plain first line - match any)
inline bool wide_match(const cm_plan_desc &d, bool pald, SysSignature have) {
    const SysSignature want = signature_wanted(d, pald);
    if (!d.main_luma_bandstop && !d.first_is_plain) { have.nr = want.nr; have.odd_r = want.odd_r; }
    return same_signature(want, have);
}
}  // namespace
namespace cm_host {
#if CM_WIDE_PALD_PART
int select_wide_pald(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    if (d.pipeline != CM_PIPE_PAL_D) return -1;
    const SysSignature want_first = signature_wanted(d, false);
#define CM_X(S, SF, LABEL) \
    if (wide_match(d, true, signature_of<S>()) && (!d.first_is_plain || same_signature(want_first, signature_of<SF>()))) \
        return select_wide<S, SF, WIDE_PALD>(p, d, LABEL, err);
    CM_WIDE_PALD_SHAPES(CM_X)
#undef CM_X
    return -1;
}
#endif
#if CM_WIDE_PAL_QAM_PART
int select_wide_pal_qam(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    if (d.pipeline == CM_PIPE_PAL_D) return -1;
#define CM_X(S, LABEL) \
    if (wide_match(d, false, signature_of<S>())) return select_wide<S, S, WIDE_PAL_QAM>(p, d, LABEL, err);
    CM_WIDE_PAL_QAM_SHAPES(CM_X)
#undef CM_X
    return -1;
}
#endif
#if CM_WIDE_NTSC_PART
int select_wide_ntsc(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    if (d.pipeline == CM_PIPE_PAL_D) return -1;
#define CM_X(S, LABEL) \
    if (wide_match(d, false, signature_of<S>())) return select_wide<S, S, WIDE_NTSC>(p, d, LABEL, err);
    CM_WIDE_NTSC_SHAPES(CM_X)
#undef CM_X
    return -1;
}
#endif
}  // namespace cm_host
namespace {
#endif  // CM_WIDE_PART
#if CM_MAIN_PART
// Pick the kernel instance (main pass + optional plain first-line pass in one launch): the PAL-BG shapes here, every other shape in CM_PART 4.
bool select_kernels(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    const bool pald = d.pipeline == CM_PIPE_PAL_D;
    const bool bsf = d.main_luma_bandstop != 0;
    const bool first = d.first_is_plain != 0;
    SysSignature want = signature_wanted(d, pald);
    const SysSignature want_first = signature_wanted(d, false);   // the plain first-line pass runs the QAM front + band-stop
    SysSignature have = signature_of<SysPal>();
    if (!bsf && !first) { have.nr = want.nr; have.odd_r = want.odd_r; }
    if ((!first || same_signature(want_first, signature_of<SysPal>())) && same_signature(want, have))
        return select_for_shape<SysPal, true, false>(p, d, "pal", err);
    return cm_host::select_other_shapes(p, d, err);
}

template <int NP, int SP, int DEPTH, bool U8 = false, bool RT = false>
int launch_qam_mod(const Geom &g, const void *kv, int blocks, hipStream_t stream) {
    ModArgs<NP> a;
    a.g = g;
    a.k = *static_cast<const ModK<float, NP> *>(kv);
    hipLaunchKernelGGL((qam_mod_kernel<NP, SP, DEPTH, U8, RT>), dim3(blocks), dim3(64), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("qam_mod_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}

// Modulator of the QAM systems (PAL / NTSC); absent tables leave the plan demodulate-only.
bool select_modulator(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    const cm_lane_table &tb = d.mod_main;
    if (!tb.table) return true;
    const bool shape1 = d.precorrect.n_sections == 1 && d.precorrect.shift == 2;   // every system but NTSC-A at 13.5 MHz
    const bool shape2 = d.precorrect.n_sections == 2 && d.precorrect.shift == 4;   // NTSC-A
    const bool shape_any = !shape1 && !shape2 && d.precorrect.n_sections <= 2 && d.precorrect.shift >= 0 &&
                           d.precorrect.shift <= kModAnyShift;                     // run-time shape: other sampling rates
    if (!shape1 && !shape2 && !shape_any) {
        err = "no modulator instance for this pre-correction filter (built: up to two sections, shift <= 12)";
        return false;
    }
    double g_pre;
    if (shape_any) {
        ModK<float, 2> k;
        k.width = d.width;
        k.s_p = d.precorrect.shift;
        if (!convert_sos<float, 2>(d.precorrect, FORM_GEN, k.pre, g_pre, err, "precorrect", true)) return false;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) k.e[i][j] = (float)d.encode_matrix[3 * i + j];
        p->mod_k.resize(sizeof k);
        std::memcpy(p->mod_k.data(), &k, sizeof k);
    } else if (shape1) {
        ModK<float, 1> k;
        k.width = d.width;
        k.s_p = d.precorrect.shift;
        if (!convert_sos<float, 1>(d.precorrect, FORM_GEN, k.pre, g_pre, err, "precorrect")) return false;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) k.e[i][j] = (float)d.encode_matrix[3 * i + j];
        p->mod_k.resize(sizeof k);
        std::memcpy(p->mod_k.data(), &k, sizeof k);
    } else {
        ModK<float, 2> k;
        k.width = d.width;
        k.s_p = d.precorrect.shift;
        if (!convert_sos<float, 2>(d.precorrect, FORM_GEN, k.pre, g_pre, err, "precorrect")) return false;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) k.e[i][j] = (float)d.encode_matrix[3 * i + j];
        p->mod_k.resize(sizeof k);
        std::memcpy(p->mod_k.data(), &k, sizeof k);
    }
    const size_t n = (size_t)tb.frame_cycle * 3 * tb.n_lines;
    std::vector<ModLaneK<float>> host(n);
    for (size_t i = 0; i < n; ++i) {
        const double *e = tb.table + i * CM_LANE_DOUBLES;
        ModLaneK<float> &l = host[i];
        l.sph = (float)(e[0] * g_pre);
        l.cph = (float)(e[1] * g_pre);
        l.vsph = (float)(e[0] * g_pre * e[6]);
        l.vcph = (float)(e[1] * g_pre * e[6]);
        l.wy0 = (float)e[2]; l.wy1 = (float)e[3]; l.wc0 = (float)e[4]; l.wc1 = (float)e[5];
    }
    if (hipMalloc((void **)&p->mod_lanes, n * sizeof(ModLaneK<float>)) != hipSuccess ||
        hipMemcpy(p->mod_lanes, host.data(), n * sizeof(ModLaneK<float>), hipMemcpyHostToDevice) != hipSuccess) {
        err = "device allocation / upload of the modulator table failed";
        return false;
    }
    p->mod_cycle = tb.frame_cycle;
    p->mod_n_lines = tb.n_lines;
    p->mod_depth = d.modulation_delay ? 1 : 0;
    p->mod_shape = shape_any ? 0 : (shape1 ? 1 : 2);
    if (shape_any) {
        p->mod_fn = p->mod_depth ? launch_qam_mod<2, kModAnyShift, 1, false, true> : launch_qam_mod<2, kModAnyShift, 0, false, true>;
        p->mod_fn_u8 = p->mod_depth ? launch_qam_mod<2, kModAnyShift, 1, true, true> : launch_qam_mod<2, kModAnyShift, 0, true, true>;
    } else if (shape1) {
        p->mod_fn = p->mod_depth ? launch_qam_mod<1, 2, 1> : launch_qam_mod<1, 2, 0>;
        p->mod_fn_u8 = p->mod_depth ? launch_qam_mod<1, 2, 1, true> : launch_qam_mod<1, 2, 0, true>;
    } else {
        p->mod_fn = p->mod_depth ? launch_qam_mod<2, 4, 1> : launch_qam_mod<2, 4, 0>;
        p->mod_fn_u8 = p->mod_depth ? launch_qam_mod<2, 4, 1, true> : launch_qam_mod<2, 4, 0, true>;
    }
    p->mod_name = std::string(p->mod_depth ? "qam_mod_kernel<line averaging" : "qam_mod_kernel<") + (shape_any ? ", run-time shape>" : ">");
    return true;
}

template <class LaneT, class Conv>
bool upload_lanes(const cm_lane_table &tb, LaneT **dev, Conv conv, std::string &err) {
    const size_t n = (size_t)tb.frame_cycle * 3 * tb.n_lines;
    std::vector<LaneT> host(n);
    for (size_t i = 0; i < n; ++i) host[i] = conv(tb.table + i * CM_LANE_DOUBLES);
    if (hipMalloc((void **)dev, n * sizeof(LaneT)) != hipSuccess ||
        hipMemcpy(*dev, host.data(), n * sizeof(LaneT), hipMemcpyHostToDevice) != hipSuccess) {
        err = "device allocation / upload of a lane table failed";
        return false;
    }
    return true;
}

void make_scan_secam_mod(cm_plan *p, const cm_plan_desc &d);      // small batches: secam_mod_scan_kernel (below)
void make_scan_secam_demod(cm_plan *p, const cm_plan_desc &d);    // ... secam_demod_scan_kernel
bool create_secam(cm_plan *p, const cm_plan_desc &d, std::string &err) {
    p->secam = true;
    if (!build_secam_demod_k<float>(d, p->sd_k, err)) return false;
    if (!build_secam_bp64(d, p->sd_e64, err)) return false;
    if (!d.demod_main.table) { err = "demod_main table missing"; return false; }
    if (!upload_lanes(d.demod_main, &p->sd_lanes, [&](const double *e) { return convert_secam_demod_lane<float>(e, d.secam); }, err))
        return false;
    p->sd_cycle = d.demod_main.frame_cycle;
    p->sd_n_lines = d.demod_main.n_lines;
    std::vector<float> fm = build_fm_reference<float>(d.secam.fm_fc, d.width + d.secam.preroll);
    std::vector<float> dc = build_fm_dc<float>(d, d.width + d.secam.preroll);
    if (hipMalloc((void **)&p->fm_ref, fm.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->fm_ref, fm.data(), fm.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void **)&p->fm_dc, dc.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->fm_dc, dc.data(), dc.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        err = "device allocation / upload of the FM reference failed";
        return false;
    }
    {   // Where float32 is too thin for 1e-5.  With the band-pass + bell of the row ends in float64 (SecamBp64)
        // the row-end transients are gone and what is left of the float32 error is (a) uniform rounding noise of the front
        // end, which the discriminator divides by the deviation - it grows like (fs / fdev)^1.5: tests/sim over 7 variants x
        // 10 widths x 16 seeds (profiles/r03_secam_sim_sweep.txt) gives 3e-6 at 1920 wide with de-emphasis, without 4.2e-6
        // at 1280 (2 / fdev = 96), 5.5e-6 at 1440, 9.2e-6 at 1920 - and (b) isolated samples where the sub-carrier's
        // envelope dips (sharp colour transitions; variants III / M / N): the angle of a small (I, Q) multiplies that noise by
        // typical / momentary amplitude - 4 - 5 x the median error in 1 of 40 random frames at 720 wide, and 2.5e-4 in one
        // SECAM-N frame at 1920 wide (profiles/r03_fuzz_summary.txt) where the float64 front end gives 2e-6.  So: float64 from
        // 2 / fdev > 100 on (1280 wide and more), as in rounds 1 - 2; the variants without de-emphasis no longer need it
        // below that (their misses were row-end transients).  lane entry e[1] = fdev / (fs / 2).
        double fdev_min = 1e9;
        const size_t n_lanes = (size_t)d.demod_main.frame_cycle * 3 * d.demod_main.n_lines;
        for (size_t i = 0; i < n_lanes; ++i) {
            const double fd = d.demod_main.table[i * CM_LANE_DOUBLES + 1];
            if (fd > 0.0 && fd < fdev_min) fdev_min = fd;
        }
        const int d_luma = p->sd_k.s_b + 20 + p->sd_k.q_l - p->sd_k.s_y;
        const bool thin = 2.0 / fdev_min > 100.0;
        // the caller may ask for the float64 front end whatever the shape (cm_secam_desc.present & CM_SECAM_FLOAT64)
        const bool want64 = thin || (d.secam.present & CM_SECAM_FLOAT64) != 0;
        p->sd_f64 = CM_SECAM_F64 && want64 && d_luma >= 4 + 4 * CM_SECAM_PAIR_REG_DELAY && d_luma <= kSecamPairMaxLumaDelay;
        if (p->sd_f64) {
            if (!build_secam_demod_k<double>(d, p->sd_k64, err)) return false;
            std::vector<double> fm64 = build_fm_reference<double>(d.secam.fm_fc, d.width + d.secam.preroll);
            if (hipMalloc((void **)&p->fm_ref64, fm64.size() * sizeof(double)) != hipSuccess ||
                hipMemcpy(p->fm_ref64, fm64.data(), fm64.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
                err = "device allocation / upload of the float64 FM reference failed";
                return false;
            }
        }
    }
    if (d.mod_main.table) {
        if (!build_secam_mod_k<float, double>(d, p->sm_k, err)) return false;
        if (p->sm_k.s_p < 0 || p->sm_k.s_p > kModAnyShift) { err = "SECAM encoder: pre-correction shift beyond the luma delay window (12)"; return false; }
        if (!upload_lanes(d.mod_main, &p->sm_lanes, convert_secam_mod_lane<float, double>, err)) return false;
        p->mod_cycle = d.mod_main.frame_cycle;
        p->mod_n_lines = d.mod_main.n_lines;
        p->mod_depth = d.modulation_delay ? 1 : 0;
        make_scan_secam_mod(p, d);
    }
    p->main.depth = 1;
    {
        const int d_luma = p->sd_k.s_b + 20 + p->sd_k.q_l - p->sd_k.s_y;
        const bool ring_ok = d_luma >= 4 + 4 * CM_SECAM_PAIR_REG_DELAY && d_luma <= kSecamPairMaxLumaDelay;
        p->sd_pair = CM_SECAM_PAIR && ring_ok;
        p->main.name = p->sd_f64 ? "secam_demod_pair64_kernel (stage A in float64)"
                     : p->sd_pair ? "secam_demod_pair_kernel" : "secam_demod_kernel";
    }
    make_scan_secam_demod(p, d);
    return true;
}

int scan_secam_demod(const cm_plan *p, const Geom &g, hipStream_t stream, bool u8);    // small batches: secam_demod_scan_kernel (below)
int run_secam_demod(const cm_plan *p, Geom g, hipStream_t stream, bool u8 = false) {
    g.lanes = reinterpret_cast<const LaneK<float> *>(p->sd_lanes);
    g.carrier4 = p->fm_ref;
    g.carrier2 = p->fm_dc;
    g.cycle = p->sd_cycle;
    g.n_lines = p->sd_n_lines;
    g.skip_first = 0;
    long long blocks = (g.total_calls + 62) / 63;
    if (blocks <= 0) return CM_OK;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    if (p->scan_sdem && (p->small_batch == CM_SMALL_BATCH_SCAN || (p->small_batch == CM_SMALL_BATCH_AUTO && g.total_calls <= 9000)))       // (no row segments on this path: the hand-over comes later)
        return scan_secam_demod(p, g, stream, u8);
    if (p->small_batch == CM_SMALL_BATCH_SCAN) return fail(CM_ERR_UNSUPPORTED, "the scan kernel does not serve this plan");
    SecamDemodArgs a;
    a.g = g;
    a.k = p->sd_k;
    a.e64 = p->sd_e64;
    // the wave pair with the luma delay ring where the delay fits the ring (cm_secam_kernels.h), else one wave per 64 calls
    const int d_luma = p->sd_k.s_b + 20 + p->sd_k.q_l - p->sd_k.s_y;
    if (p->sd_f64) {
        SecamDemodArgs64 a64;
        a64.a = a;
        a64.k64 = p->sd_k64;
        a64.fm_ref64 = p->fm_ref64;
        if (u8) hipLaunchKernelGGL(secam_demod_pair64_kernel<true>, dim3((int)blocks), dim3(128), sizeof(float) * secam_pair_lds_floats<true>(d_luma), stream, a64);
        else hipLaunchKernelGGL(secam_demod_pair64_kernel<false>, dim3((int)blocks), dim3(128), sizeof(float) * secam_pair_lds_floats<false>(d_luma), stream, a64);
    } else if (p->sd_pair && (!u8 || CM_SECAM_PAIR_U8)) {
        if (u8) hipLaunchKernelGGL(secam_demod_pair_kernel<true>, dim3((int)blocks), dim3(128), sizeof(float) * secam_pair_lds_floats<true>(d_luma), stream, a);
        else hipLaunchKernelGGL(secam_demod_pair_kernel<false>, dim3((int)blocks), dim3(128), sizeof(float) * secam_pair_lds_floats<false>(d_luma), stream, a);
    } else if (u8) hipLaunchKernelGGL(secam_demod_kernel<true>, dim3((int)blocks), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(secam_demod_kernel<false>, dim3((int)blocks), dim3(64), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("secam_demod_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}

int scan_secam_mod(const cm_plan *p, const Geom &g, hipStream_t stream, bool u8);      // small batches: secam_mod_scan_kernel (below)
int run_secam_mod(const cm_plan *p, Geom g, hipStream_t stream, bool u8 = false) {
    if (!p->sm_lanes) return fail(CM_ERR_UNSUPPORTED, "this plan has no modulator");
    g.lanes = reinterpret_cast<const LaneK<float> *>(p->sm_lanes);
    g.cycle = p->mod_cycle;
    g.n_lines = p->mod_n_lines;
    long long blocks = (g.total_calls + (64 - p->mod_depth) - 1) / (64 - p->mod_depth);
    if (blocks <= 0) return CM_OK;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    if (p->scan_smod && (p->small_batch == CM_SMALL_BATCH_SCAN || (p->small_batch == CM_SMALL_BATCH_AUTO && g.total_calls <= 40000)))
        return scan_secam_mod(p, g, stream, u8);
    SecamModArgs a;
    a.g = g;
    a.k = p->sm_k;
    const bool any = p->sm_k.s_p != 3;   // 13.5 MHz: shift 3 (tuned instance); other sampling rates: run-time window
    if (any) {
        if (p->mod_depth) {
            if (u8) hipLaunchKernelGGL((secam_mod_kernel<kModAnyShift, 1, true, true>), dim3((int)blocks), dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((secam_mod_kernel<kModAnyShift, 1, false, true>), dim3((int)blocks), dim3(64), 0, stream, a);
        } else {
            if (u8) hipLaunchKernelGGL((secam_mod_kernel<kModAnyShift, 0, true, true>), dim3((int)blocks), dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((secam_mod_kernel<kModAnyShift, 0, false, true>), dim3((int)blocks), dim3(64), 0, stream, a);
        }
    } else if (p->mod_depth) {
        if (u8) hipLaunchKernelGGL((secam_mod_kernel<3, 1, true>), dim3((int)blocks), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((secam_mod_kernel<3, 1>), dim3((int)blocks), dim3(64), 0, stream, a);
    } else {
        if (u8) hipLaunchKernelGGL((secam_mod_kernel<3, 0, true>), dim3((int)blocks), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((secam_mod_kernel<3, 0>), dim3((int)blocks), dim3(64), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("secam_mod_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}

// frame numbering of a launch: table row of the first frame and, for rotating plans, its place in the rotation cycle
void set_first_frame(const cm_plan *p, Geom &g, int64_t first_frame, int table_cycle) {
    g.first_frame = (int)(first_frame % (int64_t)table_cycle);
    g.frame_rot = p->frame_rot;
    g.rot_cycle = p->rot_cycle;
    g.rot_first = p->frame_rot ? (int)(first_frame % (int64_t)p->rot_cycle) : 0;
}

void finish_geom(const cm_plan *p, const Pass &pass, Geom &g) {
    g.lanes = pass.lanes;
    g.carrier4 = p->carrier4;
    g.carrier2 = p->carrier2;
    g.cycle = pass.cycle;
    g.n_lines = pass.n_lines;
    g.luma_prev_bits = pass.luma_prev_bits;
    g.wrap_mode = pass.wrap_mode;
}

// ---- small batches: rows cut into segments (cm_kernels.h: Geom::seg_len) ---------------------------------------------------
// A lane walks its row sample by sample, so one launch lasts as long as ONE row takes (0.2 ms for 720 samples) however few
// rows there are: a single frame fills 10 of 256 CUs for 0.2 ms, the per-row protocol one lane of one CU.  With few
// workgroups the row is cut into segments and every workgroup walks one segment of its 64 calls, entering the stream
// `warm` samples early from a zero state.  The recursive filters forget that state geometrically (slowest pole of the 2x-rate
// filters r: r^2 per sample); warm is where the memory has decayed to 1e-8 (98 samples for PAL-BG, + the FIR windows).
// Output differs from the unsegmented walk by < 1e-7 of full scale (tests: test_small_batches_run_in_row_segments).
static double slowest_pole(const cm_iir_desc &d) {
    double r = 0.0;
    for (int j = 0; j < d.n_sections && j < CM_MAX_SECTIONS; ++j) {
        const double a1 = d.sos[j][4], a2 = d.sos[j][5], disc = a1 * a1 - 4.0 * a2;
        const double rj = disc < 0.0 ? std::sqrt(a2) : std::fmax(std::fabs((-a1 + std::sqrt(disc)) * 0.5), std::fabs((-a1 - std::sqrt(disc)) * 0.5));
        r = std::fmax(r, rj);
    }
    return r;
}
static int segment_warmup(const cm_plan_desc &d) {
    const double eps = 1e-8;
    double n = 0.0;      // samples of the 1x rate
    const cm_iir_desc *two_x[4] = {&d.extract2x, &d.remove2x, &d.demod_lp, &d.pald_lp};
    for (const cm_iir_desc *f : two_x) {
        const double r = slowest_pole(*f);
        if (r >= 1.0) return 1 << 30;
        if (r > 0.0) n = std::fmax(n, std::log(eps) / std::log(r * r));
    }
    const cm_iir_desc *one_x[2] = {&d.precorrect, &d.notch};
    for (const cm_iir_desc *f : one_x) {
        const double r = slowest_pole(*f);
        if (r >= 1.0) return 1 << 30;
        if (r > 0.0) n = std::fmax(n, std::log(eps) / std::log(r));
    }
    return ((int)std::ceil(n) + 24 + 31) & ~31;     // + the half-band windows, on an input tile boundary (32 samples: byte tiles)
}
// S = number of segments for a launch of `blocks` workgroups over rows of wp samples (1: not worth it)
static int segment_geometry(const cm_plan *p, int wp, long long blocks, int &seg_len) {
    seg_len = 0;
    if (!CM_SEGMENTS || !p->pair || p->blk_tiles || blocks <= 0 || blocks > 384) return 1;
    const int warm = p->seg_warm, lat = 56;
    if (warm >= wp) return 1;
    long long want = 1536 / blocks;                             // six workgroups per CU in all: ONE round of resident workgroups
    if (want < 2) return 1;
    int len = (int)((wp + want - 1) / want);
    len = (len + 15) & ~15;
    if (len < 48) len = 48;
    const int S = (wp + len - 1) / len;
    if (S < 2 || 10 * (warm + len + lat) > 7 * (wp + lat)) return 1;      // less than 30 % shorter: not worth the extra work
    seg_len = len;
    return S;
}

#endif  // CM_MAIN_PART
// ---- small batches: one wavefront per scan line (cm_scan_kernels.h) ------------------------------------------------------
// The scan's chunk-to-chunk transitions: A^(chunk 2^k) of every section, A = [[-a1, 1], [-a2, 0]] with the float32-rounded
// coefficients the kernel filters with (float64 products, rounded once).
template <typename T, class Filter>      // Filter = ScanFilter (T = float) or ScanFilterD (T = double); cut: where a power counts as decayed
static void fill_scan_filter(const cm_iir_desc &d, const T *na1, const T *na2, const T *b1, const T *b2, int chunk, Filter &f, double cut = 1e-12) {
    std::memset(&f, 0, sizeof f);
    f.nsec = d.n_sections;
    f.shift = d.shift;
    auto mul = [](const double (&x)[4], const double (&y)[4], double (&r)[4]) {
        const double t[4] = {x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3], x[2] * y[0] + x[3] * y[2], x[2] * y[1] + x[3] * y[3]};
        std::memcpy(r, t, sizeof t);
    };
    for (int j = 0; j < d.n_sections && j < kScanSec; ++j) {
        f.na1[j] = na1[j]; f.na2[j] = na2[j]; f.b1[j] = b1[j]; f.b2[j] = b2[j];
        double a[4] = {(double)na1[j], 1.0, (double)na2[j], 0.0}, m[4] = {1.0, 0.0, 0.0, 1.0};
        for (int e = chunk; e > 0; e >>= 1) {      // m = a^chunk
            if (e & 1) mul(m, a, m);
            mul(a, a, a);
        }
        f.steps[j] = kScanSteps;
        for (int k = 0; k < kScanSteps; ++k) {
            double big = 0.0;
            for (int e = 0; e < 4; ++e) {
                f.m[j][k][e] = (T)m[e];
                big = std::fmax(big, std::fabs(m[e]));
            }
            if (big < cut && f.steps[j] == kScanSteps) f.steps[j] = k;
            mul(m, m, m);
        }
    }
}
#if CM_MAIN_PART
static bool build_scan_k(const cm_plan_desc &d, bool pald, bool bsf, int depth, bool minavg, bool notch, int c1, ScanK &s, std::string &err) {
    DemodK<float, SysAny> k;
    DemodScales sc;
    if (!fits_any(signature_wanted(d, pald))) { err = "filter shape beyond the run-time maxima"; return false; }
    if (!build_demod_k<float, SysAny>(d, pald, bsf, k, sc, err)) return false;
    std::memset(&s, 0, sizeof s);
    s.width = d.width; s.pald = pald; s.bsf = bsf; s.depth = depth; s.minavg = minavg; s.c1 = c1;
    for (int i = 0; i < 10; ++i) s.taps[i] = k.taps.c[i];
    s.c0 = k.taps.c0;
    const cm_iir_desc &lp = pald ? d.pald_lp : d.demod_lp;
    fill_scan_filter(d.extract2x, k.ext.na1, k.ext.na2, k.ext.b1, k.ext.b2, 2 * c1, s.ext);
    if (bsf) fill_scan_filter(d.remove2x, k.rem.na1, k.rem.na2, k.rem.b1, k.rem.b2, 2 * c1, s.rem);
    fill_scan_filter(lp, k.lpf.na1, k.lpf.na2, k.lpf.b1, k.lpf.b2, 2 * c1, s.lpf);
    fill_scan_filter(d.precorrect, k.pre.na1, k.pre.na2, k.pre.b1, k.pre.b2, c1, s.pre);
    if (notch && d.notch.n_sections) fill_scan_filter(d.notch, k.notch.na1, k.notch.na2, k.notch.b1, k.notch.b2, c1, s.notch);
    s.luma_gain = k.luma_gain;
    s.notch_gain = notch ? k.notch_gain : 0.f;
    for (int i = 0; i < 9; ++i) s.m[i] = k.m[i / 3][i % 3];
    const int s2 = std::max(std::max(s.ext.shift, s.lpf.shift), bsf ? s.rem.shift : 0);
    if (s2 > kScanMaxShift || s.pre.shift > kScanMaxShift) { err = "FilterFunction shift beyond the scan kernel's margins"; return false; }
    if (2 * d.width + s2 > 128 * c1 || d.width + s.pre.shift > 64 * c1) { err = "row longer than the scan kernel's chunks"; return false; }
    return true;
}
// which chunk size serves a width (0: none compiled)
static int scan_chunk_for(const cm_plan_desc &d) {
    const int lp = d.pipeline == CM_PIPE_PAL_D ? d.pald_lp.shift : d.demod_lp.shift;
    const int s2 = std::max(std::max(d.extract2x.shift, lp), d.remove2x.shift);
    for (int c1 : {12, 16, 24, 32})
        if (2 * d.width + s2 <= 128 * c1 && d.width + d.precorrect.shift <= 64 * c1) return c1;
    return 0;
}
static void make_scan(cm_plan *p, const cm_plan_desc &d) {
    if (p->secam || !p->fn || d.skip_calls) return;      // (the fused wrapped comb's plan runs long batches only)
    const int c1 = scan_chunk_for(d);
    if (!c1) return;
    const bool pald = d.pipeline == CM_PIPE_PAL_D, bsf = d.main_luma_bandstop != 0, minavg = d.chroma_average == CM_AVG_MIN;
    const int depth = p->main.depth;        // the halo of the instance the lane tables were made for
    if (depth > 2 || (p->main.luma_prev_bits && depth < 1)) return;
    std::string err;
    ScanK km, kf;
    if (!build_scan_k(d, pald, bsf, depth, minavg, true, c1, km, err)) return;
    if (p->has_first && !build_scan_k(d, false, true, 0, false, false, c1, kf, err)) return;
    if (hipMalloc((void **)&p->scan_main, sizeof km) != hipSuccess || hipMemcpy(p->scan_main, &km, sizeof km, hipMemcpyHostToDevice) != hipSuccess) {
        p->scan_main = nullptr;
        return;
    }
    if (p->has_first && (hipMalloc((void **)&p->scan_first, sizeof kf) != hipSuccess || hipMemcpy(p->scan_first, &kf, sizeof kf, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipFree(p->scan_main);
        p->scan_main = p->scan_first = nullptr;
        return;
    }
    p->scan_c1 = c1;
    p->scan_depth = depth;
}
// the row-parallel modulator of small batches (cm_scan_kernels.h: qam_mod_scan_kernel)
static void make_scan_mod(cm_plan *p, const cm_plan_desc &d) {
    if (p->secam || !p->mod_fn || d.precorrect.n_sections > 2 || d.precorrect.shift < 0 || d.precorrect.shift > kScanMaxShift) return;
    int c1 = 0;
    for (int c : {12, 16, 24, 32})
        if (d.width + d.precorrect.shift <= 64 * c) { c1 = c; break; }
    if (!c1) return;
    SosK<float, 2> pre;
    double g_pre;
    std::string err;
    if (!convert_sos<float, 2>(d.precorrect, FORM_GEN, pre, g_pre, err, "precorrect", true)) return;
    ScanModK k;
    std::memset(&k, 0, sizeof k);
    k.width = d.width;
    k.depth = p->mod_depth;
    k.c1 = c1;
    for (int i = 0; i < 9; ++i) k.e[i] = (float)d.encode_matrix[i];
    fill_scan_filter(d.precorrect, pre.na1, pre.na2, pre.b1, pre.b2, c1, k.pre);
    if (hipMalloc((void **)&p->scan_mod, sizeof k) != hipSuccess || hipMemcpy(p->scan_mod, &k, sizeof k, hipMemcpyHostToDevice) != hipSuccess) {
        p->scan_mod = nullptr;
        return;
    }
    p->scan_mod_c1 = c1;
}
// the SECAM modulator's scan constants (called from create_secam once the streaming modulator's constants exist)
void make_scan_secam_mod(cm_plan *p, const cm_plan_desc &d) {
    const cm_secam_desc &sd = d.secam;
    if (!p->sm_lanes || sd.pre_lp.n_sections > 2 || sd.lf_pre.n_sections > 1 || sd.pre_lp.shift > kScanMaxShift) return;
    int c1 = 0;
    for (int c : {12, 16, 24, 32})
        if (d.width + sd.pre_lp.shift <= 64 * c) { c1 = c; break; }
    if (!c1) return;
    ScanSecamModK k;
    std::memset(&k, 0, sizeof k);
    const SecamModK<float, double> &m = p->sm_k;
    k.width = d.width; k.depth = p->mod_depth; k.c1 = c1;
    fill_scan_filter(sd.pre_lp, m.pre_lp.na1, m.pre_lp.na2, m.pre_lp.b1, m.pre_lp.b2, c1, k.pre_lp, 1e-20);
    fill_scan_filter(sd.lf_pre, m.lf_pre.na1, m.lf_pre.na2, m.lf_pre.b1, m.lf_pre.b2, c1, k.lf_pre, 1e-20);
    k.gain = m.gain; k.f_min = m.f_min; k.f_max = m.f_max; k.f0 = m.f0; k.pi = m.pi; k.two_pi = m.two_pi;
    k.m0 = m.m0; k.kn = m.kn; k.kd = m.kd;
    for (int i = 0; i < 9; ++i) k.e[i] = m.e[i / 3][i % 3];
    if (hipMalloc((void **)&p->scan_smod, sizeof k) != hipSuccess || hipMemcpy(p->scan_smod, &k, sizeof k, hipMemcpyHostToDevice) != hipSuccess) {
        p->scan_smod = nullptr;
        return;
    }
    p->scan_smod_c1 = c1;
}
// the SECAM decoder's scan constants: band-pass + bell in float64 (SecamBp64's sections), the rest as the streaming kernel has it
void make_scan_secam_demod(cm_plan *p, const cm_plan_desc &d) {
    const cm_secam_desc &sd = d.secam;
    if (p->sd_f64 || !p->sd_lanes) return;               // (the thin-margin shapes keep their float64 front end: streaming kernel)
    const int Lc = d.width + sd.preroll;
    if (sd.chroma_bp.shift > kScanMaxShift || sd.fm_lp.shift > kScanMaxShift || sd.luma_bs.shift > kScanMaxShift || sd.preroll > kScanMaxShift ||
        sd.bell.shift != 0 || sd.lf_rev.shift != 0)
        return;
    int c1 = 0;
    for (int c : {12, 16})
        if (Lc + sd.chroma_bp.shift <= 64 * c && 2 * Lc + sd.fm_lp.shift <= 128 * c && d.width + sd.luma_bs.shift <= 64 * c) { c1 = c; break; }
    if (!c1) return;
    ScanSecamK k;
    std::memset(&k, 0, sizeof k);
    const SecamDemodK<float> &m = p->sd_k;
    k.width = d.width; k.preroll = sd.preroll; k.c1 = c1; k.has_bell = m.has_bell;
    for (int i = 0; i < 10; ++i) k.taps[i] = m.taps.c[i];
    k.c0 = m.taps.c0;
    k.two_over_pi = m.two_over_pi;
    const SecamBp64 &e64 = p->sd_e64;
    fill_scan_filter(sd.chroma_bp, e64.bpf.na1, e64.bpf.na2, e64.bpf.b1, e64.bpf.b2, c1, k.bpf, 1e-20);
    fill_scan_filter(sd.bell, e64.bell.na1, e64.bell.na2, e64.bell.b1, e64.bell.b2, c1, k.bell, 1e-20);
    fill_scan_filter(sd.fm_lp, m.lpf.na1, m.lpf.na2, m.lpf.b1, m.lpf.b2, 2 * c1, k.lpf);
    fill_scan_filter(sd.luma_bs, m.ybs.na1, m.ybs.na2, m.ybs.b1, m.ybs.b2, c1, k.ybs);
    fill_scan_filter(sd.lf_rev, m.deemph.na1, m.deemph.na2, m.deemph.b1, m.deemph.b2, c1, k.deemph);
    k.luma_gain = m.luma_gain;
    for (int i = 0; i < 9; ++i) k.m[i] = m.m[i / 3][i % 3];
    if (hipMalloc((void **)&p->scan_sdem, sizeof k) != hipSuccess || hipMemcpy(p->scan_sdem, &k, sizeof k, hipMemcpyHostToDevice) != hipSuccess) {
        p->scan_sdem = nullptr;
        return;
    }
    p->scan_sdem_c1 = c1;
}
int scan_secam_demod(const cm_plan *p, const Geom &g, hipStream_t stream, bool u8) {
    return cm_host::scan_launch_secam_demod(p->scan_sdem_c1, u8, p->device, p->scan_sdem, g, stream);
}
int scan_secam_mod(const cm_plan *p, const Geom &g, hipStream_t stream, bool u8) {
    return cm_host::scan_launch_secam_mod(p->scan_smod_c1, u8, p->device, p->scan_smod, g, stream);
}
// km / kf: the constants of the main pass and of the plain first-line pass (two plans' in the wrapped combs); depth: the main pass's comb depth
template <bool U8>
static int launch_scan_as(int c1, int device, const ScanK *km, const ScanK *kf, int depth, const Geom &gm, const Geom &gf, bool with_first,
                          hipStream_t stream) {
    return cm_host::scan_launch_demod(c1, U8, device, km, kf, depth, gm, gf, with_first, stream);
}

// gm: main-pass geometry (total_calls set); gf: first-line geometry (total_calls = number of runs) when the plan has one
#ifdef CM_DIAG
static unsigned long long *g_diag;
#endif
int run_plan(const cm_plan *p, Geom gm, Geom gf, bool with_first, hipStream_t stream, bool u8 = false) {
    finish_geom(p, p->main, gm);
    gm.simd_load = p->simd_load;
    gm.blk_tiles = p->blk_tiles;
#ifdef CM_DIAG
    gm.diag = g_diag;
#endif
    long long n_main = (gm.total_calls + (64 - p->main.depth) - 1) / (64 - p->main.depth);
    long long n_first = 0;
    if (with_first) {
        finish_geom(p, p->first, gf);
        n_first = (gf.total_calls + 63) / 64;
    }
    if (n_main + n_first <= 0) return CM_OK;
    if (n_main + n_first > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    LaunchFn fn = u8 ? p->fn_u8 : p->fn;
    if (!fn) return fail(CM_ERR_UNSUPPORTED, "no kernel instance for this request");
    const int mode = p->small_batch;
    if (p->scan_main && (mode == CM_SMALL_BATCH_SCAN || (mode == CM_SMALL_BATCH_AUTO && gm.total_calls <= CM_SCAN_MAX_CALLS)))
        return u8 ? launch_scan_as<true>(p->scan_c1, p->device, p->scan_main, p->scan_first, p->scan_depth, gm, gf, with_first, stream)
                  : launch_scan_as<false>(p->scan_c1, p->device, p->scan_main, p->scan_first, p->scan_depth, gm, gf, with_first, stream);
    if (mode == CM_SMALL_BATCH_SCAN) return fail(CM_ERR_UNSUPPORTED, "the scan kernel does not serve this plan / this entry point");
    int seg_len = 0;
    const int S = mode == CM_SMALL_BATCH_ROWS ? 1 : segment_geometry(p, gm.Wp, n_main + n_first, seg_len);
    if (S > 1) {      // few workgroups: every one walks a segment of its rows (blocks [seg * n, (seg + 1) * n) of each pass)
        gm.seg_len = gf.seg_len = seg_len;
        gm.seg_warm = gf.seg_warm = p->seg_warm;
        gm.seg_blocks = (int)n_main;
        gf.seg_blocks = (int)n_first;
        n_main *= S;
        n_first *= S;
    }
    return fn(gm, p->main.k.data(), gf, with_first ? p->first.k.data() : nullptr, (int)n_first, (int)n_main, stream);
}

int check_lines(const cm_plan *p, const Pass &pass, int max_line) {
    if (max_line >= pass.n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's phase tables");
    (void)p;
    return CM_OK;
}


#endif  // CM_MAIN_PART
// Scratch under stream capture: hipMallocAsync / hipFreeAsync on a capturing stream become memory nodes of the graph, and graphs of
// wrapped-comb calls with such nodes faulted on replay on ROCm 7.2, at 512 and at 256 frames per call, run-to-run differently
// (profiles/r03_wrapped_small_batch.txt) - the same calls made eagerly are exact at every size.  The entry points that need scratch
// (the wrapped combs; widths that are not a multiple of 4) refuse a capturing stream instead of leaving it to the runtime.
static int refuse_capture(hipStream_t stream, const char *what) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(CM_ERR_UNSUPPORTED, std::string(what) + " needs stream-ordered scratch memory and cannot be captured into a HIP graph");
    return CM_OK;
}

// ---- widths that are not multiples of 4 ---------------------------------------------------------------------------------
// The kernels move rows as 16-byte vectors and need every row 16-byte aligned.  Dense float images of such a width go
// through device buffers whose rows are pitched to the next multiple of 4 samples: one strided copy in, one out, both on
// the caller's stream (stream-ordered allocations).  The pad samples are never read as data and what lands in them on the
// way out is dropped by the copy.
struct PitchedIO {
    hipStream_t stream = nullptr;
    float *in = nullptr, *out = nullptr;     // null: the caller's dense buffer is used directly
    ~PitchedIO() {
        if (in) (void)hipFreeAsync(in, stream);
        if (out) (void)hipFreeAsync(out, stream);
    }
};
// run(in_ptr, out_ptr) launches on buffers with rows of `wp` samples
template <class F>
int with_pitched_rows(const float *in, long long in_rows, float *out, long long out_rows, int W, hipStream_t stream, F run) {
    const int wp = (W + 3) & ~3;
    if (wp == W) return run(in, out);
    if (int rc_ = refuse_capture(stream, "an image width that is not a multiple of 4")) return rc_;
    PitchedIO io;
    io.stream = stream;
    HIP_TRY(hipMallocAsync((void **)&io.in, (size_t)in_rows * wp * sizeof(float), stream), CM_ERR_LAUNCH);
    HIP_TRY(hipMallocAsync((void **)&io.out, (size_t)out_rows * wp * sizeof(float), stream), CM_ERR_LAUNCH);
    HIP_TRY(hipMemcpy2DAsync(io.in, (size_t)wp * sizeof(float), in, (size_t)W * sizeof(float), (size_t)W * sizeof(float), (size_t)in_rows,
                             hipMemcpyDeviceToDevice, stream), CM_ERR_LAUNCH);
    int rc = run(io.in, io.out);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)W * sizeof(float), io.out, (size_t)wp * sizeof(float), (size_t)W * sizeof(float), (size_t)out_rows,
                             hipMemcpyDeviceToDevice, stream), CM_ERR_LAUNCH);
    return CM_OK;
}

}  // namespace

#if CM_MAIN_PART
extern "C" {

const char *cm_last_error(void) { return g_error.c_str(); }
int cm_abi_version(void) { return CM_ABI_VERSION; }

int cm_device_count(void) {
#ifdef CM_HOST_DRY_RUN
    return 1;      // the host sanitizer build: plans are built against host memory (top of this file)
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
#endif
}

// ---- FilterFunction.__call__ (utils.py:28-36) as a callable of its own: float64, one lane per row -------------------------------
}  // extern "C"
namespace {
struct FilterRowsArgs {
    double b[CM_FILTER_MAX_TAPS], a[CM_FILTER_MAX_TAPS];     // a[0] = 1 (normalised on the host), zero-padded
    int n_taps, shift, width, skip_inner;
    long long rows, n_inner, outer_stride, inner_stride;      // row r = (outer, inner) = (r / n_inner, r % n_inner), in elements
    const void *x;
    void *y;
};
// scipy.signal.lfilter's recurrence (transposed direct form II: y = z0 + b0 x; z_i = z_{i+1} + b_{i+1} x - a_{i+1} y), unfused
// multiply / add / subtract in its order, on the row padded as utils.py:31-35 pads it: `shift` copies of the last sample behind it and the
// first `shift` results dropped (shift > 0), or -shift copies of the first sample in front and the last -shift results dropped (shift < 0).
// T: the rows' element type (the arithmetic is float64 either way); rows with inner index < skip_inner are copied unfiltered.
template <typename T>
__global__ __launch_bounds__(64) void filter_rows_kernel(const FilterRowsArgs k) {
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    if (row >= k.rows) return;
    const long long outer = row / k.n_inner, inner = row - outer * k.n_inner;
    const T *x = (const T *)k.x + outer * k.outer_stride + inner * k.inner_stride;
    T *y = (T *)k.y + outer * k.outer_stride + inner * k.inner_stride;
    const int W = k.width;
    if (inner < k.skip_inner) {
        if (x != y)
            for (int t = 0; t < W; ++t) y[t] = x[t];
        return;
    }
    double z[CM_FILTER_MAX_TAPS];
#pragma unroll
    for (int i = 0; i < CM_FILTER_MAX_TAPS; ++i) z[i] = 0.0;
    const int s = k.shift, lead = s < 0 ? -s : 0, drop = s > 0 ? s : 0;
    const int total = W + lead + drop;
    for (int t = 0; t < total; ++t) {
        int j = t - lead;
        j = j < 0 ? 0 : (j > W - 1 ? W - 1 : j);
        const double xin = (double)x[j];
        const double out = __dadd_rn(z[0], __dmul_rn(k.b[0], xin));
#pragma unroll
        for (int i = 0; i < CM_FILTER_MAX_TAPS - 1; ++i)
            if (i + 1 < k.n_taps) z[i] = __dsub_rn(__dadd_rn(z[i + 1], __dmul_rn(xin, k.b[i + 1])), __dmul_rn(out, k.a[i + 1]));
        const int o = t - drop;
        if (o >= 0 && o < W) y[o] = (T)out;
    }
}
// (r, g, b) = M (y, u, v) on [group][3][plane] floats, in place or not (decode_components after the luma notch of cm_notch_luma_f32)
__global__ __launch_bounds__(256) void matrix_planes_kernel(const float *in, float *out, long long n, long long plane, float m00, float m01, float m02,
                                                            float m10, float m11, float m12, float m20, float m21, float m22) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long g = i / plane, o = g * 3 * plane + (i - g * plane);
    const float y = in[o], u = in[o + plane], v = in[o + 2 * plane];
    out[o] = fmaf_(m00, y, fmaf_(m01, u, m02 * v));
    out[o + plane] = fmaf_(m10, y, fmaf_(m11, u, m12 * v));
    out[o + 2 * plane] = fmaf_(m20, y, fmaf_(m21, u, m22 * v));
}
int fill_filter_args(FilterRowsArgs &k, const double *b, int32_t n_b, const double *a, int32_t n_a, int32_t shift, int32_t width) {
    if (!b || !a || n_b < 1 || n_a < 1) return fail(CM_ERR_INVALID, "null or empty coefficient array");
    if (n_b > CM_FILTER_MAX_TAPS || n_a > CM_FILTER_MAX_TAPS)
        return fail(CM_ERR_UNSUPPORTED, "filter order beyond CM_FILTER_MAX_TAPS - 1 = " + std::to_string(CM_FILTER_MAX_TAPS - 1));
    if (a[0] == 0.0) return fail(CM_ERR_INVALID, "a[0] must not be zero");
    if (width < 1) return fail(CM_ERR_INVALID, "width must be positive");
    if (shift <= -width || shift >= (1 << 20)) return fail(CM_ERR_INVALID, "shift out of range");
    std::memset(&k, 0, sizeof k);
    k.n_taps = n_b > n_a ? n_b : n_a;
    for (int i = 0; i < n_b; ++i) k.b[i] = b[i] / a[0];      // lfilter normalises by a[0] first
    for (int i = 0; i < n_a; ++i) k.a[i] = a[i] / a[0];
    k.shift = shift;
    k.width = width;
    return CM_OK;
}
template <typename T>
int launch_filter_rows(const FilterRowsArgs &k, void *stream) {
    const long long blocks = (k.rows + 63) / 64;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    hipLaunchKernelGGL(filter_rows_kernel<T>, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("filter_rows_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
}  // namespace
extern "C" {
int cm_filter_rows_f64(const double *b, int32_t n_b, const double *a, int32_t n_a, int32_t shift, const double *x, double *y, int64_t n_rows,
                       int32_t width, void *stream) {
    FilterRowsArgs k;
    if (int rc = fill_filter_args(k, b, n_b, a, n_a, shift, width)) return rc;
    if (n_rows < 0) return fail(CM_ERR_INVALID, "rows must not be negative");
    if (n_rows == 0) return CM_OK;
    if (!x || !y) return fail(CM_ERR_INVALID, "null argument");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(CM_ERR_NO_DEVICE, "hipGetDevice failed");
    if (int rc_ = check_device(cur, x, y)) return rc_;
    k.rows = n_rows;
    k.n_inner = n_rows;
    k.inner_stride = width;
    k.x = x;
    k.y = y;
    return launch_filter_rows<double>(k, stream);
}

int cm_notch_luma_f32(const double *b, int32_t n_b, const double *a, int32_t n_a, int32_t shift, const float *yuv_in, float *yuv_out,
                      int64_t n_groups, int64_t rows_per_group, int32_t width, int32_t skip_rows, const double *matrix, void *stream) {
    FilterRowsArgs k;
    if (int rc = fill_filter_args(k, b, n_b, a, n_a, shift, width)) return rc;
    if (n_groups < 0 || rows_per_group < 1 || skip_rows < 0) return fail(CM_ERR_INVALID, "negative count");
    if (n_groups == 0) return CM_OK;
    if (!yuv_in || !yuv_out || yuv_in == yuv_out || !matrix) return fail(CM_ERR_INVALID, "null argument, or input and output are the same buffer");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(CM_ERR_NO_DEVICE, "hipGetDevice failed");
    if (int rc_ = check_device(cur, yuv_in, yuv_out)) return rc_;
    const long long plane = rows_per_group * (long long)width;
    // the luma plane of every group through the notch (rows below skip_rows as they are), the two chroma planes carried over
    k.rows = n_groups * rows_per_group;
    k.n_inner = rows_per_group;
    k.outer_stride = 3 * plane;
    k.inner_stride = width;
    k.skip_inner = skip_rows;
    k.x = yuv_in;
    k.y = yuv_out;
    if (int rc = launch_filter_rows<float>(k, stream)) return rc;
    for (int c = 1; c < 3; ++c)
        HIP_TRY(hipMemcpy2DAsync(yuv_out + c * plane, 3 * plane * sizeof(float), yuv_in + c * plane, 3 * plane * sizeof(float), plane * sizeof(float),
                                 (size_t)n_groups, hipMemcpyDeviceToDevice, (hipStream_t)stream), CM_ERR_LAUNCH);
    const long long n = n_groups * plane;
    if ((n + 255) / 256 > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    const double *m = matrix;
    hipLaunchKernelGGL(matrix_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, yuv_out, yuv_out, n, plane, (float)m[0],
                       (float)m[1], (float)m[2], (float)m[3], (float)m[4], (float)m[5], (float)m[6], (float)m[7], (float)m[8]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("matrix_planes_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}

int cm_plan_create(const cm_plan_desc *desc, cm_plan **out) {
    if (!desc || !out) return fail(CM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->abi_version != CM_ABI_VERSION) return fail(CM_ERR_INVALID, "descriptor ABI version mismatch");
    if (desc->width < 4) return fail(CM_ERR_UNSUPPORTED, "width must be at least 4");
    if (desc->height < 1) return fail(CM_ERR_INVALID, "height must be positive");
    if (desc->pipeline != CM_PIPE_QAM && desc->pipeline != CM_PIPE_PAL_D && desc->pipeline != CM_PIPE_SECAM)
        return fail(CM_ERR_INVALID, "unknown pipeline");
    if (desc->demod_main.wrap_mode < 0 || desc->demod_main.wrap_mode > 2) return fail(CM_ERR_INVALID, "demod_main.wrap_mode must be 0, 1 or 2");
    if (desc->depth < 0 || desc->depth > (desc->demod_main.wrap_mode ? 3 : 2)) return fail(CM_ERR_INVALID, "depth must be 0..2 (3 with demod_main.wrap_mode)");
    if (desc->skip_calls != 0 && desc->skip_calls != 2) return fail(CM_ERR_INVALID, "skip_calls must be 0 or 2");
    if (!desc->demod_main.table || desc->demod_main.frame_cycle < 1 || desc->demod_main.n_lines < 1)
        return fail(CM_ERR_INVALID, "demod_main table missing");
    if (desc->first_is_plain && (!desc->demod_first.table || desc->demod_first.n_lines != desc->demod_main.n_lines))
        return fail(CM_ERR_INVALID, "demod_first table missing or of different size");
    {   // every filter record: a section count the descriptor can hold, a FilterFunction shift of sane size (found by the host sanitizer sweep of
        // round 6: a negative count slipped through the run-time shape's padding as "no sections")
        const struct { const cm_iir_desc *f; const char *name; } filters[] = {
            {&desc->extract2x, "extract2x"}, {&desc->remove2x, "remove2x"}, {&desc->demod_lp, "demod_lp"}, {&desc->pald_lp, "pald_lp"},
            {&desc->precorrect, "precorrect"}, {&desc->notch, "notch"}, {&desc->secam.pre_lp, "secam.pre_lp"}, {&desc->secam.lf_pre, "secam.lf_pre"},
            {&desc->secam.lf_rev, "secam.lf_rev"}, {&desc->secam.bell, "secam.bell"}, {&desc->secam.chroma_bp, "secam.chroma_bp"},
            {&desc->secam.luma_bs, "secam.luma_bs"}, {&desc->secam.fm_lp, "secam.fm_lp"}};
        for (const auto &e : filters) {
            if (e.f->n_sections < 0 || e.f->n_sections > CM_MAX_SECTIONS)
                return fail(CM_ERR_INVALID, std::string(e.name) + ": n_sections must be 0 .. " + std::to_string(CM_MAX_SECTIONS));
            if (e.f->shift < -4096 || e.f->shift > 4096) return fail(CM_ERR_INVALID, std::string(e.name) + ": FilterFunction shift out of range");
        }
    }
    if (cm_device_count() < 1) return fail(CM_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    cm_plan *p = new cm_plan;
    p->desc = *desc;
    p->desc.demod_main.table = p->desc.demod_first.table = p->desc.mod_main.table = nullptr;  // not retained
    p->desc.frame_rotation = nullptr;
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return fail(CM_ERR_NO_DEVICE, "hipGetDevice failed");
    }
    std::string err;
    if (desc->pipeline == CM_PIPE_SECAM) {
        if (!create_secam(p, *desc, err)) {
            cm_plan_destroy(p);
            return fail(CM_ERR_UNSUPPORTED, err);
        }
        *out = p;
        return CM_OK;
    }
    if (CM_SIMD_BALANCE) {   // counters return to zero with every kernel (each workgroup takes back what it added)
        if (hipMalloc((void **)&p->simd_load, kSimdLoadEntries * sizeof(unsigned)) != hipSuccess ||
            hipMemset(p->simd_load, 0, kSimdLoadEntries * sizeof(unsigned)) != hipSuccess) {
            cm_plan_destroy(p);
            return fail(CM_ERR_LAUNCH, "device allocation of the SIMD load counters failed");
        }
    }
    // Carrier tables, padded by kCarrierPad entries on both sides with copies of the first / last entry: the kernels
    // index them with stream positions that run from -latency to W + latency and the padding stands for the clamp.
    std::vector<float> car0 = build_carrier<float>(desc->carrier_phase_step, desc->width);  // {C[m], S[m]}, m < 2W
    const int W_ = desc->width, P_ = kCarrierPad;
    std::vector<float> car(4 * (size_t)(W_ + 2 * P_)), car2(2 * (size_t)(W_ + 2 * P_));   // {C, S}[2n, 2n+1]; {C, S}[2n]
    for (int i = 0; i < W_ + 2 * P_; ++i) {
        int n = i - P_;
        n = n < 0 ? 0 : (n > W_ - 1 ? W_ - 1 : n);
        for (int j = 0; j < 4; ++j) car[4 * (size_t)i + j] = car0[4 * (size_t)n + j];
        car2[2 * (size_t)i] = car0[4 * (size_t)n];
        car2[2 * (size_t)i + 1] = car0[4 * (size_t)n + 1];
    }
    if (hipMalloc((void **)&p->carrier4_base, car.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->carrier4_base, car.data(), car.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void **)&p->carrier2_base, car2.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->carrier2_base, car2.data(), car2.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        cm_plan_destroy(p);
        return fail(CM_ERR_NO_DEVICE, "device allocation / upload of the carrier tables failed");
    }
    p->carrier4 = p->carrier4_base + 4 * (size_t)P_;   // entry 0
    p->carrier2 = p->carrier2_base + 2 * (size_t)P_;
    if (desc->frame_rotation) {
        const int n = desc->frame_rotation_cycle;
        const cm_lane_table *tabs[3] = {&desc->demod_main, &desc->demod_first, &desc->mod_main};
        bool ok = n >= 2 && n % 2 == 0;
        for (const cm_lane_table *t : tabs) ok = ok && (!t->table || t->frame_cycle == 2);
        if (!ok) {
            cm_plan_destroy(p);
            return fail(CM_ERR_INVALID, "frame_rotation needs an even cycle and lane tables of exactly two frames");
        }
        std::vector<float> rot(2 * (size_t)n);
        for (size_t i = 0; i < rot.size(); ++i) rot[i] = (float)desc->frame_rotation[i];
        if (hipMalloc((void **)&p->frame_rot, rot.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(p->frame_rot, rot.data(), rot.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            cm_plan_destroy(p);
            return fail(CM_ERR_NO_DEVICE, "device allocation / upload of the frame rotation table failed");
        }
        p->rot_cycle = n;
    }
    // a plan is usable in one direction when only the other one lacks a kernel instance
    std::string mod_err;
    p->seg_warm = segment_warmup(*desc);
    const bool have_demod = select_kernels(p, *desc, err);
    const bool have_mod = select_modulator(p, *desc, mod_err) && p->mod_fn;
    if (!have_demod) {
        p->fn = nullptr;
        p->demod_error = err;
    } else {
        make_scan(p, *desc);
    }
    if (have_mod) make_scan_mod(p, *desc);
    if (!have_demod && !have_mod) {
        cm_plan_destroy(p);
        return fail(CM_ERR_UNSUPPORTED, err);
    }
    *out = p;
    return CM_OK;
}

void cm_plan_destroy(cm_plan *p) {
    if (!p) return;
    if (p->carrier4_base) (void)hipFree(p->carrier4_base);
    if (p->carrier2_base) (void)hipFree(p->carrier2_base);
    if (p->frame_rot) (void)hipFree(p->frame_rot);
    if (p->simd_load) (void)hipFree(p->simd_load);
    if (p->blk_tiles) (void)hipFree(p->blk_tiles);
    if (p->scan_main) (void)hipFree(p->scan_main);
    if (p->scan_first) (void)hipFree(p->scan_first);
    if (p->scan_mod) (void)hipFree(p->scan_mod);
    if (p->scan_smod) (void)hipFree(p->scan_smod);
    if (p->scan_sdem) (void)hipFree(p->scan_sdem);
    if (p->main.lanes) (void)hipFree(p->main.lanes);
    if (p->first.lanes) (void)hipFree(p->first.lanes);
    if (p->mod_lanes) (void)hipFree(p->mod_lanes);
    if (p->sd_lanes) (void)hipFree(p->sd_lanes);
    if (p->sm_lanes) (void)hipFree(p->sm_lanes);
    if (p->fm_ref) (void)hipFree(p->fm_ref);
    if (p->fm_ref64) (void)hipFree(p->fm_ref64);
    if (p->fm_dc) (void)hipFree(p->fm_dc);
    delete p;
}

int cm_demodulate_frames(const cm_plan *p, const float *composite, float *rgb, int64_t n_frames, int64_t first_frame,
                         void *stream) {
    if (p && n_frames == 0) return CM_OK;   // an empty batch may come with null buffers
    if (!p || !composite || !rgb) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (!p->fn && !p->secam) return fail(CM_ERR_UNSUPPORTED, p->demod_error);
    if (int rc_ = check_device(p->device, composite, rgb)) return rc_;
    const cm_plan_desc &d = p->desc;
    const int W = d.width, H = d.height, D = d.demodulation_delay;
    const int wp = (W + 3) & ~3;
    return with_pitched_rows(composite, n_frames * H, rgb, n_frames * 3 * H, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        g.W = W;
        g.Wp = wp;
        g.H = H;
        g.in_frame_stride = (long long)wp * H;
        g.in_row_stride = wp;
        g.out_plane_stride = (long long)wp * H;
        g.out_frame_stride = 3LL * wp * H;
        g.out_row_stride = wp;
        set_first_frame(p, g, first_frame, p->secam ? p->sd_cycle : p->main.cycle);
        const int rows0 = (H + 1) / 2, rows1 = H / 2;
        g.calls_run0 = rows0 + D;
        const int calls_run1 = rows1 > 0 ? rows1 + D : 0;
        g.calls_per_frame = g.calls_run0 + calls_run1;
        g.runs_per_frame = rows1 > 0 ? 2 : 1;
        g.first_line[0] = 0;
        g.first_line[1] = 1;
        g.delay = D;
        g.total_calls = n_frames * g.calls_per_frame;
        g.skip_first = d.skip_calls ? d.skip_calls : d.first_is_plain;
        if (p->secam) {
            if (H - 1 >= p->sd_n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's tables");
            return run_secam_demod(p, g, (hipStream_t)stream);
        }
        int rc = check_lines(p, p->main, H - 1 + 2 * D);
        if (rc) return rc;
        Geom s = g;
        if (p->has_first) {
            s.sparse = 1;
            s.skip_first = 0;
            s.total_calls = n_frames * g.runs_per_frame;
            set_first_frame(p, s, first_frame, p->first.cycle);
        }
        return run_plan(p, g, s, p->has_first, (hipStream_t)stream);
    });
}

int cm_demodulate_frames_u8(const cm_plan *p, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames, int64_t first_frame,
                            void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !composite8 || !rgb8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (!p->secam && !p->fn_u8)
        return fail(CM_ERR_UNSUPPORTED, p->fn ? "no kernel instance with the fused uint8 boundary for this decoder"
                                              : p->demod_error);
    if (int rc_ = check_device(p->device, composite8, rgb8)) return rc_;
    const cm_plan_desc &d = p->desc;
    const int W = d.width, H = d.height, D = d.demodulation_delay;
    Geom g;
    std::memset(&g, 0, sizeof g);
    if (W % 4 != 0) return fail(CM_ERR_UNSUPPORTED, "the fused uint8 boundary needs a width that is a multiple of 4");
    g.in = reinterpret_cast<const float *>(composite8);   // strides below count bytes (PassCfg::U8)
    g.out = reinterpret_cast<float *>(rgb8);
    g.W = W;
    g.Wp = W;
    g.H = H;
    g.in_frame_stride = (long long)W * H;
    g.in_row_stride = W;
    g.out_plane_stride = 0;
    g.out_frame_stride = 3LL * W * H;
    g.out_row_stride = 3LL * W;
    set_first_frame(p, g, first_frame, p->secam ? p->sd_cycle : p->main.cycle);
    const int rows0 = (H + 1) / 2, rows1 = H / 2;
    g.calls_run0 = rows0 + D;
    const int calls_run1 = rows1 > 0 ? rows1 + D : 0;
    g.calls_per_frame = g.calls_run0 + calls_run1;
    g.runs_per_frame = rows1 > 0 ? 2 : 1;
    g.first_line[0] = 0;
    g.first_line[1] = 1;
    g.delay = D;
    g.total_calls = n_frames * g.calls_per_frame;
    g.skip_first = d.skip_calls ? d.skip_calls : d.first_is_plain;
    if (p->secam) {
        if (H - 1 >= p->sd_n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's tables");
        return run_secam_demod(p, g, (hipStream_t)stream, true);
    }
    int rc = check_lines(p, p->main, H - 1 + 2 * D);
    if (rc) return rc;
    Geom s = g;
    if (p->has_first) {
        s.sparse = 1;
        s.skip_first = 0;
        s.total_calls = n_frames * g.runs_per_frame;
        set_first_frame(p, s, first_frame, p->first.cycle);
    }
    return run_plan(p, g, s, p->has_first, (hipStream_t)stream, true);
}

int cm_demodulate_run(const cm_plan *p, const float *composite, float *rgb, int32_t n_calls, int32_t frame,
                      int32_t first_line, int32_t k0, void *stream) {
    if (!p || !composite || !rgb) return fail(CM_ERR_INVALID, "null argument");
    if (n_calls < 0 || frame < 0 || k0 < 0) return fail(CM_ERR_INVALID, "negative count / frame / k0");
    if (n_calls == 0) return CM_OK;
    if (first_line < 0) return fail(CM_ERR_INVALID, "negative line number");
    if (!p->fn && !p->secam) return fail(CM_ERR_UNSUPPORTED, p->demod_error);
    if (int rc_ = check_device(p->device, composite, rgb)) return rc_;
    const cm_plan_desc &d = p->desc;
    const int W = d.width, wp = (W + 3) & ~3;
    return with_pitched_rows(composite, n_calls, rgb, 3LL * n_calls, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        g.W = W;
        g.Wp = wp;
        g.H = n_calls;
        g.in_frame_stride = 0;
        g.out_frame_stride = 0;
        g.rows_mode = 1;
        set_first_frame(p, g, frame, p->secam ? p->sd_cycle : p->main.cycle);
        g.calls_run0 = n_calls;
        g.calls_per_frame = n_calls;
        g.runs_per_frame = 1;
        g.first_line[0] = g.first_line[1] = first_line;
        g.k0 = k0;
        g.total_calls = n_calls;
        g.skip_first = d.skip_calls ? d.skip_calls : d.first_is_plain;
        g.out_plane_stride = wp;            // rows mode writes [call][plane][W]
        g.out_row_stride = 3LL * wp;
        if (p->secam) {
            if (first_line + 2 * (n_calls - 1) >= p->sd_n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's tables");
            return run_secam_demod(p, g, (hipStream_t)stream);
        }
        int rc = check_lines(p, p->main, first_line + 2 * (n_calls - 1));
        if (rc) return rc;
        Geom s = g;
        const bool with_first = p->has_first && k0 == 0;
        if (with_first) {
            s.sparse = 1;
            s.skip_first = 0;
            s.total_calls = 1;
            set_first_frame(p, s, frame, p->first.cycle);
        }
        return run_plan(p, g, s, with_first, (hipStream_t)stream);
    });
}

#ifndef CM_SCAN_MOD_MAX_CALLS
#define CM_SCAN_MOD_MAX_CALLS 40000
#endif
static int run_mod(const cm_plan *p, Geom g, hipStream_t stream, bool u8 = false) {
    if (p->secam) return run_secam_mod(p, g, stream, u8);
    if (!p->mod_fn) return fail(CM_ERR_UNSUPPORTED, "this plan has no modulator");
    g.lanes = reinterpret_cast<const LaneK<float> *>(p->mod_lanes);
    g.carrier4 = p->carrier4;
    g.carrier2 = p->carrier2;
    g.cycle = p->mod_cycle;
    g.n_lines = p->mod_n_lines;
    long long blocks = (g.total_calls + (64 - p->mod_depth) - 1) / (64 - p->mod_depth);
    if (blocks <= 0) return CM_OK;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    const int mode = p->small_batch;
    if (p->scan_mod && (mode == CM_SMALL_BATCH_SCAN || (mode == CM_SMALL_BATCH_AUTO && g.total_calls <= CM_SCAN_MOD_MAX_CALLS)))
        return cm_host::scan_launch_qam_mod(p->scan_mod_c1, u8, p->device, p->scan_mod, g, stream);
    return (u8 ? p->mod_fn_u8 : p->mod_fn)(g, p->mod_k.data(), (int)blocks, stream);
}

int cm_modulate_frames(const cm_plan *p, const float *rgb, float *composite, int64_t n_frames, int64_t first_frame,
                       void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !rgb || !composite) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (!p->mod_fn && !p->sm_lanes) return fail(CM_ERR_UNSUPPORTED, "this plan has no modulator");
    if (int rc_ = check_device(p->device, rgb, composite)) return rc_;
    const cm_plan_desc &d = p->desc;
    const int W = d.width, H = d.height, D = d.modulation_delay;
    const int wp = (W + 3) & ~3;
    if (H - 1 + 2 * D >= p->mod_n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's phase tables");
    return with_pitched_rows(rgb, n_frames * 3 * H, composite, n_frames * H, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        g.W = W;
        g.Wp = wp;
        g.H = H;
        g.in_frame_stride = 3LL * wp * H;
        g.in_plane_stride = (long long)wp * H;
        g.in_row_stride = wp;
        g.out_frame_stride = (long long)wp * H;
        g.out_row_stride = wp;
        set_first_frame(p, g, first_frame, p->mod_cycle);
        const int rows0 = (H + 1) / 2, rows1 = H / 2;
        g.calls_run0 = rows0 + D;
        const int calls_run1 = rows1 > 0 ? rows1 + D : 0;
        g.calls_per_frame = g.calls_run0 + calls_run1;
        g.runs_per_frame = rows1 > 0 ? 2 : 1;
        g.first_line[0] = 0;
        g.first_line[1] = 1;
        g.delay = D;
        g.total_calls = n_frames * g.calls_per_frame;
        return run_mod(p, g, (hipStream_t)stream);
    });
}

int cm_modulate_frames_u8(const cm_plan *p, const uint8_t *rgb8, uint8_t *composite8, int64_t n_frames, int64_t first_frame,
                          void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !rgb8 || !composite8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (!p->mod_fn && !p->sm_lanes) return fail(CM_ERR_UNSUPPORTED, "this plan has no modulator");
    if (int rc_ = check_device(p->device, rgb8, composite8)) return rc_;
    const cm_plan_desc &d = p->desc;
    const int W = d.width, H = d.height, D = d.modulation_delay;
    if (W % 16 != 0) return fail(CM_ERR_UNSUPPORTED, "the fused uint8 boundary of the encoders needs a width that is a multiple of 16");
    Geom g;
    std::memset(&g, 0, sizeof g);
    g.in = reinterpret_cast<const float *>(rgb8);         // strides below count bytes (U8 kernels)
    g.out = reinterpret_cast<float *>(composite8);
    g.W = W;
    g.Wp = W;
    g.H = H;
    g.in_frame_stride = 3LL * W * H;
    g.in_plane_stride = 0;
    g.in_row_stride = 3LL * W;
    g.out_frame_stride = (long long)W * H;
    g.out_row_stride = W;
    set_first_frame(p, g, first_frame, p->mod_cycle);
    const int rows0 = (H + 1) / 2, rows1 = H / 2;
    g.calls_run0 = rows0 + D;
    const int calls_run1 = rows1 > 0 ? rows1 + D : 0;
    g.calls_per_frame = g.calls_run0 + calls_run1;
    g.runs_per_frame = rows1 > 0 ? 2 : 1;
    g.first_line[0] = 0;
    g.first_line[1] = 1;
    g.delay = D;
    g.total_calls = n_frames * g.calls_per_frame;
    if (H - 1 + 2 * D >= p->mod_n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's phase tables");
    return run_mod(p, g, (hipStream_t)stream, true);
}

int cm_modulate_run(const cm_plan *p, const float *rgb, float *composite, int32_t n_calls, int32_t frame, int32_t first_line,
                    int32_t k0, void *stream) {
    if (!p || !rgb || !composite) return fail(CM_ERR_INVALID, "null argument");
    if (n_calls < 0 || frame < 0 || k0 < 0 || first_line < 0) return fail(CM_ERR_INVALID, "negative count / frame / line / k0");
    if (n_calls == 0) return CM_OK;
    if (!p->mod_fn && !p->sm_lanes) return fail(CM_ERR_UNSUPPORTED, "this plan has no modulator");
    if (int rc_ = check_device(p->device, rgb, composite)) return rc_;
    const cm_plan_desc &d = p->desc;
    const int W = d.width, wp = (W + 3) & ~3;
    if (first_line + 2 * (n_calls - 1) >= p->mod_n_lines) return fail(CM_ERR_INVALID, "line number beyond the plan's phase tables");
    return with_pitched_rows(rgb, 3LL * n_calls, composite, n_calls, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        g.W = W;
        g.Wp = wp;
        g.H = n_calls;
        g.in_plane_stride = wp;             // rows mode reads [call][plane][W]
        g.in_row_stride = 3LL * wp;
        g.out_row_stride = wp;
        g.rows_mode = 1;
        set_first_frame(p, g, frame, p->mod_cycle);
        g.calls_run0 = n_calls;
        g.calls_per_frame = n_calls;
        g.runs_per_frame = 1;
        g.first_line[0] = g.first_line[1] = first_line;
        g.k0 = k0;
        g.total_calls = n_calls;
        return run_mod(p, g, (hipStream_t)stream);
    });
}

// ---- D2-MAC style time-multiplex modem (cm_mac_kernels.h) -------------------------------------------------------------
struct cm_mac_plan {
    cm_mac_desc desc;
    float *fir[4] = {nullptr, nullptr, nullptr, nullptr};   // device copies of luma_in, chroma_in, line_out, line_in
    bool tuned = false;                                      // 720-sample rows <-> 1080-sample lines
    int device = 0;                                          // the device that was current in cm_mac_plan_create
};

namespace {
const cm_mac_fir *mac_fir(const cm_mac_desc &d, int i) {
    return i == 0 ? &d.luma_in : (i == 1 ? &d.chroma_in : (i == 2 ? &d.line_out : &d.line_in));
}
cm::MacFir mac_dev_fir(const cm_mac_plan *p, int i) {
    const cm_mac_fir &f = *mac_fir(p->desc, i);
    cm::MacFir r;
    r.h = p->fir[i];
    r.up = f.up;
    r.down = f.down;
    r.half_len = (f.n_taps - 1) / 2;
    r.stage = 0;
    return r;
}
int mac_launch(const cm_mac_plan *p, bool demod, const float *in, float *out, int n_frames, int height, int rows_mode,
               int first_line, int64_t first_frame, hipStream_t stream, bool u8 = false) {
    const cm_mac_desc *d = &p->desc;
    cm::MacArgs a;
    std::memset(&a, 0, sizeof a);
    if (int rc_ = check_device(p->device, in, out)) return rc_;
    a.in = in;
    a.out = out;
    a.n_frames = n_frames;
    a.H = height;
    a.rows_mode = rows_mode;
    a.first_line = first_line;
    a.first_frame = first_frame;
    a.averaging = d->averaging ? 1 : 0;
    a.line_shift = d->line_shift;
    a.even_first = d->even_first;
    a.odd_first = d->odd_first;
    const double scale = demod ? 2.0 : 1.0;     // resample_poly scales the filter by `up`
    a.c0 = (float)(scale * d->resample_fir[20]);
    for (int j = 0; j < 20; ++j) a.taps[j] = (float)(scale * d->resample_fir[2 * j + 1]);
    for (int i = 0; i < 9; ++i) a.m[i] = (float)(demod ? d->decode_matrix[i] : d->encode_matrix[i]);
    if (p->tuned && !u8) {
        long long blocks;
        if (rows_mode) blocks = (height + cm::kMacSegment - 1) / cm::kMacSegment;
        else blocks = (long long)n_frames * 2 * ((((height + 1) >> 1) + cm::kMacSegment - 1) / cm::kMacSegment);
        if (blocks <= 0) return CM_OK;
        if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
        if (demod) hipLaunchKernelGGL(cm::mac_demod_kernel, dim3((unsigned)blocks), dim3(cm::kMacThreads), 0, stream, a);
        else hipLaunchKernelGGL(cm::mac_mod_kernel, dim3((unsigned)blocks), dim3(cm::kMacThreads), 0, stream, a);
    } else {
        cm::MacGenArgs g;
        g.a = a;
        g.W = d->width;
        g.CW = d->line_width;
        g.luma_in = mac_dev_fir(p, 0);
        g.chroma_in = mac_dev_fir(p, 1);
        g.line_out = mac_dev_fir(p, 2);
        g.line_in = mac_dev_fir(p, 3);
        long long blocks = (long long)n_frames * height;             // encoder: one workgroup per call
        if (demod) {                                                 // decoder: segments of a field, like the tuned kernel
            if (rows_mode) blocks = (height + cm::kMacSegment - 1) / cm::kMacSegment;
            else blocks = (long long)n_frames * 2 * ((((height + 1) >> 1) + cm::kMacSegment - 1) / cm::kMacSegment);
        }
        if (blocks <= 0) return CM_OK;
        if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
        size_t lds_demod = sizeof(float) * (cm::kMacLine + cm::kMacChroma + 24 + 2 * cm::kMacLuma + (size_t)d->line_width);
        size_t lds_mod = sizeof(float) * (cm::kMacLine + cm::kMacLuma + cm::kMacChroma + (d->averaging ? 7 : 4) * (size_t)d->width);
        // the taps go to LDS while the workgroup stays within 48 KiB (cm_mac_kernels.h: mac_stage_taps)
        auto stage = [](cm::MacFir &f, size_t &lds) {
            const size_t bytes = f.h ? sizeof(float) * (2 * (size_t)f.half_len + 1) : 0;
            f.stage = bytes && lds + bytes <= 48 * 1024 ? 1 : 0;
            if (f.stage) lds += bytes;
        };
        g.luma_in.stage = g.chroma_in.stage = g.line_out.stage = g.line_in.stage = 0;
        if (demod) stage(g.line_in, lds_demod);
        else { stage(g.luma_in, lds_mod); stage(g.chroma_in, lds_mod); stage(g.line_out, lds_mod); }
        if (demod && u8) hipLaunchKernelGGL(cm::mac_demod_generic_kernel<true>, dim3((unsigned)blocks), dim3(cm::kMacThreads), lds_demod, stream, g);
        else if (demod) hipLaunchKernelGGL(cm::mac_demod_generic_kernel<false>, dim3((unsigned)blocks), dim3(cm::kMacThreads), lds_demod, stream, g);
        else if (u8) hipLaunchKernelGGL(cm::mac_mod_generic_kernel<true>, dim3((unsigned)blocks), dim3(cm::kMacThreads), lds_mod, stream, g);
        else hipLaunchKernelGGL(cm::mac_mod_generic_kernel<false>, dim3((unsigned)blocks), dim3(cm::kMacThreads), lds_mod, stream, g);
    }
    HIP_TRY(hipGetLastError(), CM_ERR_LAUNCH);
    return CM_OK;
}
int mac_check(const cm_mac_plan *p, const void *in, const void *out, long long n) {
    if (!p) return fail(CM_ERR_INVALID, "null argument");
    if (n == 0) return CM_OK;     // an empty batch may come with null buffers
    if (!in || !out) return fail(CM_ERR_INVALID, "null argument");
    if (p->tuned && (((unsigned long long)in | (unsigned long long)out) & 15)) return fail(CM_ERR_INVALID, "buffers must be 16-byte aligned");
    return CM_OK;
}
}  // namespace

extern "C" {
int cm_mac_plan_create(const cm_mac_desc *desc, cm_mac_plan **out) {
    if (!desc || !out) return fail(CM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->height <= 0 || desc->width <= 0 || desc->line_width <= 0) return fail(CM_ERR_INVALID, "width, height and line width must be positive");
    if (desc->width > 1920) return fail(CM_ERR_UNSUPPORTED, "MAC: rows of more than 1920 samples do not fit the encoder's LDS layout");
    if (desc->line_width > 4096) return fail(CM_ERR_UNSUPPORTED, "MAC: lines of more than 4096 samples are not supported");
    if (cm_device_count() <= 0) return fail(CM_ERR_NO_DEVICE, "no HIP device: the MAC path runs on the GPU only");
    cm_mac_plan *p = new cm_mac_plan();
    p->desc = *desc;
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return fail(CM_ERR_NO_DEVICE, "hipGetDevice failed");
    }
    p->tuned = desc->width == CM_MAC_LUMA_WIDTH && desc->line_width == CM_MAC_LINE_WIDTH;
    for (int i = 0; i < 4; ++i) {
        const cm_mac_fir &f = *mac_fir(*desc, i);
        if (f.up <= 0 || f.down <= 0) { cm_mac_plan_destroy(p); return fail(CM_ERR_INVALID, "MAC: resampling ratio must be positive"); }
        if (f.up == f.down) continue;
        const int max_rate = f.up > f.down ? f.up : f.down;
        if (!f.taps || f.n_taps != 2 * 10 * max_rate + 1) { cm_mac_plan_destroy(p); return fail(CM_ERR_INVALID, "MAC: resampling filter must have 2 * 10 * max(up, down) + 1 taps"); }
        std::vector<float> h(f.n_taps);
        for (int j = 0; j < f.n_taps; ++j) h[j] = (float)f.taps[j];
        if (hipMalloc((void **)&p->fir[i], h.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(p->fir[i], h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            cm_mac_plan_destroy(p);
            return fail(CM_ERR_LAUNCH, "device allocation / upload of a resampling filter failed");
        }
    }
    p->desc.luma_in.taps = p->desc.chroma_in.taps = p->desc.line_out.taps = p->desc.line_in.taps = nullptr;   // the caller's arrays are not kept
    *out = p;
    return CM_OK;
}
void cm_mac_plan_destroy(cm_mac_plan *p) {
    if (!p) return;
    for (int i = 0; i < 4; ++i)
        if (p->fir[i]) (void)hipFree(p->fir[i]);
    delete p;
}
int cm_mac_modulate_frames(const cm_mac_plan *p, const float *rgb, float *composite, int64_t n_frames, int64_t first_frame,
                           void *stream) {
    int rc = mac_check(p, rgb, composite, n_frames);
    if (rc) return rc;
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (n_frames == 0) return CM_OK;
    if (n_frames > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    return mac_launch(p, false, rgb, composite, (int)n_frames, p->desc.height, 0, 0, first_frame, (hipStream_t)stream);
}
int cm_mac_demodulate_frames(const cm_mac_plan *p, const float *composite, float *rgb, int64_t n_frames, int64_t first_frame,
                             void *stream) {
    int rc = mac_check(p, composite, rgb, n_frames);
    if (rc) return rc;
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (n_frames == 0) return CM_OK;
    if (n_frames > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    return mac_launch(p, true, composite, rgb, (int)n_frames, p->desc.height, 0, 0, first_frame, (hipStream_t)stream);
}
int cm_mac_modulate_frames_u8(const cm_mac_plan *p, const uint8_t *rgb8, uint8_t *composite8, int64_t n_frames, int64_t first_frame,
                              void *stream) {
    if (!p) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (n_frames == 0) return CM_OK;
    if (!rgb8 || !composite8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    return mac_launch(p, false, (const float *)rgb8, (float *)composite8, (int)n_frames, p->desc.height, 0, 0, first_frame, (hipStream_t)stream, true);
}
int cm_mac_demodulate_frames_u8(const cm_mac_plan *p, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames, int64_t first_frame,
                                void *stream) {
    if (!p) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (n_frames == 0) return CM_OK;
    if (!rgb8 || !composite8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    return mac_launch(p, true, (const float *)composite8, (float *)rgb8, (int)n_frames, p->desc.height, 0, 0, first_frame, (hipStream_t)stream, true);
}
int cm_mac_modulate_run(const cm_mac_plan *p, const float *rgb, float *composite, int32_t n_calls, int32_t frame,
                        int32_t first_line, int32_t k0, void *stream) {
    int rc = mac_check(p, rgb, composite, n_calls);
    if (rc) return rc;
    if (n_calls < 0 || frame < 0 || k0 < 0) return fail(CM_ERR_INVALID, "negative count / frame / k0");
    if (n_calls == 0) return CM_OK;
    return mac_launch(p, false, rgb, composite, 1, n_calls, 1, first_line, frame, (hipStream_t)stream);
}
int cm_mac_demodulate_run(const cm_mac_plan *p, const float *composite, float *rgb, int32_t n_calls, int32_t frame,
                          int32_t first_line, int32_t k0, void *stream) {
    int rc = mac_check(p, composite, rgb, n_calls);
    if (rc) return rc;
    if (n_calls < 0 || frame < 0 || k0 < 0) return fail(CM_ERR_INVALID, "negative count / frame / k0");
    if (n_calls == 0) return CM_OK;
    return mac_launch(p, true, composite, rgb, 1, n_calls, 1, first_line, frame, (hipStream_t)stream);
}
}

#endif  // CM_MAIN_PART
#if CM_AM_PART
// ---- amplitude-modulated line-sequential standards: Proto-SECAM, NIIR (cm_am_kernels.h) ------------------------------------
struct cm_am_plan {
    cm_am_desc desc;
    int device = 0;
    float *carrier = nullptr;          // {cos, sin}(n * carrier_phase_step), n < width
    ProtoDemodK<float> pd;
    ProtoModK<float> pm;
    NiirDemodK<float> nd;
    NiirDemodK<double> ndd;            // the decoder's float64 hue path (cm_am_stages.h: NiirHue)
    double *niir_syn = nullptr;        // [2][3 width]: the first lines' phase reference for cos / sin(n step) (cm_am_plan.h: build_niir_syn)
    NiirModK<float> nm;
    std::string demod_error, mod_error;
    // small batches: one wavefront per call (cm_am_scan_kernels.h); null where the plan's shape does not fit
    ScanProtoK *scan_pd = nullptr;
    ScanProtoModK *scan_pm = nullptr;
    ScanNiirK *scan_nd = nullptr;
    ScanNiirK64 *scan_nd64 = nullptr;   // the float64 hue path's constants
    ScanNiirModK *scan_nm = nullptr;
    int scan_pd_c1 = 0, scan_pm_c1 = 0, scan_nd_c1 = 0, scan_nm_c1 = 0;
    mutable std::atomic<int> small_batch{CM_SMALL_BATCH_AUTO};     // cm_am_plan_set_small_batch
};
#ifndef CM_AM_SCAN_MAX_CALLS
#define CM_AM_SCAN_MAX_CALLS 30000
#endif
#ifndef CM_AM_SCAN_MOD_MAX_CALLS
#define CM_AM_SCAN_MOD_MAX_CALLS 36000
#endif
// (NIIR: the decoder's five float64 decimators make a wave's row expensive - one 720 x 576 frame 142 us, 16 frames 74 us each, against 650 us for
// any batch up to 16 frames on the streaming pair: hand-over near 8 frames; the encoder is one packed scan - the scan kernel keeps up with the
// streaming one beyond 100 frames; profiles/r04_am_small_batch.txt)
#define CM_NIIR_SCAN_MAX_CALLS 4600
#define CM_NIIR_SCAN_MOD_MAX_CALLS 60000

namespace {
int am_geom(const cm_am_plan *p, int64_t first_frame, AmGeom &a) {
    a.line = am_line(p->desc);
    a.carrier = p->carrier;
    a.frame_base = (int)(first_frame % (2LL * a.line.frame_cycle));
    return CM_OK;
}
// the frames geometry of cm_demodulate_frames / cm_modulate_frames for a plan with `delay` lines of delay
void am_frames_geom(Geom &g, int W, int wp, int H, int D, int64_t n_frames) {
    g.W = W;
    g.Wp = wp;
    g.H = H;
    const int rows0 = (H + 1) / 2, rows1 = H / 2;
    g.calls_run0 = rows0 + D;
    const int calls_run1 = rows1 > 0 ? rows1 + D : 0;
    g.calls_per_frame = g.calls_run0 + calls_run1;
    g.runs_per_frame = rows1 > 0 ? 2 : 1;
    g.first_line[0] = 0;
    g.first_line[1] = 1;
    g.delay = D;
    g.total_calls = n_frames * g.calls_per_frame;
}
// NIIR: the main pass over every call plus the sparse pass over the calls that open a run (k0 == 0 in rows mode)
// ---- Proto-SECAM in small batches: the scan kernels' constants and launchers ---------------------------------------------------
static bool taps3_sparse(const float *h) {
    for (int q = 0; 3 * q < kAmTaps; ++q)
        if (q != kAmHalf && h[3 * q] != 0.f) return false;
    return true;
}
static int am_scan_chunk(int width, std::initializer_list<int> shifts3) {      // chunk of 1x-rate samples per lane, or 0
    int q = 0;
    for (int s : shifts3) q = std::max(q, (s + 2) / 3);
    if (q > kScanMaxShift) return 0;
    for (int c : {12, 16})
        if (width + q <= 64 * c) return c;
    return 0;
}
void make_scan_proto(cm_am_plan *p) {
    const cm_am_desc &d = p->desc;
    if (d.kind != CM_AM_PROTO_SECAM) return;
    if (p->demod_error.empty()) {
        const int c1 = am_scan_chunk(d.width, {d.bandpass_up.shift, d.bandstop_up.shift, d.lowpass_up.shift});
        if (c1) {
            ScanProtoK k;
            std::memset(&k, 0, sizeof k);
            const ProtoDemodK<float> &m = p->pd;
            k.width = d.width; k.c1 = c1;
            for (int i = 0; i < kAmTaps; ++i) k.h[i] = m.taps.h[i];
            k.sparse_taps = taps3_sparse(k.h) ? 1 : 0;
            fill_scan_filter(d.bandpass_up, m.ext.na1, m.ext.na2, m.ext.b1, m.ext.b2, 3 * c1, k.ext);
            fill_scan_filter(d.bandstop_up, m.rem.na1, m.rem.na2, m.rem.b1, m.rem.b2, 3 * c1, k.rem);
            fill_scan_filter(d.lowpass_up, m.post.na1, m.post.na2, m.post.b1, m.post.b2, 3 * c1, k.post);
            k.chroma_gain = m.chroma_gain; k.luma_gain = m.luma_gain;
            for (int i = 0; i < 9; ++i) k.m[i] = m.m[i / 3][i % 3];
            if (hipMalloc((void **)&p->scan_pd, sizeof k) == hipSuccess && hipMemcpy(p->scan_pd, &k, sizeof k, hipMemcpyHostToDevice) == hipSuccess)
                p->scan_pd_c1 = c1;
            else p->scan_pd = nullptr;
        }
    }
    if (p->mod_error.empty() && d.precorrect.shift <= kScanMaxShift) {
        const int c1 = am_scan_chunk(d.width + d.precorrect.shift, {d.premod_luma_filter ? d.bandstop_up.shift : 0});
        if (c1) {
            ScanProtoModK k;
            std::memset(&k, 0, sizeof k);
            const ProtoModK<float> &m = p->pm;
            k.width = d.width; k.c1 = c1; k.luma_filter = m.luma_filter; k.averaging = d.averaging ? 1 : 0;
            for (int i = 0; i < kAmTaps; ++i) k.h[i] = m.taps.h[i];
            k.sparse_taps = taps3_sparse(k.h) ? 1 : 0;
            fill_scan_filter(d.precorrect, m.pre.na1, m.pre.na2, m.pre.b1, m.pre.b2, c1, k.pre);
            fill_scan_filter(d.bandstop_up, m.rem.na1, m.rem.na2, m.rem.b1, m.rem.b2, 3 * c1, k.rem);
            k.pre_gain = m.pre_gain; k.luma_gain = m.luma_gain;
            for (int i = 0; i < 9; ++i) k.e[i] = m.e[i / 3][i % 3];
            if (hipMalloc((void **)&p->scan_pm, sizeof k) == hipSuccess && hipMemcpy(p->scan_pm, &k, sizeof k, hipMemcpyHostToDevice) == hipSuccess)
                p->scan_pm_c1 = c1;
            else p->scan_pm = nullptr;
        }
    }
}
extern "C++" {
template <int C1, int NW, bool U8>
int launch_scan_proto_demod(const cm_am_plan *p, const Geom &g, const AmGeom &a, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_proto_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)proto_demod_scan_kernel<C1, NW, U8>, p->device, lds, "the Proto-SECAM decoder's scan kernel")) return rc;
    const long long blocks = (g.total_calls + (NW - 1) - 1) / (NW - 1);
    hipLaunchKernelGGL((proto_demod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, a, p->scan_pd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("proto_demod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <int C1, int NW, bool U8>
int launch_scan_proto_mod(const cm_am_plan *p, const Geom &g, const AmGeom &a, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_proto_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)proto_mod_scan_kernel<C1, NW, U8>, p->device, lds, "the Proto-SECAM encoder's scan kernel")) return rc;
    const long long blocks = (g.total_calls + NW - 1) / NW;
    hipLaunchKernelGGL((proto_mod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, a, p->scan_pm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("proto_mod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
}  // extern "C++"
// 1: launched on the scan kernel (rc holds the status); 0: the streaming kernel's turn
static bool am_scan_wanted(const cm_am_plan *p, const void *scan, long long calls, long long max_calls, int &rc) {
    rc = CM_OK;
    const int mode = p->small_batch;
    if (scan && (mode == CM_SMALL_BATCH_SCAN || (mode == CM_SMALL_BATCH_AUTO && calls <= max_calls))) return true;
    if (mode == CM_SMALL_BATCH_SCAN) rc = fail(CM_ERR_UNSUPPORTED, "the scan kernel does not serve this plan / this direction");
    return false;
}
void make_scan_niir(cm_am_plan *p) {
    const cm_am_desc &d = p->desc;
    if (d.kind != CM_AM_NIIR) return;
    if (p->demod_error.empty()) {
        const int c1 = am_scan_chunk(d.width, {d.bandpass_up.shift, d.lowpass_up.shift});
        if (c1) {
            ScanNiirK k;
            std::memset(&k, 0, sizeof k);
            const NiirDemodK<float> &m = p->nd;
            k.width = d.width; k.c1 = c1;
            for (int i = 0; i < kAmTaps; ++i) k.h[i] = m.taps.h[i];
            k.sparse_taps = taps3_sparse(k.h) ? 1 : 0;
            fill_scan_filter(d.bandpass_up, m.bp.na1, m.bp.na2, m.bp.b1, m.bp.b2, 3 * c1, k.bp);
            fill_scan_filter(d.lowpass_up, m.lp.na1, m.lp.na2, m.lp.b1, m.lp.b2, 3 * c1, k.lp);
            k.c_pm = m.c_pm; k.g_b = m.g_b; k.sat_gain = m.sat_gain; k.alt_scale = m.alt_scale; k.third = m.third;
            for (int i = 0; i < 9; ++i) k.m[i] = m.m[i / 3][i % 3];
            if (hipMalloc((void **)&p->scan_nd, sizeof k) == hipSuccess && hipMemcpy(p->scan_nd, &k, sizeof k, hipMemcpyHostToDevice) == hipSuccess)
                p->scan_nd_c1 = c1;
            else p->scan_nd = nullptr;
            if (p->scan_nd) {      // the hue path's float64 constants; without them the plan has no scan decoder
                const NiirDemodK<double> &md = p->ndd;
                ScanNiirK64 k64;
                std::memset(&k64, 0, sizeof k64);
                for (int i = 0; i < kAmTaps; ++i) k64.h[i] = md.taps.h[i];
                fill_scan_filter(d.bandpass_up, md.bp.na1, md.bp.na2, md.bp.b1, md.bp.b2, 3 * c1, k64.bp, 1e-20);
                fill_scan_filter(d.lowpass_up, md.lp.na1, md.lp.na2, md.lp.b1, md.lp.b2, 3 * c1, k64.lp, 1e-20);
                k64.c_pm = md.c_pm;
                k64.alt_scale = md.alt_scale;
                if (hipMalloc((void **)&p->scan_nd64, sizeof k64) != hipSuccess || hipMemcpy(p->scan_nd64, &k64, sizeof k64, hipMemcpyHostToDevice) != hipSuccess) {
                    p->scan_nd64 = nullptr;
                    (void)hipFree(p->scan_nd);
                    p->scan_nd = nullptr;
                    p->scan_nd_c1 = 0;
                }
            }
        }
    }
    if (p->mod_error.empty() && d.precorrect.shift <= kScanMaxShift) {
        int c1 = 0;
        for (int c : {12, 16, 24, 32})
            if (d.width + d.precorrect.shift <= 64 * c) { c1 = c; break; }
        if (c1) {
            ScanNiirModK k;
            std::memset(&k, 0, sizeof k);
            const NiirModK<float> &m = p->nm;
            k.width = d.width; k.c1 = c1; k.averaging = d.averaging ? 1 : 0;
            fill_scan_filter(d.precorrect, m.pre.na1, m.pre.na2, m.pre.b1, m.pre.b2, c1, k.pre);
            k.pre_gain = m.pre_gain;
            for (int i = 0; i < 9; ++i) k.e[i] = m.e[i / 3][i % 3];
            for (int i = 0; i < 6; ++i) k.ed[i] = m.ed[i];
            if (hipMalloc((void **)&p->scan_nm, sizeof k) == hipSuccess && hipMemcpy(p->scan_nm, &k, sizeof k, hipMemcpyHostToDevice) == hipSuccess)
                p->scan_nm_c1 = c1;
            else p->scan_nm = nullptr;
        }
    }
}
extern "C++" {
template <int C1, int NW, bool U8>
int launch_scan_niir_demod(const cm_am_plan *p, const Geom &g, const AmGeom &a, bool strip, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_niir_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)niir_demod_scan_kernel<C1, NW, U8>, p->device, lds, "the NIIR decoder's scan kernel")) return rc;
    const long long blocks = (g.total_calls + (NW - 1) - 1) / (NW - 1);
    hipLaunchKernelGGL((niir_demod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, a, p->scan_nd, p->scan_nd64, p->niir_syn,
                       p->desc.line_phase_shift, p->desc.bandpass_phase_shift, strip ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("niir_demod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <int C1, int NW, bool U8>
int launch_scan_niir_mod(const cm_am_plan *p, const Geom &g, const AmGeom &a, const float *noise, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_mod_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)niir_mod_scan_kernel<C1, NW, U8>, p->device, lds, "the NIIR encoder's scan kernel")) return rc;
    const long long blocks = (g.total_calls + NW - 1) / NW;
    hipLaunchKernelGGL((niir_mod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, a, p->scan_nm, noise);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("niir_mod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <bool U8>
int scan_niir_mod_as(const cm_am_plan *p, const Geom &g, const AmGeom &a, const float *noise, hipStream_t stream) {
    switch (p->scan_nm_c1) {
        case 12: return launch_scan_niir_mod<12, 4, U8>(p, g, a, noise, stream);
        case 16: return launch_scan_niir_mod<16, 4, U8>(p, g, a, noise, stream);
        case 24: return launch_scan_niir_mod<24, 4, U8>(p, g, a, noise, stream);
        default: return launch_scan_niir_mod<32, 4, U8>(p, g, a, noise, stream);
    }
}
}  // extern "C++"
int niir_launch_demod(const cm_am_plan *p, Geom g, int64_t first_frame, hipStream_t stream, bool strip, bool u8 = false) {
    NiirDemodArgs a;
    am_geom(p, first_frame, a.a);
    a.k = p->nd;
    a.kd = p->ndd;
    a.syn = p->niir_syn;
    a.line_phase_shift = p->desc.line_phase_shift;
    a.bandpass_phase_shift = p->desc.bandpass_phase_shift;
    a.carrier_phase_step = p->desc.carrier_phase_step;
    a.strip = strip ? 1 : 0;
    const bool with_first = g.k0 == 0;
    {   // small batches: one wavefront per call, the first lines of the runs in the same pass (cm_am_scan_kernels.h)
        int rc;
        if (am_scan_wanted(p, p->scan_nd, g.total_calls, CM_NIIR_SCAN_MAX_CALLS, rc)) {
            if (g.total_calls <= 0) return CM_OK;
            if (p->scan_nd_c1 == 12) return u8 ? launch_scan_niir_demod<12, 3, true>(p, g, a.a, strip, stream) : launch_scan_niir_demod<12, 3, false>(p, g, a.a, strip, stream);
            return u8 ? launch_scan_niir_demod<16, 2, true>(p, g, a.a, strip, stream) : launch_scan_niir_demod<16, 2, false>(p, g, a.a, strip, stream);
        }
        if (rc) return rc;
    }
    g.skip_first = 1;
    long long blocks = (g.total_calls + 62) / 63;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    if (g.rows_mode && g.total_calls == 1 && with_first) blocks = 0;      // a lone first call needs no main pass
    {   // the wave pair: main pass and sparse first-line pass in one launch (cm_am_kernels.h: niir_demod_pair_kernel)
        NiirPairArgs pa;
        a.g = g;
        pa.m = a;
        pa.gf = g;
        pa.n_first = 0;
        if (with_first) {
            pa.gf.sparse = 1;
            pa.gf.skip_first = 0;
            pa.gf.total_calls = g.rows_mode ? 1 : (g.total_calls / g.calls_per_frame) * g.runs_per_frame;
            pa.n_first = (int)((pa.gf.total_calls + 63) / 64);
        }
        if (blocks + pa.n_first <= 0) return CM_OK;
        const int lat = 2 * kAmHalf + 1 + p->nd.gb.q + p->nd.gl.q;
        if (u8) {
            const size_t lds = sizeof(float) * (size_t)niir_pair_lds_floats<true>(lat, p->nd.gl.q);
            hipLaunchKernelGGL(niir_demod_pair_kernel<true>, dim3((int)blocks + pa.n_first), dim3(128), lds, stream, pa);
        } else {
            const size_t lds = sizeof(float) * (size_t)niir_pair_lds_floats<false>(lat, p->nd.gl.q);
            hipLaunchKernelGGL(niir_demod_pair_kernel<false>, dim3((int)blocks + pa.n_first), dim3(128), lds, stream, pa);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("niir_demod_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
int am_launch_demod(const cm_am_plan *p, Geom g, int64_t first_frame, hipStream_t stream, bool u8 = false) {
    if (!p->demod_error.empty()) return fail(CM_ERR_UNSUPPORTED, p->demod_error);
    if (p->desc.kind == CM_AM_NIIR) return niir_launch_demod(p, g, first_frame, stream, p->desc.strip_chroma != 0, u8);
    long long blocks = (g.total_calls + 62) / 63;
    if (blocks <= 0) return CM_OK;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    {
        int rc;
        if (am_scan_wanted(p, p->scan_pd, g.total_calls, CM_AM_SCAN_MAX_CALLS, rc)) {
            AmGeom ag;
            am_geom(p, first_frame, ag);
            if (p->scan_pd_c1 == 12) return u8 ? launch_scan_proto_demod<12, 4, true>(p, g, ag, stream) : launch_scan_proto_demod<12, 4, false>(p, g, ag, stream);
            return u8 ? launch_scan_proto_demod<16, 4, true>(p, g, ag, stream) : launch_scan_proto_demod<16, 4, false>(p, g, ag, stream);
        }
        if (rc) return rc;
    }
    ProtoDemodArgs a;
    a.g = g;
    am_geom(p, first_frame, a.a);
    a.k = p->pd;
    {
        const int dly = ProtoDemod<float>::lat_chroma(p->pd) - ProtoDemod<float>::lat_luma(p->pd);
        if (u8) hipLaunchKernelGGL(proto_demod_pair_kernel<true>, dim3((int)blocks), dim3(128), sizeof(float) * (size_t)proto_pair_lds_floats<true>(dly), stream, a);
        else hipLaunchKernelGGL(proto_demod_pair_kernel<false>, dim3((int)blocks), dim3(128), sizeof(float) * (size_t)proto_pair_lds_floats<false>(dly), stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("proto_demod_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
int am_launch_mod(const cm_am_plan *p, Geom g, int64_t first_frame, hipStream_t stream, const float *noise = nullptr, bool u8 = false) {
    if (!p->mod_error.empty()) return fail(CM_ERR_UNSUPPORTED, p->mod_error);
    const int depth = p->desc.averaging ? 1 : 0;
    long long blocks = (g.total_calls + (64 - depth) - 1) / (64 - depth);
    if (blocks <= 0) return CM_OK;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    if (p->desc.kind == CM_AM_NIIR) {
        {
            int rc;
            if (am_scan_wanted(p, p->scan_nm, g.total_calls, CM_NIIR_SCAN_MOD_MAX_CALLS, rc)) {
                AmGeom ag;
                am_geom(p, first_frame, ag);
                return u8 ? scan_niir_mod_as<true>(p, g, ag, noise, stream) : scan_niir_mod_as<false>(p, g, ag, noise, stream);
            }
            if (rc) return rc;
        }
        NiirModArgs a;
        a.g = g;
        am_geom(p, first_frame, a.a);
        a.k = p->nm;
        a.noise = noise;
        // the luma delay ring in the smallest power of two above the pre-correction shift (plan creation checked s_c < kAmRing)
        auto launch = [&](auto ring_tag) {
            constexpr int RING = decltype(ring_tag)::value;
            if (u8) {
                if (depth) hipLaunchKernelGGL((niir_mod_kernel<1, true, RING>), dim3((int)blocks), dim3(64), 0, stream, a);
                else hipLaunchKernelGGL((niir_mod_kernel<0, true, RING>), dim3((int)blocks), dim3(64), 0, stream, a);
            } else if (depth) hipLaunchKernelGGL((niir_mod_kernel<1, false, RING>), dim3((int)blocks), dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((niir_mod_kernel<0, false, RING>), dim3((int)blocks), dim3(64), 0, stream, a);
        };
        if (p->nm.s_c < 8) launch(std::integral_constant<int, 8>());
        else if (p->nm.s_c < 16) launch(std::integral_constant<int, 16>());
        else launch(std::integral_constant<int, 32>());
    } else {
        if (noise) return fail(CM_ERR_INVALID, "noise planes are a NIIR encoder input (niir.py:45-46)");
        {
            int rc;
            if (am_scan_wanted(p, p->scan_pm, g.total_calls, CM_AM_SCAN_MOD_MAX_CALLS, rc)) {
                AmGeom ag;
                am_geom(p, first_frame, ag);
                if (p->scan_pm_c1 == 12) return u8 ? launch_scan_proto_mod<12, 4, true>(p, g, ag, stream) : launch_scan_proto_mod<12, 4, false>(p, g, ag, stream);
                return u8 ? launch_scan_proto_mod<16, 4, true>(p, g, ag, stream) : launch_scan_proto_mod<16, 4, false>(p, g, ag, stream);
            }
            if (rc) return rc;
        }
        ProtoModArgs a;
        a.g = g;
        am_geom(p, first_frame, a.a);
        a.k = p->pm;
        a.averaging = depth;
        {
            const int lat_y = ProtoMod<float>::lat_luma(p->pm), lat_c = ProtoMod<float>::lat_chroma(p->pm);
            const int dly = lat_y > lat_c ? lat_y - lat_c : lat_c - lat_y;
            if (u8) {
                const size_t lds = sizeof(float) * (size_t)proto_mod_pair_lds_floats<true>(dly);
                if (depth) hipLaunchKernelGGL((proto_mod_pair_kernel<1, true>), dim3((int)blocks), dim3(128), lds, stream, a);
                else hipLaunchKernelGGL((proto_mod_pair_kernel<0, true>), dim3((int)blocks), dim3(128), lds, stream, a);
            } else {
                const size_t lds = sizeof(float) * (size_t)proto_mod_pair_lds_floats<false>(dly);
                if (depth) hipLaunchKernelGGL((proto_mod_pair_kernel<1, false>), dim3((int)blocks), dim3(128), lds, stream, a);
                else hipLaunchKernelGGL((proto_mod_pair_kernel<0, false>), dim3((int)blocks), dim3(128), lds, stream, a);
            }
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("cm_am modulator launch: ") + hipGetErrorString(e));
    return CM_OK;
}
}  // namespace

extern "C" {
int cm_am_plan_create(const cm_am_desc *desc, cm_am_plan **out) {
    if (!desc || !out) return fail(CM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->abi_version != CM_ABI_VERSION) return fail(CM_ERR_INVALID, "descriptor ABI version mismatch");
    if (desc->width < 4) return fail(CM_ERR_UNSUPPORTED, "width must be at least 4");
    if (desc->height < 1) return fail(CM_ERR_INVALID, "height must be positive");
    if (desc->kind != CM_AM_PROTO_SECAM && desc->kind != CM_AM_NIIR) return fail(CM_ERR_INVALID, "unknown cm_am_kind");
    if (desc->frame_cycle < 1) return fail(CM_ERR_INVALID, "frame_cycle must be positive");
    if (cm_device_count() < 1) return fail(CM_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    cm_am_plan *p = new cm_am_plan;
    p->desc = *desc;
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return fail(CM_ERR_NO_DEVICE, "hipGetDevice failed");
    }
    std::string err;
    if (desc->kind == CM_AM_NIIR) {
        if (!build_niir_demod_k<float>(*desc, p->nd, err) || !build_niir_demod_k<double>(*desc, p->ndd, err)) p->demod_error = err;
        else if (p->nd.gl.q >= kNiirRing) p->demod_error = "decoder: the low-pass delay does not fit the band-pass ring";
        if (!build_niir_mod_k<float>(*desc, p->nm, err)) p->mod_error = err;
        else if (p->nm.s_c >= kAmRing) p->mod_error = "encoder: the pre-correction shift does not fit the luma delay ring";
    } else if (!build_proto_demod_k<float>(*desc, p->pd, err)) p->demod_error = err;
    else {
        const int dly = ProtoDemod<float>::lat_chroma(p->pd) - ProtoDemod<float>::lat_luma(p->pd);
        if (dly < 0 || dly > kAmRing) p->demod_error = "decoder: the luma delay does not fit the delay ring";
    }
    if (desc->kind == CM_AM_NIIR) {
    } else if (!build_proto_mod_k<float>(*desc, p->pm, err)) p->mod_error = err;
    else {
        const int ly = ProtoMod<float>::lat_luma(p->pm), lc = ProtoMod<float>::lat_chroma(p->pm);
        const int dly = ly > lc ? ly - lc : lc - ly;
        if (dly >= kAmRing) p->mod_error = "encoder: the path delay does not fit the delay ring";
    }
    if (!p->demod_error.empty() && !p->mod_error.empty()) {
        err = p->demod_error;
        delete p;
        return fail(CM_ERR_UNSUPPORTED, err);
    }
    std::vector<float> car(2 * (size_t)desc->width);
    for (int n = 0; n < desc->width; ++n) {
        const double ph = (double)n * desc->carrier_phase_step;
        car[2 * (size_t)n] = (float)std::cos(ph);
        car[2 * (size_t)n + 1] = (float)std::sin(ph);
    }
    if (hipMalloc((void **)&p->carrier, car.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->carrier, car.data(), car.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        cm_am_plan_destroy(p);
        return fail(CM_ERR_NO_DEVICE, "device allocation / upload of the carrier table failed");
    }
    if (desc->kind == CM_AM_NIIR && p->demod_error.empty()) {
        std::vector<double> syn;
        if (!build_niir_syn(*desc, syn, err)) p->demod_error = err;
        else if (hipMalloc((void **)&p->niir_syn, syn.size() * sizeof(double)) != hipSuccess ||
                 hipMemcpy(p->niir_syn, syn.data(), syn.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            cm_am_plan_destroy(p);
            return fail(CM_ERR_NO_DEVICE, "device allocation / upload of the NIIR reference tables failed");
        }
    }
    make_scan_proto(p);
    make_scan_niir(p);
    *out = p;
    return CM_OK;
}
void cm_am_plan_destroy(cm_am_plan *p) {
    if (!p) return;
    if (p->carrier) (void)hipFree(p->carrier);
    if (p->niir_syn) (void)hipFree(p->niir_syn);
    if (p->scan_pd) (void)hipFree(p->scan_pd);
    if (p->scan_pm) (void)hipFree(p->scan_pm);
    if (p->scan_nd) (void)hipFree(p->scan_nd);
    if (p->scan_nd64) (void)hipFree(p->scan_nd64);
    if (p->scan_nm) (void)hipFree(p->scan_nm);
    delete p;
}
int cm_am_plan_set_small_batch(const cm_am_plan *p, int32_t mode) {
    if (!p) return fail(CM_ERR_INVALID, "null argument");
    if (mode < CM_SMALL_BATCH_AUTO || mode > CM_SMALL_BATCH_SCAN) return fail(CM_ERR_INVALID, "unknown small-batch mode");
    if (mode == CM_SMALL_BATCH_SEGMENTS) return fail(CM_ERR_UNSUPPORTED, "the Proto-SECAM / NIIR kernels have no row segments");
    if (mode == CM_SMALL_BATCH_SCAN && !p->scan_pd && !p->scan_pm && !p->scan_nd && !p->scan_nm) return fail(CM_ERR_UNSUPPORTED, "the scan kernels do not serve this plan");
    p->small_batch = mode;
    return CM_OK;
}
int cm_am_demodulate_frames(const cm_am_plan *p, const float *composite, float *rgb, int64_t n_frames, int64_t first_frame, void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !composite || !rgb) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(p->device, composite, rgb)) return rc_;
    const int W = p->desc.width, H = p->desc.height, wp = (W + 3) & ~3;
    return with_pitched_rows(composite, n_frames * H, rgb, n_frames * 3 * H, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        am_frames_geom(g, W, wp, H, 0, n_frames);
        g.in_frame_stride = (long long)wp * H;
        g.in_row_stride = wp;
        g.out_plane_stride = (long long)wp * H;
        g.out_frame_stride = 3LL * wp * H;
        g.out_row_stride = wp;
        return am_launch_demod(p, g, first_frame, (hipStream_t)stream);
    });
}
static int am_modulate_frames_core(const cm_am_plan *p, const float *rgb, const float *noise, float *composite, int64_t n_frames,
                                   int64_t first_frame, void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !rgb || !composite) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(p->device, rgb, composite)) return rc_;
    const int W = p->desc.width, H = p->desc.height, wp = (W + 3) & ~3, D = p->desc.averaging ? 1 : 0;
    return with_pitched_rows(rgb, n_frames * 3 * H, composite, n_frames * H, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        am_frames_geom(g, W, wp, H, D, n_frames);
        g.in_frame_stride = 3LL * wp * H;
        g.in_plane_stride = (long long)wp * H;
        g.in_row_stride = wp;
        g.out_frame_stride = (long long)wp * H;
        g.out_row_stride = wp;
        return am_launch_mod(p, g, first_frame, (hipStream_t)stream, noise);
    });
}
int cm_am_modulate_frames(const cm_am_plan *p, const float *rgb, float *composite, int64_t n_frames, int64_t first_frame, void *stream) {
    return am_modulate_frames_core(p, rgb, nullptr, composite, n_frames, first_frame, stream);
}
int cm_am_modulate_frames_noise(const cm_am_plan *p, const float *rgb, const float *noise, float *composite, int64_t n_frames,
                                int64_t first_frame, void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!noise) return fail(CM_ERR_INVALID, "null argument");
    return am_modulate_frames_core(p, rgb, noise, composite, n_frames, first_frame, stream);
}
// the ImageModem byte boundary fused into the kernels (image.py:27-56, 58-84), as cm_demodulate_frames_u8 / cm_modulate_frames_u8
int cm_am_demodulate_frames_u8(const cm_am_plan *p, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames, int64_t first_frame, void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !composite8 || !rgb8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(p->device, composite8, rgb8)) return rc_;
    const int W = p->desc.width, H = p->desc.height;
    if (W % 4 != 0) return fail(CM_ERR_UNSUPPORTED, "the fused uint8 boundary needs a width that is a multiple of 4");
    Geom g;
    std::memset(&g, 0, sizeof g);
    g.in = reinterpret_cast<const float *>(composite8);     // strides below count bytes
    g.out = reinterpret_cast<float *>(rgb8);
    am_frames_geom(g, W, W, H, 0, n_frames);
    g.in_frame_stride = (long long)W * H;
    g.in_row_stride = W;
    g.out_plane_stride = 0;
    g.out_frame_stride = 3LL * W * H;
    g.out_row_stride = 3LL * W;
    return am_launch_demod(p, g, first_frame, (hipStream_t)stream, true);
}
int cm_am_modulate_frames_u8(const cm_am_plan *p, const uint8_t *rgb8, uint8_t *composite8, int64_t n_frames, int64_t first_frame, void *stream) {
    if (p && n_frames == 0) return CM_OK;
    if (!p || !rgb8 || !composite8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(p->device, rgb8, composite8)) return rc_;
    const int W = p->desc.width, H = p->desc.height, D = p->desc.averaging ? 1 : 0;
    if (W % 16 != 0) return fail(CM_ERR_UNSUPPORTED, "the fused uint8 boundary of the encoders needs a width that is a multiple of 16");
    if (H < 2 * D) return fail(CM_ERR_INVALID, "the image has too few rows for the modulation delay");
    Geom g;
    std::memset(&g, 0, sizeof g);
    g.in = reinterpret_cast<const float *>(rgb8);            // strides below count bytes
    g.out = reinterpret_cast<float *>(composite8);
    am_frames_geom(g, W, W, H, D, n_frames);
    g.in_frame_stride = 3LL * W * H;
    g.in_plane_stride = 0;
    g.in_row_stride = 3LL * W;
    g.out_frame_stride = (long long)W * H;
    g.out_row_stride = W;
    return am_launch_mod(p, g, first_frame, (hipStream_t)stream, nullptr, true);
}
int cm_am_demodulate_run(const cm_am_plan *p, const float *composite, float *rgb, int32_t n_calls, int32_t frame, int32_t first_line,
                         int32_t k0, void *stream) {
    if (!p || !composite || !rgb) return fail(CM_ERR_INVALID, "null argument");
    if (n_calls < 0 || frame < 0 || k0 < 0) return fail(CM_ERR_INVALID, "negative count / frame / k0");
    if (n_calls == 0) return CM_OK;
    if (int rc_ = check_device(p->device, composite, rgb)) return rc_;
    const int W = p->desc.width, wp = (W + 3) & ~3;
    return with_pitched_rows(composite, n_calls, rgb, 3LL * n_calls, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        g.W = W;
        g.Wp = wp;
        g.H = n_calls;
        g.rows_mode = 1;
        g.calls_run0 = g.calls_per_frame = n_calls;
        g.runs_per_frame = 1;
        g.first_line[0] = g.first_line[1] = first_line;
        g.k0 = k0;
        g.total_calls = n_calls;
        g.out_plane_stride = wp;            // rows mode writes [call][plane][W]
        g.out_row_stride = 3LL * wp;
        return am_launch_demod(p, g, frame, (hipStream_t)stream);
    });
}
static int am_modulate_run_core(const cm_am_plan *p, const float *rgb, const float *noise, float *composite, int32_t n_calls, int32_t frame,
                                int32_t first_line, int32_t k0, void *stream) {
    if (!p || !rgb || !composite) return fail(CM_ERR_INVALID, "null argument");
    if (n_calls < 0 || frame < 0 || k0 < 0) return fail(CM_ERR_INVALID, "negative count / frame / k0");
    if (n_calls == 0) return CM_OK;
    if (int rc_ = check_device(p->device, rgb, composite)) return rc_;
    const int W = p->desc.width, wp = (W + 3) & ~3;
    return with_pitched_rows(rgb, 3LL * n_calls, composite, n_calls, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.out = out;
        g.W = W;
        g.Wp = wp;
        g.H = n_calls;
        g.in_plane_stride = wp;             // rows mode reads [call][plane][W]
        g.in_row_stride = 3LL * wp;
        g.out_row_stride = wp;
        g.rows_mode = 1;
        g.calls_run0 = g.calls_per_frame = n_calls;
        g.runs_per_frame = 1;
        g.first_line[0] = g.first_line[1] = first_line;
        g.k0 = k0;
        g.total_calls = n_calls;
        return am_launch_mod(p, g, frame, (hipStream_t)stream, noise);
    });
}
int cm_am_modulate_run(const cm_am_plan *p, const float *rgb, float *composite, int32_t n_calls, int32_t frame, int32_t first_line,
                       int32_t k0, void *stream) {
    return am_modulate_run_core(p, rgb, nullptr, composite, n_calls, frame, first_line, k0, stream);
}
int cm_am_modulate_run_noise(const cm_am_plan *p, const float *rgb, const float *noise, float *composite, int32_t n_calls, int32_t frame,
                             int32_t first_line, int32_t k0, void *stream) {
    if (n_calls == 0) return CM_OK;
    if (!noise) return fail(CM_ERR_INVALID, "null argument");
    return am_modulate_run_core(p, rgb, noise, composite, n_calls, frame, first_line, k0, stream);
}
}  // extern "C"

#endif  // CM_AM_PART
#if CM_MAIN_PART
// ---- SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem (cm_wrap_kernels.h) ------------------------------
extern "C++" {
namespace {
template <int NP, int SP, bool U8, bool RT, bool MINAVG, bool NOTCH>
int launch_wrap_back_i(const WrapBackArgs<NP> &a, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL((comb_wrap_back_kernel<NP, SP, U8, RT, MINAVG, NOTCH>), dim3(blocks), dim3(64), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("comb_wrap_back_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <int NP, int SP, bool U8, bool RT>
int launch_wrap_back(const Geom &g, const cm_plan *backend, const cm_comb_wrap_desc &w, hipStream_t stream) {
    WrapBackArgs<NP> a;
    std::memset(&a, 0, sizeof a);
    a.g = g;
    a.k = *reinterpret_cast<const ModK<float, NP> *>(backend->mod_k.data());
    double g_n = 0.0;
    std::string err;
    if (!convert_sos_optional<float, 1>(w.notch, FORM_SYM, a.notch, g_n, err, "notch")) return fail(CM_ERR_UNSUPPORTED, err);
    a.notch_gain = w.notch.n_sections ? (float)g_n : 0.f;
    for (int i = 0; i < 9; ++i) a.m[i] = (float)w.matrix[i];
    a.own_delay = w.own_delay ? 1 : 0;
    a.minavg = w.minavg;
    a.strip = w.strip_chroma ? 1 : 0;
    const long long blocks = (g.total_calls + 62) / 63;
    if (blocks <= 0) return CM_OK;
    if (blocks > 0x7fffffffLL) return fail(CM_ERR_INVALID, "batch too large for one launch");
    const bool notch = a.notch_gain != 0.f;
    if (a.minavg == 1) return notch ? launch_wrap_back_i<NP, SP, U8, RT, true, true>(a, (int)blocks, stream)
                               : launch_wrap_back_i<NP, SP, U8, RT, true, false>(a, (int)blocks, stream);
    return notch ? launch_wrap_back_i<NP, SP, U8, RT, false, true>(a, (int)blocks, stream)
                 : launch_wrap_back_i<NP, SP, U8, RT, false, false>(a, (int)blocks, stream);
}
template <bool U8>
int wrap_back_scan(const Geom &g, const cm_plan *backend, const cm_comb_wrap_desc &w, hipStream_t stream) {
    ScanWrapArgs a;
    std::memset(&a, 0, sizeof a);
    SosK<float, 1> notch;
    double g_n = 0.0;
    std::string err;
    if (!convert_sos_optional<float, 1>(w.notch, FORM_SYM, notch, g_n, err, "notch")) return fail(CM_ERR_UNSUPPORTED, err);
    if (w.notch.n_sections) {
        ScanFilter f;
        fill_scan_filter(w.notch, notch.na1, notch.na2, notch.b1, notch.b2, backend->scan_mod_c1, f);
        a.na1 = f.na1[0]; a.na2 = f.na2[0]; a.b1 = f.b1[0]; a.b2 = f.b2[0];
        a.notch_steps = f.steps[0];
        std::memcpy(a.nm, f.m[0], sizeof a.nm);
        a.notch_gain = (float)g_n;
    }
    for (int i = 0; i < 9; ++i) a.m[i] = (float)w.matrix[i];
    a.own_delay = w.own_delay ? 1 : 0;
    a.minavg = w.minavg;
    a.strip = w.strip_chroma ? 1 : 0;
    return cm_host::scan_launch_wrap_back(backend->scan_mod_c1, U8, backend->device, backend->scan_mod, a, g, stream);
}
int run_wrap_back(Geom g, const cm_plan *backend, const cm_comb_wrap_desc &w, int64_t first_frame, bool u8, hipStream_t stream) {
    g.lanes = reinterpret_cast<const LaneK<float> *>(backend->mod_lanes);
    g.carrier4 = backend->carrier4;
    g.carrier2 = backend->carrier2;
    g.cycle = backend->mod_cycle;
    g.n_lines = backend->mod_n_lines;
    set_first_frame(backend, g, first_frame, backend->mod_cycle);
    {
        const int mode = backend->small_batch;
        if (backend->scan_mod && (mode == CM_SMALL_BATCH_SCAN || (mode == CM_SMALL_BATCH_AUTO && g.total_calls <= CM_SCAN_MOD_MAX_CALLS)))
            return u8 ? wrap_back_scan<true>(g, backend, w, stream) : wrap_back_scan<false>(g, backend, w, stream);
        if (mode == CM_SMALL_BATCH_SCAN) return fail(CM_ERR_UNSUPPORTED, "the scan kernel does not serve this backend plan");
    }
    switch (backend->mod_shape) {
    case 1: return u8 ? launch_wrap_back<1, 2, true, false>(g, backend, w, stream) : launch_wrap_back<1, 2, false, false>(g, backend, w, stream);
    case 2: return u8 ? launch_wrap_back<2, 4, true, false>(g, backend, w, stream) : launch_wrap_back<2, 4, false, false>(g, backend, w, stream);
    default: return u8 ? launch_wrap_back<2, kModAnyShift, true, true>(g, backend, w, stream)
                       : launch_wrap_back<2, kModAnyShift, false, true>(g, backend, w, stream);
    }
}
int check_wrap(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w) {
    if (!inner || !backend || !w) return fail(CM_ERR_INVALID, "null argument");
    if (inner->secam || backend->secam || (first && first->secam)) return fail(CM_ERR_INVALID, "comb wrappers take QAM-family plans");
    if (!inner->fn) return fail(CM_ERR_UNSUPPORTED, inner->demod_error);
    if (first && !first->fn) return fail(CM_ERR_UNSUPPORTED, first->demod_error);
    if (!backend->mod_fn || backend->mod_depth) return fail(CM_ERR_UNSUPPORTED, "the backend plan needs a plain (not line-averaging) modulator");
    const cm_plan_desc &d = inner->desc;
    // one device for the three plans: their lane / carrier / scan tables are that device's memory, and check_device() below looks at inner's only
    if (backend->device != inner->device || (first && first->device != inner->device))
        return fail(CM_ERR_INVALID, "the inner, first and backend plans of a wrapped comb belong to different devices");
    if (backend->desc.width != d.width || (first && first->desc.width != d.width)) return fail(CM_ERR_INVALID, "the plans differ in width");
    if (backend->desc.height != d.height || (first && first->desc.height != d.height)) return fail(CM_ERR_INVALID, "the plans differ in height");
    if ((first != nullptr) != (d.first_is_plain != 0))
        return fail(CM_ERR_INVALID, "a `first` plan is needed exactly when the inner decoder takes call 0 of a run from the plain decoder");
    if (first && (first->has_first || first->desc.first_is_plain)) return fail(CM_ERR_INVALID, "the `first` plan must be a plain decoder");
    if (w->notch.n_sections && (w->notch.n_sections != 1 || w->notch.shift != 0)) return fail(CM_ERR_UNSUPPORTED, "notch: one section, shift 0");
    if (w->own_delay < 0 || w->own_delay > 1) return fail(CM_ERR_INVALID, "own_delay must be 0 or 1");
    if (w->minavg < 0 || w->minavg > 2) return fail(CM_ERR_INVALID, "minavg must be 0 (comb.avg), 1 (comb.minavg) or 2 (averaged by the caller)");
    return CM_OK;
}
// one non-blocking side stream per device, created on first use (the wrapped combs' first-line pass runs on it)
hipStream_t wrap_side_stream(int device) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        streams[device] = nullptr;
    }
    return streams[device];
}
// inner decoder over the calls of `g` into `scratch` ([frame][call][3][wp], or [call][3][wp] in rows mode), + the plain call 0s
int run_wrap_inner(const cm_plan *inner, const cm_plan *first, Geom g, float *scratch, int64_t first_frame, bool with_first,
                   hipStream_t stream) {
    g.out = scratch;
    g.out_plane_stride = g.Wp;
    g.out_row_stride = 3LL * g.Wp;
    g.out_frame_stride = 3LL * g.Wp * g.calls_per_frame;
    g.out_calls = g.rows_mode ? 0 : 1;
    g.skip_first = inner->desc.first_is_plain;
    set_first_frame(inner, g, first_frame, inner->main.cycle);
    Geom none = g;
    Geom s = g;
    if (first && with_first) {
        s.sparse = 1;
        s.skip_first = 0;
        s.total_calls = g.rows_mode ? 1 : (g.total_calls / g.calls_per_frame) * g.runs_per_frame;
        set_first_frame(first, s, first_frame, first->main.cycle);
        // small batches: both passes in ONE launch of the scan kernel, as a plan with a first-line pass of its own has them
        const int mode = inner->small_batch;
        if (inner->scan_main && first->scan_main && inner->scan_c1 == first->scan_c1 && first->small_batch == mode && g.total_calls > 0 &&
            (mode == CM_SMALL_BATCH_SCAN || (mode == CM_SMALL_BATCH_AUTO && g.total_calls <= CM_SCAN_MAX_CALLS))) {
            finish_geom(inner, inner->main, g);
            finish_geom(first, first->main, s);
            return launch_scan_as<false>(inner->scan_c1, inner->device, inner->scan_main, first->scan_main, inner->scan_depth, g, s, true, stream);
        }
    }
    if (!first || !with_first) return run_plan(inner, g, none, false, stream);
    // The plain first-line pass is one lane per run: a handful of workgroups whose launch lasts as long as walking one row (0.23 ms at 720
    // samples).  Behind the main pass on one stream that latency is paid per chunk; on a side stream of the device it runs beside the main
    // pass (forked after everything queued on `stream` - the previous chunk's back end still reads this scratch - and joined before the back end).
    hipStream_t side = wrap_side_stream(inner->device);
    struct EventPair {      // destroyed on every path out (the runtime releases them once the queued record / wait have completed)
        hipEvent_t forked = nullptr, joined = nullptr;
        ~EventPair() {
            if (forked) (void)hipEventDestroy(forked);
            if (joined) (void)hipEventDestroy(joined);
        }
    } ev;
    if (side && (hipEventCreateWithFlags(&ev.forked, hipEventDisableTiming) != hipSuccess ||
                 hipEventCreateWithFlags(&ev.joined, hipEventDisableTiming) != hipSuccess)) side = nullptr;
    if (side && (hipEventRecord(ev.forked, stream) != hipSuccess || hipStreamWaitEvent(side, ev.forked, 0) != hipSuccess)) side = nullptr;   // nothing queued on the side yet
    if (!side) {
        int rc = run_plan(inner, g, none, false, stream);
        if (!rc) rc = run_plan(first, s, none, false, stream);
        return rc;
    }
    // forked: from here on the main stream must join the side stream on EVERY path, or a later hipFreeAsync of the scratch on `stream`
    // could overtake the first-line kernel still running beside it
    const int rc_first = run_plan(first, s, none, false, side);
    const int rc_main = run_plan(inner, g, none, false, stream);
    const bool joined = hipEventRecord(ev.joined, side) == hipSuccess && hipStreamWaitEvent(stream, ev.joined, 0) == hipSuccess;
    if (!joined) {
        (void)hipStreamSynchronize(side);
        if (!rc_first && !rc_main) return fail(CM_ERR_LAUNCH, "joining the first-line pass of a wrapped comb failed");
    }
    return rc_first ? rc_first : rc_main;
}
struct AsyncBuf {
    hipStream_t stream = nullptr;
    void *p = nullptr;
    ~AsyncBuf() { if (p) (void)hipFreeAsync(p, stream); }
};
#ifndef CM_WRAP_SCRATCH_BYTES
#define CM_WRAP_SCRATCH_BYTES ((size_t)2 << 30)
#endif
// in: float rows (pitch wp), or in8: composite bytes (width = wp, a multiple of 4) with bytes out as well.
// h_top > 0: only the top h_top rows of every frame are decoded (frames stay full_H rows apart in both buffers) and only the calls with
// k < keep_calls of every run are stored - the share of a fused wrapped comb that mixes two front ends (wrap_frames_fused).
int wrap_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w, const float *in,
                const uint8_t *in8, void *out, int wp, int64_t n_frames, int64_t first_frame, hipStream_t stream, int h_top = 0,
                int keep_calls = 0, float *components = nullptr, int phase = 0) {
    // components != null: the caller's [frame][call][3][wp] buffer takes the place of the scratch, whole batch at once; phase 1 stops behind
    // the inner decoder (the buffer is the result), phase 2 starts at the back end (the buffer is the input) - avg= callables average in between
    const bool u8 = in8 != nullptr;
    const cm_plan_desc &d = inner->desc;
    const int W = d.width, full_H = d.height, H = h_top > 0 ? h_top : full_H, D = d.demodulation_delay + (w->own_delay ? 1 : 0);
    int rc = check_lines(inner, inner->main, H - 1 + 2 * D);
    if (rc) return rc;
    if (first && (rc = check_lines(first, first->main, H - 1))) return rc;
    if (H - 1 + 2 * D >= backend->mod_n_lines) return fail(CM_ERR_INVALID, "line number beyond the backend plan's phase tables");
    Geom g;
    std::memset(&g, 0, sizeof g);
    g.W = W;
    g.Wp = wp;
    g.H = H;
    g.in_frame_stride = (long long)wp * full_H;
    g.in_row_stride = wp;
    const int rows0 = (H + 1) / 2, rows1 = H / 2;
    g.calls_run0 = rows0 + D;
    g.calls_per_frame = g.calls_run0 + (rows1 > 0 ? rows1 + D : 0);
    g.runs_per_frame = rows1 > 0 ? 2 : 1;
    g.first_line[0] = 0;
    g.first_line[1] = 1;
    g.delay = D;
    // the component scratch: a byte budget (CM_WRAP_SCRATCH_BYTES, 2 GiB: 400 frames of 720 x 576 at a time; the header documents the peak;
    // every chunk costs the tails of two launches: 1 / 2 / 3 GiB measured 64 / 72 / 73 Gpixel/s at 1000 frames, round 3's 5 GB chunks 75).
    // Not smaller: the plain first-line pass is one lane per run - 2 runs per frame, a handful of workgroups whose time is the latency of
    // walking one row (0.23 ms) - and it is paid once per chunk.  With bytes at the boundary the level-decoded composite is a second,
    // chunk-sized buffer (`in8`: the whole batch's bytes; round 3 decoded them all at once).
    const size_t frame_bytes = (size_t)g.calls_per_frame * 3 * wp * sizeof(float);
    int64_t chunk = (int64_t)(CM_WRAP_SCRATCH_BYTES / frame_bytes);
    if (chunk < 1) chunk = 1;
    if (chunk > n_frames || components) chunk = n_frames;
    AsyncBuf scratch, comp;
    scratch.stream = comp.stream = stream;
    if (int rc_ = refuse_capture(stream, "a wrapped comb decoder")) return rc_;
    if (!components) HIP_TRY(hipMallocAsync(&scratch.p, (size_t)chunk * frame_bytes, stream), CM_ERR_LAUNCH);
    float *const sc = components ? components : (float *)scratch.p;
    const long long frame_quads = (long long)H * (wp / 4);
    if (in8) HIP_TRY(hipMallocAsync(&comp.p, (size_t)chunk * frame_quads * 16, stream), CM_ERR_LAUNCH);
    for (int64_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const int64_t nf = n_frames - f0 < chunk ? n_frames - f0 : chunk;
        Geom gi = g;
        if (in8) {      // image.py:24-25, 62: the inner decoder's component output has no byte form, so it reads float rows
            const long long quads = nf * frame_quads;
            hipLaunchKernelGGL(decode_level_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, in8 + f0 * (long long)wp * full_H,
                               (float *)comp.p, quads, frame_quads, (long long)wp * full_H);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("decode_level_kernel launch: ") + hipGetErrorString(e));
            gi.in = (const float *)comp.p;
            gi.in_frame_stride = (long long)wp * H;      // the decoded copy holds the decoded rows only
        } else
        gi.in = in + f0 * g.in_frame_stride;
        gi.total_calls = nf * g.calls_per_frame;
        if (phase != 2 && (rc = run_wrap_inner(inner, first, gi, sc, first_frame + f0, true, stream))) return rc;
        if (phase == 1) continue;
        Geom gb = g;
        gb.in = sc;
        gb.in_plane_stride = wp;
        gb.in_row_stride = 3LL * wp;
        gb.in_frame_stride = 3LL * wp * g.calls_per_frame;
        gb.in_calls = 1;
        gb.total_calls = gi.total_calls;
        gb.keep_calls = keep_calls;
        if (u8) {   // interleaved bytes [F][H][W][3]: strides count bytes
            gb.out = reinterpret_cast<float *>((unsigned char *)out + f0 * 3LL * W * full_H);
            gb.out_frame_stride = 3LL * W * full_H;
            gb.out_row_stride = 3LL * W;
        } else {
            gb.out = (float *)out + f0 * 3LL * wp * full_H;
            gb.out_plane_stride = (long long)wp * full_H;
            gb.out_frame_stride = 3LL * wp * full_H;
            gb.out_row_stride = wp;
        }
        if ((rc = run_wrap_back(gb, backend, *w, first_frame + f0, u8, stream))) return rc;
    }
    return CM_OK;
}
// SimpleCombModem / Simple3DCombModem around PalDModem without the component scratch (40 -> 16 B per pixel through HBM): from its third
// call on, a run's two chroma estimates (comb.py:103-104) both come from the PAL-D front end, so the average, the re-modulation at the
// wrapper's line (comb.py:105-106) and the notch are one more line of history of the fused decoder - `fused`: PAL-D front end, depth 2,
// the lane tables of plan.py (QamTables: fused_main) - which stores every call with k >= 2.  The calls k < 2 of every run mix in the
// plain first-line decode (the QAM front end): they are the top four rows of every frame, and go through the composition above.
bool wrap_fused_applies(const cm_plan *fused, const cm_plan *inner, int64_t n_frames) {
    if (!fused || !fused->fn || (fused->desc.skip_calls != 2 && !fused->main.wrap_mode)) return false;
    const cm_plan_desc &d = inner->desc;
    if (d.height < 8 || d.width % 4 != 0) return false;
    if (inner->small_batch != CM_SMALL_BATCH_AUTO) return false;          // a pinned kernel family: the composition honours it
    return n_frames * (long long)(d.height + 4) > 4LL * CM_SCAN_MAX_CALLS;   // below: the scan kernels' regime
}
int check_fused(const cm_plan *fused, const cm_plan *inner, const cm_comb_wrap_desc *w) {
    if (!fused) return CM_OK;
    if (fused->secam) return fail(CM_ERR_INVALID, "comb wrappers take QAM-family plans");
    if (fused->device != inner->device) return fail(CM_ERR_INVALID, "the fused and inner plans of a wrapped comb belong to different devices");
    if (fused->desc.width != inner->desc.width || fused->desc.height != inner->desc.height) return fail(CM_ERR_INVALID, "the plans differ in size");
    if (fused->desc.demodulation_delay != inner->desc.demodulation_delay + (w->own_delay ? 1 : 0))
        return fail(CM_ERR_INVALID, "the fused plan's demodulation delay is not the inner decoder's plus the wrapper's");
    return CM_OK;
}
int wrap_frames_fused(const cm_plan *fused, const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                      const float *in, const uint8_t *in8, void *out, int wp, int64_t n_frames, int64_t first_frame, hipStream_t stream) {
    const cm_plan_desc &d = fused->desc;
    const int W = d.width, H = d.height, D = d.demodulation_delay;
    int rc = check_lines(fused, fused->main, H - 1 + 2 * D);
    if (rc) return rc;
    Geom g;
    std::memset(&g, 0, sizeof g);
    g.W = W;
    g.H = H;
    if (in8) {      // strides count bytes (PassCfg::U8)
        g.in = reinterpret_cast<const float *>(in8);
        g.out = reinterpret_cast<float *>(out);
        g.Wp = W;
        g.in_frame_stride = (long long)W * H;
        g.in_row_stride = W;
        g.out_frame_stride = 3LL * W * H;
        g.out_row_stride = 3LL * W;
    } else {
        g.in = in;
        g.out = (float *)out;
        g.Wp = wp;
        g.in_frame_stride = (long long)wp * H;
        g.in_row_stride = wp;
        g.out_plane_stride = (long long)wp * H;
        g.out_frame_stride = 3LL * wp * H;
        g.out_row_stride = wp;
    }
    set_first_frame(fused, g, first_frame, fused->main.cycle);
    const int rows0 = (H + 1) / 2, rows1 = H / 2;
    g.calls_run0 = rows0 + D;
    g.calls_per_frame = g.calls_run0 + (rows1 > 0 ? rows1 + D : 0);
    g.runs_per_frame = rows1 > 0 ? 2 : 1;
    g.first_line[0] = 0;
    g.first_line[1] = 1;
    g.delay = D;
    g.total_calls = n_frames * g.calls_per_frame;
    if (fused->main.wrap_mode) {      // a two-level comb (around Pal3DModem: one front end, so every call of every run): the whole decode
        Geom none = g;
        return run_plan(fused, g, none, false, stream, in8 != nullptr);
    }
    g.skip_first = 2;
    Geom none = g;
    if ((rc = run_plan(fused, g, none, false, stream, in8 != nullptr))) return rc;
    return wrap_frames(inner, first, backend, w, in, in8, out, wp, n_frames, first_frame, stream, 4, 2);
}
}  // namespace
}  // extern "C++"

extern "C" {
int cm_comb_wrap_demodulate_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                                   const float *composite, float *rgb, int64_t n_frames, int64_t first_frame, void *stream) {
    if (int rc = check_wrap(inner, first, backend, w)) return rc;
    if (n_frames == 0) return CM_OK;
    if (!composite || !rgb) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(inner->device, composite, rgb)) return rc_;
    const cm_plan_desc &d = inner->desc;
    const int W = d.width, H = d.height, wp = (W + 3) & ~3;
    return with_pitched_rows(composite, n_frames * H, rgb, n_frames * 3 * H, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        return wrap_frames(inner, first, backend, w, in, nullptr, out, wp, n_frames, first_frame, (hipStream_t)stream);
    });
}

int cm_comb_wrap_demodulate_frames_u8(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                                      const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames, int64_t first_frame, void *stream) {
    if (int rc = check_wrap(inner, first, backend, w)) return rc;
    if (n_frames == 0) return CM_OK;
    if (!composite8 || !rgb8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(inner->device, composite8, rgb8)) return rc_;
    const cm_plan_desc &d = inner->desc;
    const int W = d.width, H = d.height;
    if (W % 4 != 0) return fail(CM_ERR_UNSUPPORTED, "the fused uint8 boundary needs a width that is a multiple of 4");
    return wrap_frames(inner, first, backend, w, nullptr, composite8, rgb8, W, n_frames, first_frame, (hipStream_t)stream);
}

int cm_comb_wrap_demodulate_frames_fused(const cm_plan *fused, const cm_plan *inner, const cm_plan *first, const cm_plan *backend,
                                         const cm_comb_wrap_desc *w, const float *composite, float *rgb, int64_t n_frames, int64_t first_frame,
                                         void *stream) {
    if (int rc = check_wrap(inner, first, backend, w)) return rc;
    if (int rc = check_fused(fused, inner, w)) return rc;
    if (!wrap_fused_applies(fused, inner, n_frames))
        return cm_comb_wrap_demodulate_frames(inner, first, backend, w, composite, rgb, n_frames, first_frame, stream);
    if (!composite || !rgb) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(inner->device, composite, rgb)) return rc_;
    return wrap_frames_fused(fused, inner, first, backend, w, composite, nullptr, rgb, inner->desc.width, n_frames, first_frame, (hipStream_t)stream);
}

int cm_comb_wrap_demodulate_frames_fused_u8(const cm_plan *fused, const cm_plan *inner, const cm_plan *first, const cm_plan *backend,
                                            const cm_comb_wrap_desc *w, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames,
                                            int64_t first_frame, void *stream) {
    if (int rc = check_wrap(inner, first, backend, w)) return rc;
    if (int rc = check_fused(fused, inner, w)) return rc;
    if (!wrap_fused_applies(fused, inner, n_frames) || !fused->fn_u8)
        return cm_comb_wrap_demodulate_frames_u8(inner, first, backend, w, composite8, rgb8, n_frames, first_frame, stream);
    if (!composite8 || !rgb8) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(inner->device, composite8, rgb8)) return rc_;
    return wrap_frames_fused(fused, inner, first, backend, w, nullptr, composite8, rgb8, inner->desc.width, n_frames, first_frame, (hipStream_t)stream);
}

// phase 0: composite rows -> rgb rows; 1: composite rows -> `components` [n][3][W]; 2: `components` -> rgb rows (widths that are multiples of 4)
static int wrap_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w, const float *composite,
                    float *components, float *rgb, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0, void *stream, int phase) {
    if (int rc = check_wrap(inner, first, backend, w)) return rc;
    if (n_calls == 0) return CM_OK;
    if ((phase != 2 && !composite) || (phase != 1 && !rgb) || (phase != 0 && !components)) return fail(CM_ERR_INVALID, "null argument");
    if (n_calls < 0 || frame < 0 || k0 < 0 || first_line < 0) return fail(CM_ERR_INVALID, "negative count / frame / line / k0");
    if (int rc_ = check_device(inner->device, phase == 2 ? components : composite, phase == 1 ? components : rgb)) return rc_;
    const cm_plan_desc &d = inner->desc;
    const int W = d.width, wp = (W + 3) & ~3;
    if (phase != 0 && wp != W) return fail(CM_ERR_UNSUPPORTED, "the component buffer form needs a width that is a multiple of 4");
    if (phase == 2) composite = components;      // (any valid rows: with_pitched_rows passes aligned rows through untouched)
    if (phase == 1) rgb = components;
    const int last_line = first_line + 2 * (n_calls - 1);
    int rc = check_lines(inner, inner->main, last_line);
    if (rc) return rc;
    if (first && k0 == 0 && (rc = check_lines(first, first->main, first_line))) return rc;
    if (last_line >= backend->mod_n_lines) return fail(CM_ERR_INVALID, "line number beyond the backend plan's phase tables");
    return with_pitched_rows(composite, n_calls, rgb, 3LL * n_calls, W, (hipStream_t)stream, [&](const float *in, float *out) -> int {
        Geom g;
        std::memset(&g, 0, sizeof g);
        g.in = in;
        g.W = W;
        g.Wp = wp;
        g.H = n_calls;
        g.rows_mode = 1;
        g.calls_run0 = g.calls_per_frame = n_calls;
        g.runs_per_frame = 1;
        g.first_line[0] = g.first_line[1] = first_line;
        g.k0 = k0;
        g.total_calls = n_calls;
        if (int rc_ = refuse_capture((hipStream_t)stream, "a wrapped comb decoder")) return rc_;
        AsyncBuf scratch;
        scratch.stream = (hipStream_t)stream;
        if (phase == 0) HIP_TRY(hipMallocAsync(&scratch.p, (size_t)n_calls * 3 * wp * sizeof(float), (hipStream_t)stream), CM_ERR_LAUNCH);
        float *const sc = phase == 0 ? (float *)scratch.p : components;
        if (phase != 2) {
            int rc2 = run_wrap_inner(inner, first, g, sc, frame, k0 == 0, (hipStream_t)stream);
            if (rc2 || phase == 1) return rc2;
        }
        Geom gb = g;
        gb.in = sc;
        gb.in_plane_stride = wp;
        gb.in_row_stride = 3LL * wp;
        gb.out = out;
        gb.out_plane_stride = wp;            // rows mode writes [call][plane][W]
        gb.out_row_stride = 3LL * wp;
        return run_wrap_back(gb, backend, *w, frame, false, (hipStream_t)stream);
    });
}
int cm_comb_wrap_demodulate_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                                const float *composite, float *rgb, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0,
                                void *stream) {
    return wrap_run(inner, first, backend, w, composite, nullptr, rgb, n_calls, frame, first_line, k0, stream, 0);
}
// The composition cut in two for avg= callables (comb.py:72, 81-84, 103-104): the caller averages the (u, v) planes of consecutive calls of the
// component buffer between the halves (cm_comb_wrap_desc.minavg = 2: the back end takes them as they are).
int cm_comb_wrap_components_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                                const float *composite, float *components, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0,
                                void *stream) {
    return wrap_run(inner, first, backend, w, composite, components, nullptr, n_calls, frame, first_line, k0, stream, 1);
}
int cm_comb_wrap_finish_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                            float *components, float *rgb, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0, void *stream) {
    return wrap_run(inner, first, backend, w, nullptr, components, rgb, n_calls, frame, first_line, k0, stream, 2);
}
int cm_comb_wrap_calls_per_frame(const cm_plan *inner, const cm_comb_wrap_desc *w) {
    if (!inner || !w) return fail(CM_ERR_INVALID, "null argument");
    const int H = inner->desc.height, D = inner->desc.demodulation_delay + (w->own_delay ? 1 : 0);
    return (H + 1) / 2 + D + (H / 2 > 0 ? H / 2 + D : 0);
}
static int wrap_frames_split(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w, const float *composite,
                             float *components, float *rgb, int64_t n_frames, int64_t first_frame, void *stream, int phase) {
    if (int rc = check_wrap(inner, first, backend, w)) return rc;
    if (n_frames == 0) return CM_OK;
    if ((phase == 1 && !composite) || (phase == 2 && !rgb) || !components) return fail(CM_ERR_INVALID, "null argument");
    if (n_frames < 0 || first_frame < 0) return fail(CM_ERR_INVALID, "negative frame count / number");
    if (int rc_ = check_device(inner->device, phase == 1 ? composite : components, phase == 1 ? components : rgb)) return rc_;
    const int W = inner->desc.width;
    if (W % 4 != 0) return fail(CM_ERR_UNSUPPORTED, "the component buffer form needs a width that is a multiple of 4");
    return wrap_frames(inner, first, backend, w, composite, nullptr, rgb, W, n_frames, first_frame, (hipStream_t)stream, 0, 0, components, phase);
}
int cm_comb_wrap_components_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                                   const float *composite, float *components, int64_t n_frames, int64_t first_frame, void *stream) {
    return wrap_frames_split(inner, first, backend, w, composite, components, nullptr, n_frames, first_frame, stream, 1);
}
int cm_comb_wrap_finish_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *w,
                               float *components, float *rgb, int64_t n_frames, int64_t first_frame, void *stream) {
    return wrap_frames_split(inner, first, backend, w, nullptr, components, rgb, n_frames, first_frame, stream, 2);
}
}  // extern "C"

#ifdef CM_DIAG
extern "C" void cm_diag_set_buffer(unsigned long long *dev) { g_diag = dev; }
#endif

void cm_set_pointer_check(int32_t on) { g_pointer_check = on != 0; }
int cm_plan_set_small_batch(const cm_plan *p, int32_t mode) {
    if (!p) return fail(CM_ERR_INVALID, "null argument");
    if (mode < CM_SMALL_BATCH_AUTO || mode > CM_SMALL_BATCH_SCAN) return fail(CM_ERR_INVALID, "unknown small-batch mode");
    if (mode == CM_SMALL_BATCH_SCAN && !p->scan_main && !p->scan_mod && !p->scan_smod && !p->scan_sdem) return fail(CM_ERR_UNSUPPORTED, "the scan kernels do not serve this plan");
    p->small_batch = mode;
    return CM_OK;
}
int cm_plan_describe(const cm_plan *p, char *buf, int32_t buf_len) {
    if (!p || !buf || buf_len < 1) return 0;
#ifdef CM_EXPERIMENTS
    const char *exp = "; EXPERIMENTS BUILD (-DCM_EXPERIMENTS: ablation switches may be active, results may be wrong)";
#else
    const char *exp = "";
#endif
    int n = snprintf(buf, buf_len, "%s; calls per workgroup 64 (%s), halo %d%s", p->main.name.c_str(),
                     (p->pair || p->main.name.find("_pair") != std::string::npos) ? "two wavefronts: front end | detectors + back end" : "one wavefront",
                     p->main.depth, exp);
    return n < buf_len ? n : buf_len - 1;
}

}  // extern "C"

#endif  // CM_MAIN_PART

#if CM_SCAN_PART
// ---- CM_PART 3: the row-parallel scan kernels of the QAM / SECAM families (cm_scan_kernels.h) behind five launch functions ----------------
namespace {
template <int C1, int NW, bool U8>
int scan_demod_i(int device, const ScanK *km, const ScanK *kf, int depth, const Geom &gm, const Geom &gf, bool with_first, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)demod_scan_kernel<C1, NW, U8>, device, lds, "the scan kernel")) return rc;
    const long long n_first = with_first ? (gf.total_calls + NW - 1) / NW : 0;
    const int per = gm.sparse ? NW : NW - depth;      // calls per workgroup behind the halo waves
    const long long n_main = (gm.total_calls + per - 1) / per;
    hipLaunchKernelGGL((demod_scan_kernel<C1, NW, U8>), dim3((int)(n_first + n_main)), dim3(64 * NW), lds, stream, gm, gf, km,
                       with_first ? kf : km, (int)n_first);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("demod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <int C1, int NW, bool U8>
int scan_qam_mod_i(int device, const ScanModK *k, const Geom &g, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_mod_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)qam_mod_scan_kernel<C1, NW, U8>, device, lds, "the modulator's scan kernel")) return rc;
    const long long blocks = (g.total_calls + NW - 1) / NW;
    hipLaunchKernelGGL((qam_mod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("qam_mod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <int C1, int NW, bool U8>
int scan_secam_mod_i(int device, const ScanSecamModK *k, const Geom &g, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_secam_mod_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)secam_mod_scan_kernel<C1, NW, U8>, device, lds, "the SECAM modulator's scan kernel")) return rc;
    const long long blocks = (g.total_calls + NW - 1) / NW;
    hipLaunchKernelGGL((secam_mod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("secam_mod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
template <int C1, int NW, bool U8>
int scan_secam_demod_i(int device, const ScanSecamK *k, const Geom &g, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_secam_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)secam_demod_scan_kernel<C1, NW, U8>, device, lds, "the SECAM decoder's scan kernel")) return rc;
    const long long blocks = (g.total_calls + (NW - 1) - 1) / (NW - 1);
    hipLaunchKernelGGL((secam_demod_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("secam_demod_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
// the wrapped combs' back end: one wavefront per call (wrap_back_scan_kernel), the pre-correction constants are the backend modulator's
template <int C1, int NW, bool U8>
int scan_wrap_back_i(int device, const ScanModK *k, const ScanWrapArgs &a, const Geom &g, hipStream_t stream) {
    const size_t lds = sizeof(float) * (size_t)NW * scan_mod_wave_floats<C1>();
    if (int rc = allow_dynamic_lds((const void *)wrap_back_scan_kernel<C1, NW, U8>, device, lds, "the wrapped combs' scan kernel")) return rc;
    const long long blocks = (g.total_calls + NW - 1) / NW;
    hipLaunchKernelGGL((wrap_back_scan_kernel<C1, NW, U8>), dim3((int)blocks), dim3(64 * NW), lds, stream, g, k, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CM_ERR_LAUNCH, std::string("wrap_back_scan_kernel launch: ") + hipGetErrorString(e));
    return CM_OK;
}
}  // namespace
// (chunks of 24 / 32 samples run three waves per workgroup in the decoder: its rows are twice as long)
#define CM_SCAN_DISPATCH(fn, nw_long, ...)                                                                                  \
    switch (c1) {                                                                                                           \
        case 12: return u8 ? fn<12, 4, true>(__VA_ARGS__) : fn<12, 4, false>(__VA_ARGS__);                                  \
        case 16: return u8 ? fn<16, 4, true>(__VA_ARGS__) : fn<16, 4, false>(__VA_ARGS__);                                  \
        case 24: return u8 ? fn<24, nw_long, true>(__VA_ARGS__) : fn<24, nw_long, false>(__VA_ARGS__);                      \
        default: return u8 ? fn<32, nw_long, true>(__VA_ARGS__) : fn<32, nw_long, false>(__VA_ARGS__);                      \
    }
int cm_host::scan_launch_demod(int c1, bool u8, int device, const ScanK *km, const ScanK *kf, int depth, const Geom &gm, const Geom &gf,
                               bool with_first, hipStream_t stream) {
    CM_SCAN_DISPATCH(scan_demod_i, 3, device, km, kf, depth, gm, gf, with_first, stream)
}
int cm_host::scan_launch_qam_mod(int c1, bool u8, int device, const ScanModK *k, const Geom &g, hipStream_t stream) {
    CM_SCAN_DISPATCH(scan_qam_mod_i, 4, device, k, g, stream)
}
int cm_host::scan_launch_secam_mod(int c1, bool u8, int device, const ScanSecamModK *k, const Geom &g, hipStream_t stream) {
    CM_SCAN_DISPATCH(scan_secam_mod_i, 4, device, k, g, stream)
}
int cm_host::scan_launch_secam_demod(int c1, bool u8, int device, const ScanSecamK *k, const Geom &g, hipStream_t stream) {
    if (c1 == 12) return u8 ? scan_secam_demod_i<12, 4, true>(device, k, g, stream) : scan_secam_demod_i<12, 4, false>(device, k, g, stream);
    return u8 ? scan_secam_demod_i<16, 4, true>(device, k, g, stream) : scan_secam_demod_i<16, 4, false>(device, k, g, stream);
}
int cm_host::scan_launch_wrap_back(int c1, bool u8, int device, const ScanModK *k, const ScanWrapArgs &a, const Geom &g, hipStream_t stream) {
    CM_SCAN_DISPATCH(scan_wrap_back_i, 4, device, k, a, g, stream)
}
#undef CM_SCAN_DISPATCH
#endif  // CM_SCAN_PART

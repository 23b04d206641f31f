// cm_api_select.h - the decoder plan (cm_plan), its passes and the choice of kernel instance per filter-set shape:
// launch_demod, make_pass(es), select_for_shape (tuned shapes), select_any (run-time shape), the wide rasters (cm_shapes_wide.h).
// CM_PART 1, 4, 5 - 7.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)

namespace {
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
}  // namespace

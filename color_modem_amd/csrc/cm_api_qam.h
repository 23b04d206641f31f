// cm_api_qam.h - the QAM / SECAM families: plan construction, launch geometry, small-batch routing (segments, scan constants), pitched staging,
// then the C entry points cm_plan_* / cm_demodulate_* / cm_modulate_* / cm_filter_rows_f64 / cm_notch_luma_f32.  CM_PART 1.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)

namespace {
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
    // (ABI 8 gave cm_lane_table::reserved a meaning for demod_main only: anything else there is an ABI-7 caller's garbage or a request on the wrong table)
    if (desc->demod_first.wrap_mode != 0 || desc->mod_main.wrap_mode != 0)
        return fail(CM_ERR_INVALID, "wrap_mode belongs to demod_main: demod_first.wrap_mode and mod_main.wrap_mode must be 0");
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
    // a two-level comb plan (demod_main.wrap_mode) serves the frame entry points: its run form - the wrapper's state across the calls of a
    // caller's run - is the composition's (cm_comb_wrap_demodulate_run), and nothing tests this kernel on runs
    if (p->main.wrap_mode != 0) return fail(CM_ERR_UNSUPPORTED, "a two-level comb plan (wrap_mode) serves the frame entry points only");
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

}  // extern "C"
#endif  // CM_MAIN_PART

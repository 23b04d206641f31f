// cm_api_wrap.h - SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem: cm_comb_wrap_* (cm_wrap_kernels.h), and the plan utilities
// (cm_set_pointer_check, cm_plan_set_small_batch, cm_plan_describe).  CM_PART 1.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)

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
// two non-blocking side streams per device, created on first use: 0 - the wrapped combs' first-line pass; 1 - the composition on the top
// rows of a fused wrapped comb (wrap_frames_fused), which forks to 0 itself
hipStream_t wrap_side_stream(int device, int which = 0) {
    static std::mutex mu;
    static hipStream_t streams[2][64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    hipStream_t &s = streams[which ? 1 : 0][device];
    if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {       // (created with the device's highest priority: no
        (void)hipGetLastError();                                                          //  difference measured, profiles/r06_wrapped_top_rows.txt)
        s = nullptr;
    }
    return s;
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
#ifndef CM_WRAP_TOP_BESIDE
#define CM_WRAP_TOP_BESIDE 1
#endif
    // The composition on the top four rows is a few thousand calls on the scan kernels (0.1 ms at 720 samples per line, 0.27 ms at 1280: 5 - 10 %
    // of the batch behind the fused pass on one stream); it stores the calls k < 2, the fused pass the others, so on a side stream of the
    // device it runs BESIDE the fused pass: forked behind everything queued on `stream`, joined before anything queued after this call.
    hipStream_t side = CM_WRAP_TOP_BESIDE ? wrap_side_stream(fused->device, 1) : nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (side && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) side = nullptr;   // wrap_frames below refuses it
    struct EventPair {
        hipEvent_t forked = nullptr, joined = nullptr;
        ~EventPair() {
            if (forked) (void)hipEventDestroy(forked);
            if (joined) (void)hipEventDestroy(joined);
        }
    } ev;
    if (side && (hipEventCreateWithFlags(&ev.forked, hipEventDisableTiming) != hipSuccess ||
                 hipEventCreateWithFlags(&ev.joined, hipEventDisableTiming) != hipSuccess)) side = nullptr;
    if (side && (hipEventRecord(ev.forked, stream) != hipSuccess || hipStreamWaitEvent(side, ev.forked, 0) != hipSuccess)) side = nullptr;
    if (!side) {
        if ((rc = run_plan(fused, g, none, false, stream, in8 != nullptr))) return rc;
        return wrap_frames(inner, first, backend, w, in, in8, out, wp, n_frames, first_frame, stream, 4, 2);
    }
    // forked: the caller's stream joins the side stream on every path out.  (The top rows on the STREAMING kernels instead - least work per
    // call, three row walks of latency hidden behind a long fused pass - measured no better than the scan kernels beside it: 114 / 118 / 105
    // against 119 / 117 / 110 Gpixel/s at 1024 / 1280 / 1920 samples per line, profiles/r06_wrapped_top_rows.txt.)
    const int rc_top = wrap_frames(inner, first, backend, w, in, in8, out, wp, n_frames, first_frame, side, 4, 2);
    const int rc_main = run_plan(fused, g, none, false, stream, in8 != nullptr);
    const bool joined = hipEventRecord(ev.joined, side) == hipSuccess && hipStreamWaitEvent(stream, ev.joined, 0) == hipSuccess;
    if (!joined) {
        (void)hipStreamSynchronize(side);
        if (!rc_top && !rc_main) return fail(CM_ERR_LAUNCH, "joining the top rows of a fused wrapped comb failed");
    }
    return rc_top ? rc_top : rc_main;
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

extern "C" {
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

// cm_api_mac.h - the D2-MAC style time-multiplex modem: cm_mac_* (cm_mac_kernels.h).  CM_PART 1.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)

#if CM_MAIN_PART
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
        // rows beyond 1920 samples / lines beyond ~13000 take more than the 64 KiB a kernel gets by default (round 6: up to the CU's 160 KiB)
        if (demod && lds_demod > 64 * 1024) {
            if (int rc = u8 ? allow_dynamic_lds((const void *)cm::mac_demod_generic_kernel<true>, p->device, lds_demod, "the MAC decoder")
                            : allow_dynamic_lds((const void *)cm::mac_demod_generic_kernel<false>, p->device, lds_demod, "the MAC decoder")) return rc;
        }
        if (!demod && lds_mod > 64 * 1024) {
            if (int rc = u8 ? allow_dynamic_lds((const void *)cm::mac_mod_generic_kernel<true>, p->device, lds_mod, "the MAC encoder")
                            : allow_dynamic_lds((const void *)cm::mac_mod_generic_kernel<false>, p->device, lds_mod, "the MAC encoder")) return rc;
        }
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
    // one workgroup holds a call's rows in LDS: (7 with line averaging, else 4) rows of `width` floats + 2160 in the encoder, the line + 2904
    // in the decoder, of the CU's 160 KiB (ref mac.py:49-55, 71-74, 88-91 take any rational ratio; rounds 1 - 5 stopped at 1920 / 4096)
    if (desc->width > 4096) return fail(CM_ERR_UNSUPPORTED, "MAC: rows of more than 4096 samples do not fit the encoder's LDS layout");
    if (desc->line_width > 16384) return fail(CM_ERR_UNSUPPORTED, "MAC: lines of more than 16384 samples do not fit the decoder's LDS layout");
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

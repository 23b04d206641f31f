// cm_api_am.h - the amplitude-modulated line-sequential standards, Proto-SECAM and NIIR: cm_am_* (cm_am_kernels.h, cm_am_scan_kernels.h).  CM_PART 2.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)

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

// cm_api_scan.h - CM_PART 3: the row-parallel scan kernels of the QAM / SECAM families (cm_scan_kernels.h) behind five launch functions.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)


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


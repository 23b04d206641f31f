// cm_api_common.h - state shared by the parts of cm_api.hip and the helpers every part uses:
// last error, pointer check, device / buffer checks, HIP_TRY, dynamic-LDS permission.
// (a fragment of the one translation unit cm_api.hip includes in order: not a header to include on its own)

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

}  // namespace

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

// The source is cut by family into the fragments included at the bottom (round 6): cm_api_common.h (shared state and helpers),
// cm_api_select.h (cm_plan, passes, kernel-instance selection), cm_api_qam.h, cm_api_mac.h, cm_api_am.h, cm_api_wrap.h, cm_api_scan.h.
// CM_PART: the whole compiles as ONE translation unit (0, the default) or as seven that __graft_entry__.build() compiles side by side and
// links into the one library - 1: the QAM / SECAM / MAC / wrapped-comb entry points and their streaming kernels (decoder instances of the
// PAL-BG filter shapes), 2: the cm_am_* entry points (Proto-SECAM / NIIR, streaming and scan kernels), 3: the row-parallel scan kernels of
// part 1's families behind five launch functions (cm_host::scan_launch_*), 4: the decoder instances of every other filter-set shape (NTSC /
// PAL-M/N, NTSC-I, NTSC-A, the 640 / 704 / 768 sample rasters, the run-time shape) behind cm_host::select_other_shapes.  The helpers of
// cm_api_common.h are in every part; the process-wide state (last error, pointer check) lives in part 1.
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

#include "cm_api_common.h"     // shared state and helpers
#include "cm_api_select.h"     // cm_plan, passes, kernel-instance selection (parts 1, 4 - 7)
#include "cm_api_qam.h"        // QAM / SECAM plans and entry points (part 1)
#include "cm_api_mac.h"        // cm_mac_* (part 1)
#include "cm_api_am.h"         // cm_am_* (part 2)
#include "cm_api_wrap.h"       // cm_comb_wrap_*, plan utilities (part 1)
#include "cm_api_scan.h"       // scan-kernel launchers (part 3)

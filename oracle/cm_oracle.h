/*
 * cm_oracle.h - CPU oracle for the color_modem hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A float64, single-row-at-a-time restatement of the reference's per-line algorithms
 * (kFYatek/color_modem: qam.py, comb.py, color/{ntsc,pal,secam}.py, utils.py, line.py and the
 * row schedule of image.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product path (color_modem_amd + libcolor_modem_hip.so) never
 * does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py holds this code to <= 1e-11 against
 * golden vectors produced by the reference itself in the build container
 * (tests/golden/make_golden.py, scipy 1.15.3 / numpy 2.2.6, NTSC via the legacy-iirdesign shim
 * described there).
 *
 * Filter *design* is not restated here: (b, a, shift, phase_shift) come in through the
 * descriptor (in the tests: from color_modem_amd's scipy-based design code, itself checked
 * against tests/golden/plans.json).  Everything applied per sample - lfilter, the
 * resample_poly FIR (designed here, firwin(41, 0.5, kaiser 5.0)), carrier phases, line
 * geometry, comb logic, FM modulation/demodulation, the image.py row schedule - is restated.
 */
#ifndef CM_ORACLE_H
#define CM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_COEF 12
#define ORC_MAX_FILTERS 8

/* utils.py:9-36 FilterFunction state after construction */
typedef struct {
    int32_t nb, na;
    int32_t shift;
    int32_t present; /* 0: filter absent (reference holds None) */
    double b[ORC_MAX_COEF];
    double a[ORC_MAX_COEF];
    double phase_shift;
} orc_filter_t;

enum orc_kind {
    ORC_PAL_S = 1,     /* color/pal.py:28-59   */
    ORC_PAL_D = 2,     /* color/pal.py:62-127  */
    ORC_PAL_3D = 3,    /* color/pal.py:130-234 */
    ORC_NTSC = 4,      /* color/ntsc.py:23-49  */
    ORC_NTSC_COMB = 5, /* color/ntsc.py:52-82  */
    ORC_SECAM = 6      /* color/secam.py:152-304 */
};

enum orc_wrapper {
    ORC_WRAP_NONE = 0,
    ORC_WRAP_SIMPLE_COMB = 1,    /* comb.py:71-122, delay=False */
    ORC_WRAP_SIMPLE_3D_COMB = 2, /* comb.py:125-127 (delay=True) */
    ORC_WRAP_COLOR_AVERAGING = 3 /* comb.py:130-167 */
};

/* filter slots in orc_desc_t.filters */
enum {
    ORC_F_QAM_PRE = 0,      /* qam.py:16 */
    ORC_F_QAM_EXTRACT2X = 1, /* qam.py:17 */
    ORC_F_QAM_REMOVE2X = 2, /* qam.py:17 */
    ORC_F_QAM_DEMOD_LP = 3, /* qam.py:18 */
    ORC_F_PALD_LP = 4,      /* pal.py:67-69 */
    ORC_F_COMB_NOTCH = 5,   /* comb.py:29-31: notch of PalD / Pal3D / NtscComb (absent: notch=0.0) */
    ORC_F_WRAP_NOTCH = 6,   /* comb.py:86-88: notch of SimpleCombModem / Simple3DCombModem */
    ORC_F_SECAM_PRE_LP = 0,   /* secam.py:171-172 */
    ORC_F_SECAM_LF_PRE = 1,   /* secam.py:175-177 forward */
    ORC_F_SECAM_LF_REV = 2,   /* secam.py:175-177 backward */
    ORC_F_SECAM_BELL = 3,     /* secam.py:168-170 */
    ORC_F_SECAM_CHROMA_BP = 4, /* secam.py:183-184 */
    ORC_F_SECAM_LUMA_BS = 5,  /* secam.py:185-186 */
    ORC_F_SECAM_FM_LP = 6     /* secam.py:131-132 */
};

typedef struct {
    int32_t kind;    /* enum orc_kind */
    int32_t wrapper; /* enum orc_wrapper */
    int32_t use_minavg; /* comb.py:13-15 instead of comb.py:9-10: bit 0 Pal3DModem(avg=), bit 1 SimpleCombModem(avg=) */
    int32_t alternate_phases; /* secam.py:164-167 */
    /* line.py:6-13 LineStandard fields + line.py:50-55 LineConfig */
    double frame_rate;
    int32_t total_lines;
    int32_t odd_first, odd_last, even_first, even_last;
    int32_t width, height;
    double total_width_factor;
    /* QAM systems: qam.py:8, utils.py:77-80 (frame_cycle computed by the caller with
       fractions.Fraction, exactly as the reference does) */
    double fsc;
    int32_t frame_cycle;
    int32_t pal3d_disable; /* Pal3DModem(use_sin=False): bit 0, (use_cos=False): bit 1 (pal.py:132-142) */
    double carrier_phase_step; /* qam.py:15 */
    /* SECAM: secam.py:156-162 normalised values + variant m0/kn/kd */
    double fsc_dr, fsc_db, fdev_dr, fdev_db, flimit_min, flimit_max, bell_f0, m0, bell_kn, bell_kd;
    double fm_fc; /* secam.py:179,187: centre passed to FmDecoder */
    orc_filter_t filters[ORC_MAX_FILTERS];
} orc_desc_t;

typedef struct orc_modem orc_modem;

orc_modem *orc_create(const orc_desc_t *desc);
void orc_destroy(orc_modem *m);
const char *orc_last_error(void);

int orc_modulation_delay(const orc_modem *m);
int orc_demodulation_delay(const orc_modem *m);

/* The duck-typed Modem protocol, one row per call, stateful exactly
   like the reference objects. n = row length. */
int orc_modulate(orc_modem *m, int frame, int line, const double *r, const double *g, const double *b, int n,
                 double *composite);
int orc_demodulate(orc_modem *m, int frame, int line, const double *composite, int n, double *r, double *g,
                   double *b);

/* component-level rows: (y, u, v) / (luma, dr, db) instead of (r, g, b); -1 if the stack has no such member */
int orc_modulate_components(orc_modem *m, int frame, int line, const double *y, const double *u, const double *v, int n,
                            double *composite);
int orc_demodulate_components(orc_modem *m, int frame, int line, const double *composite, int n, int strip_chroma,
                              double *y, double *u, double *v);

/* image.py:47-55 / 75-83 row schedule without the uint8 conversion.
   rgb is planar [3][H][W], composite [H][W]. */
int orc_modulate_frame(orc_modem *m, int frame, const double *rgb, double *composite);
int orc_demodulate_frame(orc_modem *m, int frame, const double *composite, double *rgb);

/* image.py:27-84 including the level mapping (:16-25) and _as_bytes (:7-8).
   rgb8 is interleaved [H][W][3], comp8 [H][W]. */
int orc_image_modulate(orc_modem *m, int frame, const uint8_t *rgb8, uint8_t *comp8);
int orc_image_demodulate(orc_modem *m, int frame, const uint8_t *comp8, uint8_t *rgb8);

/* float32 batch drivers used as the timed CPU baseline and as the checker for the HIP batch
   path: frames first_frame .. first_frame+n_frames-1, frame-sharded over n_threads threads
   (each thread owns a private modem built from desc). */
int orc_demodulate_frames_f32(const orc_desc_t *desc, const float *composite, float *rgb, int64_t n_frames,
                              int64_t first_frame, int n_threads);
int orc_modulate_frames_f32(const orc_desc_t *desc, const float *rgb, float *composite, int64_t n_frames,
                            int64_t first_frame, int n_threads);

/* exposed primitives (for unit tests against scipy) */
void orc_firwin41(double *h41);                                   /* scipy.signal.firwin(41, .5, window=('kaiser', 5.)) */
void orc_resample_up2(const double *x, int n, double *y);         /* resample_poly(x, 2, 1): 2n out */
int orc_resample_dn2(const double *x, int n, double *y);          /* resample_poly(x, 1, 2): returns len */
void orc_filter_apply(const orc_filter_t *f, const double *x, int n, double *y); /* utils.py:28-36 */
double orc_start_phase(const orc_desc_t *d, int frame, int line); /* utils.py:82-88 */
int orc_analog_line(const orc_desc_t *d, int line);               /* line.py:57-62 */
int orc_is_alternate_line(const orc_desc_t *d, int frame, int line); /* line.py:64-65 */

#ifdef __cplusplus
}
#endif
#endif

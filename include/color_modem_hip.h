/*
 * color_modem_hip.h - C ABI of libcolor_modem_hip.so, the MI355X (gfx950) implementation of
 * kFYatek/color_modem's per-line colour modulate/demodulate hot path.
 *
 * The reference has no FFI: its hot path is Python calling numpy/scipy once per scan line from
 * ImageModem's row loops.  The entry points below are what a binding for that path needs - one
 * immutable *plan* per modem stack (the state the reference builds in its constructors) and
 * batch calls that replace the per-row loops:
 *
 *   cm_plan_create        <- the constructors: qam.py:14-18 (QamColorModem), pal.py:28-31,
 *                            63-69, 131-178 (PalS/PalD/Pal3D), ntsc.py:23-26, 52-59, comb.py:
 *                            24-31, 72-88, 126-127, 131-139, secam.py:153-190
 *   cm_demodulate_frames  <- ImageModem.demodulate's row loop, image.py:75-83, calling
 *                            Modem.demodulate(frame, line, composite) (comb.py:67-68, 121-122;
 *                            qam.py:71-72; secam.py:278-304) for every row of every frame
 *   cm_modulate_frames    <- ImageModem.modulate's row loop, image.py:47-55, calling
 *                            Modem.modulate(frame, line, r, g, b) (qam.py:68-69, comb.py:154-155,
 *                            secam.py:258-259)
 *   cm_demodulate_run /   <- the same Modem.demodulate / Modem.modulate protocol for an explicit
 *   cm_modulate_run          run of consecutive same-field lines (what the stateful per-row
 *                            objects of the reference see between two resets)
 *
 * Conventions: plain C, no exceptions; every function returns CM_OK (0) or a negative code and
 * records a message for cm_last_error() (thread-local).  All image buffers are DEVICE pointers
 * to float32, caller-owned, row-major:  composite [frames][height][width],
 * rgb [frames][3][height][width] (planar R, G, B).  `stream` is a hipStream_t (NULL = default
 * stream); calls are asynchronous with respect to the host.  Images are dense (no row padding) whatever the width;
 * when the width is not a multiple of 4 the float entry points stage them through pitched device buffers
 * (stream-ordered allocation + two strided copies on `stream`), the byte entry points report CM_ERR_UNSUPPORTED.
 * Plans are immutable after creation and may be shared by threads; a plan belongs to the device that was
 * current when it was created.
 *
 * There is no CPU implementation behind this ABI: without a usable HIP device every compute
 * entry point fails with CM_ERR_NO_DEVICE.
 */
#ifndef COLOR_MODEM_HIP_H
#define COLOR_MODEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM_ABI_VERSION 8

enum cm_status {
    CM_OK = 0,
    CM_ERR_INVALID = -1,     /* bad argument / descriptor (ValueError on the Python side) */
    CM_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for */
    CM_ERR_NO_DEVICE = -3,   /* no HIP device / HIP runtime failure */
    CM_ERR_LAUNCH = -4       /* kernel launch or memory operation failed */
};

/* which streaming pipeline runs the main pass */
enum cm_pipeline {
    CM_PIPE_QAM = 1,   /* synchronous QAM detector (qam.py:43-58): PAL-S, NTSC, NTSC comb, Pal3D, Simple*Comb */
    CM_PIPE_PAL_D = 2, /* PAL delay-line decoder (pal.py:71-127) */
    CM_PIPE_SECAM = 3  /* SECAM FM (secam.py:127-149, 240-304) */
};

enum cm_chroma_average {
    CM_AVG_FOLDED = 0, /* none, or the arithmetic mean (comb.py:9-10): folded into the lane tables by linearity */
    CM_AVG_MIN = 1     /* comb.py:13-15 minavg: sign-aware minimum of two linear combinations */
};

#define CM_MAX_SECTIONS 4
#define CM_LANE_DOUBLES 32

/* One IIR filter of the reference (utils.py:9-26) in cascade form.
 * sos rows follow scipy: [b0 b1 b2 1 a1 a2]; shift is FilterFunction._shift. */
typedef struct {
    int32_t n_sections;
    int32_t shift;
    double sos[CM_MAX_SECTIONS][6];
} cm_iir_desc;

/* Per-(frame mod cycle, regime, line) constants of one pass; CM_LANE_DOUBLES doubles each:
 *   [0] sin, [1] cos of the detector phase at the first 2x sample of the line (informative; already folded
 *            into [4..15])
 *   [2] sin, [3] cos of the re-modulation phase; both 0 = luma passes unstripped
 *   [4..9]   u = sum_j t[4+2j] * Rs[k-j] + t[5+2j] * Rc[k-j],  j = 0..2
 *   [10..15] v likewise
 *   [16]     +1 / -1: sign applied to v on re-modulation (the PAL V switch, pal.py:50-51)
 *   [17..19] reserved (0)
 *   [20..31] chroma_average == CM_AVG_MIN only: a second (u, v) coefficient set laid out like [4..15]; the decoder
 *            outputs minavg(set A, set B) (comb.py:13-15, 103-104).  0 otherwise.
 * where (Rs, Rc)[k] is the phase-free base demodulation of call k's own input line: the detector chain of
 * qam.py:45-54 (CM_PIPE_QAM) or of pal.py:71-77 applied to qam.py:34-37 (CM_PIPE_PAL_D) run with the carriers
 * sin(m cps) / cos(m cps) and without its gains; the line's carrier phase is a rotation of that pair.
 * regime = min(k, 2), k = index of the call within its run (0 = first line after a reset). */
typedef struct {
    int32_t frame_cycle;  /* table rows per regime: frames repeat with this period */
    int32_t n_lines;      /* line numbers 0 .. n_lines-1 are tabulated */
    const double *table;  /* [frame_cycle][3][n_lines][CM_LANE_DOUBLES]; NULL = pass absent */
    int32_t luma_from_prev; /* per regime (bit r): luma source is the previous call's input line; bit 8 + r (wrap_mode != 0): of the
                             * call before that one */
    int32_t wrap_mode;      /* demod_main only, ABI 8.  0: the tables are the whole decoder.  1 / 2: a TWO-LEVEL comb - SimpleCombModem /
                             * Simple3DCombModem around Pal3DModem (comb.py:96-113 over pal.py:180-234): the tables are the inner decoder's
                             * (components, strip_chroma = False) except [2], [3], [16] and luma_from_prev, which describe the wrapper's
                             * strip (comb.py:105-106); the kernel combines every call's (u, v) with the previous call's by comb.avg (1) or
                             * comb.minavg (2) - a run's first call passes through (comb.py:97-99) - before it strips, notches and
                             * applies the matrix.  cm_plan_desc.depth counts the wrapper's line (3), frames entry points only. */
} cm_lane_table;

/* SECAM constants (secam.py:153-190), frequencies normalised to the Nyquist rate like the reference.
 * Lane tables of a CM_PIPE_SECAM plan (CM_LANE_DOUBLES doubles per (frame mod cycle, regime, line)):
 *   demod_main: [0] fsc, [1] fdev of the colour-difference signal this line carries, [2] 1 = Db line
 *               (LineConfig.is_alternate_line), [3] 0 on the first call of a run (last_chroma = 0) else 1
 *   mod_main:   [0] fsc, [1] fdev, [2] 1 = Db line, [3] start phase (0 or pi, secam.py:248-256, 273),
 *               [4] luma weight of the call's own row, [5] of the previous call's row, [6], [7] chroma
 *               weights likewise (comb.py:141-152); all for the line that is actually modulated */
#define CM_SECAM_PRESENT 1  /* cm_secam_desc.present: the constants below are filled in */
#define CM_SECAM_FLOAT64 2  /* ... | this: run the decoder's chroma front end (band-pass .. (I, Q) low-pass) in float64 whatever the
                             * shape - the library selects it by itself where float32 rounding noise would come near 1e-5
                             * (cm_api.hip: create_secam), at about half the throughput */
typedef struct {
    int32_t present;       /* CM_SECAM_PRESENT [| CM_SECAM_FLOAT64] */
    int32_t preroll;       /* len(composite) // 40 - 1 mirrored samples in front of the chroma band-pass (secam.py:283) */
    double flimit_min, flimit_max, bell_f0, m0, bell_kn, bell_kd;
    double fm_fc;          /* FmDecoder centre (secam.py:179, 187) */
    cm_iir_desc pre_lp;    /* secam.py:171-172 */
    cm_iir_desc lf_pre;    /* secam.py:175-177 forward (absent: n_sections = 0) */
    cm_iir_desc lf_rev;    /* secam.py:175-177 backward */
    cm_iir_desc bell;      /* secam.py:168-170 */
    cm_iir_desc chroma_bp; /* secam.py:183-184 */
    cm_iir_desc luma_bs;   /* secam.py:185-186 */
    cm_iir_desc fm_lp;     /* secam.py:131-132 */
} cm_secam_desc;

typedef struct {
    int32_t abi_version;   /* CM_ABI_VERSION */
    int32_t pipeline;      /* enum cm_pipeline of the main pass */
    int32_t width, height; /* W, H of a frame */
    int32_t demodulation_delay; /* image.py:63 */
    int32_t modulation_delay;   /* image.py:30 */
    int32_t depth;         /* how many previous calls of a run a demodulated line depends on (0..2) */
    int32_t first_is_plain;/* 1: calls with k == 0 come from the plain band-stop decoder (comb.py:48-49), table demod_first */
    int32_t main_luma_bandstop; /* 1: the main pass takes luma from the band-stop path (qam.py:57): plain PAL-S / NTSC */
    int32_t skip_calls;    /* 0, or 2: the main pass leaves the calls k < 2 of every run to another launch - the fused wrapped combs
                              (cm_comb_wrap_demodulate_frames_fused: PAL-D front end, depth 2, first_is_plain 0) */
    double carrier_phase_step; /* qam.py:15 */
    double resample_fir[41];   /* scipy.signal.firwin(41, 0.5, window=('kaiser', 5.0)) */
    cm_iir_desc extract2x;  /* qam.py:17 band-pass */
    cm_iir_desc remove2x;   /* qam.py:17 band-stop */
    cm_iir_desc demod_lp;   /* qam.py:18 */
    cm_iir_desc pald_lp;    /* pal.py:67-69 (CM_PIPE_PAL_D only) */
    cm_iir_desc precorrect; /* qam.py:16 */
    double decode_matrix[9]; /* (r, g, b) = M (y, u, v): pal.py:43-45, ntsc.py:38-40 */
    double encode_matrix[9]; /* (y, u, v) = M (r, g, b): pal.py:35-37, ntsc.py:30-32 */
    cm_lane_table demod_main;  /* main pass */
    cm_lane_table demod_first; /* plain pass for k == 0 (only regime 0 is read) */
    cm_secam_desc secam;       /* CM_PIPE_SECAM only */
    cm_lane_table mod_main;    /* modulator: [0] sin, [1] cos of the start phase of the modulated line, [2] luma weight of the
                                  call's own row, [3] of the previous call's row, [4], [5] chroma weights likewise
                                  (comb.py:141-152), [6] V-switch sign */
    /* Sub-carrier phase cycles too long to tabulate per frame (utils.py:78-80 yields e.g. 4800 frames for 4.43 MHz
     * colour on 525 lines): when frame_rotation is not NULL every lane table above holds exactly two frames -
     * frame numbers 0 and 1, i.e. both parities of LineConfig.is_alternate_line - and frame F uses table row F % 2
     * with all its phases advanced by the angle whose {cos, sin} is frame_rotation[2 * (F % frame_rotation_cycle)],
     * which the kernels apply to the lane constants when a lane starts.  NULL: tables are indexed by F % frame_cycle. */
    const double *frame_rotation;
    int32_t frame_rotation_cycle; /* even */
    int32_t chroma_average;       /* enum cm_chroma_average */
    cm_iir_desc notch;            /* comb.py:18-20 luma notch after the chroma strip (n_sections 0 = none); shift must be 0 */
} cm_plan_desc;

typedef struct cm_plan cm_plan;

const char *cm_last_error(void);
int cm_abi_version(void);

/* Number of usable HIP devices (0 when there is none); never fails. */
int cm_device_count(void);

int cm_plan_create(const cm_plan_desc *desc, cm_plan **out);
void cm_plan_destroy(cm_plan *plan);

/* Demodulate n_frames frames; frame numbers first_frame .. first_frame + n_frames - 1.
 * Equals looping ImageModem.demodulate's schedule (image.py:75-83) with a fresh modem. */
int cm_demodulate_frames(const cm_plan *plan, const float *composite, float *rgb, int64_t n_frames,
                         int64_t first_frame, void *stream);
int cm_modulate_frames(const cm_plan *plan, const float *rgb, float *composite, int64_t n_frames,
                       int64_t first_frame, void *stream);

/* The same with ImageModem's byte boundary fused in (image.py:58-84): composite8 is uint8 [frames][height][width]
 * (PIL mode 'L'), decoded as (5 * (byte / 255) - 1) / 3 (image.py:24-25, 62); rgb8 is interleaved uint8
 * [frames][height][width][3] (PIL mode 'RGB') = rint(255 * clip(x, 0, 1)) (image.py:7-8).  4 bytes per pixel
 * cross HBM instead of 16.  Every decoder (PAL / NTSC / SECAM). */
int cm_demodulate_frames_u8(const cm_plan *plan, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames,
                            int64_t first_frame, void *stream);

/* The encoder side of the same boundary (image.py:27-56): rgb8 is interleaved uint8 [frames][height][width][3] (PIL mode
 * 'RGB'), entering as byte / 255 (image.py:43-45); composite8 is uint8 [frames][height][width] (PIL mode 'L') =
 * rint(255 * clip(0.6 x + 0.2, 0, 1)) (encode_composite_level image.py:20-21, _as_bytes image.py:7-8).  Every encoder;
 * width must be a multiple of 16 (rows are moved as 16-byte vectors), else CM_ERR_UNSUPPORTED. */
int cm_modulate_frames_u8(const cm_plan *plan, const uint8_t *rgb8, uint8_t *composite8, int64_t n_frames,
                          int64_t first_frame, void *stream);

/* One run: n_calls consecutive calls Modem.demodulate(frame, first_line + 2 i, composite[i]),
 * i = 0 .. n_calls-1, where the first of them is the k0-th call since the modem's last reset
 * (k0 = 0: the run starts with a reset).  composite is [n_calls][width]; rgb receives what each
 * call returns, [n_calls][3][width].  Calls whose history (depth lines) lies before the run
 * start return unspecified data; the caller supplies enough history. */
int cm_demodulate_run(const cm_plan *plan, const float *composite, float *rgb, int32_t n_calls, int32_t frame,
                      int32_t first_line, int32_t k0, void *stream);
int cm_modulate_run(const cm_plan *plan, const float *rgb, float *composite, int32_t n_calls, int32_t frame,
                    int32_t first_line, int32_t k0, void *stream);

/* ---- D2-MAC style time-multiplex modem (ref color_modem/color/mac.py:16-125) --------------------
 * MacModem(line_config, variant_or_width): rows of `width` samples are brought to 720 (luma) and 360 (one colour-
 * difference signal per line) samples (mac.py:49-55), time-multiplexed into the 1080-sample line of mac.py:57-69 and
 * brought to `line_width` samples (mac.py:71-74: 1080 = D2MAC_12MHZ, 720 = D2MAC_7MHZ, or any number); the decoder
 * inverts this and always returns rows of 720 samples (mac.py:84-125).  `averaging` = the encoder sits inside
 * ColorAveragingModem (comb.py:130-152: modulation_delay 1).  720-sample rows <-> 1080-sample lines run on the tuned
 * kernels, everything else on the resampling ones. */
#define CM_MAC_LUMA_WIDTH 720
#define CM_MAC_LINE_WIDTH 1080
typedef struct cm_mac_fir {       /* one scipy.signal.resample_poly(x, up, down) of the path */
    int32_t up, down;             /* reduced fraction; up == down == 1: no resampling, taps ignored */
    int32_t n_taps;               /* 2 * 10 * max(up, down) + 1 */
    int32_t reserved;
    const double *taps;           /* up * firwin(n_taps, 1 / max(up, down), window=('kaiser', 5.0)) */
} cm_mac_fir;
typedef struct cm_mac_desc {
    int32_t width;                /* samples per rgb row (1 .. 1920) */
    int32_t height;               /* rows per frame */
    int32_t line_width;           /* samples per transmitted line (1 .. 4096) */
    int32_t line_shift;           /* LineConfig._line_shift (line.py:53) */
    int32_t even_first, odd_first;/* LineStandard.even_field_first_active_line / odd_... (line.py:56-60) */
    int32_t averaging;            /* 1: ColorAveragingModem(MacModem) on the encoder side */
    int32_t reserved;
    double resample_fir[41];      /* firwin(41, 0.5, ('kaiser', 5.0)): the decoder's chroma 360 -> 720, as in cm_plan_desc */
    double decode_matrix[9];      /* (r, g, b) = M . (luma, dr, db), mac.py:38-41 (identity: the *_components protocol) */
    double encode_matrix[9];      /* (luma, dr, db) = M . (r, g, b), mac.py:29-32 */
    cm_mac_fir luma_in;           /* width -> 720        mac.py:49-52 */
    cm_mac_fir chroma_in;         /* width -> 360        mac.py:53-55 */
    cm_mac_fir line_out;          /* 1080 -> line_width  mac.py:71-74 */
    cm_mac_fir line_in;           /* line_width -> 1080  mac.py:88-91 */
} cm_mac_desc;
typedef struct cm_mac_plan cm_mac_plan;

int cm_mac_plan_create(const cm_mac_desc *desc, cm_mac_plan **out);
void cm_mac_plan_destroy(cm_mac_plan *plan);

/* rgb [n_frames][3][height][width] -> composite [n_frames][height][line_width]; equals ImageModem.modulate's row
 * schedule (image.py:47-55) over frames first_frame .. with a fresh modem per frame. */
int cm_mac_modulate_frames(const cm_mac_plan *plan, const float *rgb, float *composite, int64_t n_frames,
                           int64_t first_frame, void *stream);
/* composite [n_frames][height][line_width] -> rgb [n_frames][3][height][720] (image.py:75-83). */
int cm_mac_demodulate_frames(const cm_mac_plan *plan, const float *composite, float *rgb, int64_t n_frames,
                             int64_t first_frame, void *stream);
/* The same with ImageModem's byte boundary fused in (image.py:7-8, 20-25, 43-45, 62): rgb8 interleaved uint8
 * [n_frames][height][width][3] -> composite8 uint8 [n_frames][height][line_width] -> rgb8 [n_frames][height][720][3]. */
int cm_mac_modulate_frames_u8(const cm_mac_plan *plan, const uint8_t *rgb8, uint8_t *composite8, int64_t n_frames,
                              int64_t first_frame, void *stream);
int cm_mac_demodulate_frames_u8(const cm_mac_plan *plan, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames,
                                int64_t first_frame, void *stream);
/* One run of n_calls consecutive calls (lines first_line, first_line + 2, ...), the first being the k0-th call since
 * the modem's reset; rows [n_calls][3][width] / [n_calls][line_width].  Row 0 has no history inside the buffers: with
 * k0 > 0 its output is unspecified (the caller submits one row of history, as for cm_demodulate_run). */
int cm_mac_modulate_run(const cm_mac_plan *plan, const float *rgb, float *composite, int32_t n_calls, int32_t frame,
                        int32_t first_line, int32_t k0, void *stream);
int cm_mac_demodulate_run(const cm_mac_plan *plan, const float *composite, float *rgb, int32_t n_calls, int32_t frame,
                          int32_t first_line, int32_t k0, void *stream);


/* ---- amplitude-modulated line-sequential standards ------------------------------------------------
 * CM_AM_PROTO_SECAM: ProtoSecamModem, ref color_modem/color/protosecam.py:27-112 (the 1957 819-line prototype: one colour-
 *   difference signal per line as the amplitude of a sub-carrier on a pedestal; decoder = envelope detector), optionally
 *   inside ColorAveragingModem on the encoder side (comb.py:130-167, `averaging`).
 * CM_AM_NIIR: NiirModem / HueCorrectingNiirModem, ref color_modem/color/niir.py:10-202 (SECAM-IV: saturation as amplitude,
 *   hue as the phase against the previous line's reference), `averaging` = the hue-correcting encoder (niir.py:167-202).
 * Both run every recursive filter of the decoder at three times the sampling rate between scipy.signal.resample_poly(x, 3, 1)
 * and resample_poly(x, 1, 3) (61-tap Kaiser(5) FIR, `resample_fir3`).  Filters are given as in the cm_plan_desc structure: second-order
 * sections + FilterFunction shift.  The sub-carrier start phase of a line is computed on the device in float64 from
 * frame_phase_shift / line_phase_shift / frame_cycle (utils.py:67-88) and the line geometry (line.py:57-65). */
enum cm_am_kind { CM_AM_PROTO_SECAM = 1, CM_AM_NIIR = 2 };
/* cm_am_desc.flags.  CM_AM_FLOAT64: accepted and ignored since round 4 (ABI v7).  The NIIR decoder takes the hue as the angle of a decimated
 * product pair and divides by its length (niir.py:131-137); with a float32 front end that left isolated samples beyond 1e-5 of full scale
 * wherever the pair gets short (4e-5 of the samples of random pictures, worst 4e-3), and round 3 offered a float64 front end behind this flag
 * (row-parallel kernel only, 9.5 Gpixel/s).  Now EVERY NIIR decoder - the streaming wave pair and the row-parallel kernel, floats or bytes -
 * runs the whole hue path in float64 (interpolator, band-pass, low-pass, the quotient M / S, the hue products and their decimators;
 * csrc/cm_am_stages.h: NiirHue): worst sample 5e-7 of full scale, 63 Gpixel/s on long batches (profiles/r04_niir_notes.txt).  The NIIR
 * ENCODERS form (db, dr) and the pedestal in float64 in the reference's own operation order where a pixel's saturation is below 1e-2. */
#define CM_AM_FLOAT64 1
typedef struct cm_am_desc {
    int32_t abi_version;          /* CM_ABI_VERSION */
    int32_t kind;                 /* enum cm_am_kind */
    int32_t width, height;
    int32_t line_shift;           /* LineConfig._line_shift (line.py:53) */
    int32_t even_first, odd_first;/* LineStandard.even_field_first_active_line / odd_... (line.py:56-60) */
    int32_t averaging;            /* 1: encoder inside ColorAveragingModem (Proto-SECAM) / HueCorrectingNiirModem (NIIR): modulation_delay 1 */
    int32_t premod_luma_filter;   /* Proto-SECAM encoder: protosecam.py:82-85 */
    int32_t frame_cycle;          /* ConstantFrequencyCarrier.frame_cycle (utils.py:78-80) */
    int32_t strip_chroma;         /* NIIR decoder: 0 = demodulate_components(..., strip_chroma=False) (niir.py:145), else 1 */
    int32_t flags;                /* 0 (CM_AM_FLOAT64 is accepted and ignored) */
    double frame_phase_shift;     /* ... .frame_shift (utils.py:74-76) */
    double line_phase_shift;      /* ... .line_shift (utils.py:69-72) */
    double carrier_phase_step;    /* radians per 1x sample: 2 * protosecam.py:31 _carrier_phase_step; niir.py:12 */
    double resample_fir3[61];     /* scipy.signal.firwin(61, 1 / 3, window=('kaiser', 5.0)) */
    cm_iir_desc precorrect;       /* chroma pre-correction low-pass at 1x (protosecam.py:33-34, niir.py:17-18) */
    cm_iir_desc bandpass_up;      /* at 3x: _extract_chroma_up (protosecam.py:36-39) / _demodulate_upsampled_filter (niir.py:21-24) */
    cm_iir_desc bandstop_up;      /* at 3x: _remove_chroma_up (protosecam.py:36-39); NIIR: unused */
    cm_iir_desc lowpass_up;       /* at 3x: _chroma_up_post_demod_filter (protosecam.py:45-48) / _demodulate_upsampled_baseband_filter */
    double bandpass_phase_shift;  /* NIIR: _demodulate_upsampled_filter.phase_shift (niir.py:157) */
    double decode_matrix[9];      /* (r, g, b) = M (c0, c1, c2): protosecam.py:63-69 (luma, dr, db), niir.py:52-61 (luma, db, dr) */
    double encode_matrix[9];
} cm_am_desc;
typedef struct cm_am_plan cm_am_plan;

int cm_am_plan_create(const cm_am_desc *desc, cm_am_plan **out);
void cm_am_plan_destroy(cm_am_plan *plan);
/* rgb [n_frames][3][height][width] -> composite [n_frames][height][width] (image.py:47-55) and back (image.py:75-83) */
int cm_am_modulate_frames(const cm_am_plan *plan, const float *rgb, float *composite, int64_t n_frames, int64_t first_frame,
                          void *stream);
int cm_am_demodulate_frames(const cm_am_plan *plan, const float *composite, float *rgb, int64_t n_frames, int64_t first_frame,
                            void *stream);
/* ImageModem's byte boundary fused into the kernels, as cm_modulate_frames_u8 / cm_demodulate_frames_u8 (image.py:27-56,
 * 58-84): interleaved 'RGB' bytes [n_frames][height][width][3] <-> 'L' bytes [n_frames][height][width].  The decoders need
 * width % 4 == 0, the encoders width % 16 == 0 (CM_ERR_UNSUPPORTED otherwise); the noisy NIIR encoder has no byte form. */
int cm_am_modulate_frames_u8(const cm_am_plan *plan, const uint8_t *rgb8, uint8_t *composite8, int64_t n_frames,
                             int64_t first_frame, void *stream);
int cm_am_demodulate_frames_u8(const cm_am_plan *plan, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames,
                               int64_t first_frame, void *stream);
/* One run of n_calls consecutive calls, as cm_demodulate_run / cm_modulate_run (one row of history in front when k0 > 0). */
int cm_am_modulate_run(const cm_am_plan *plan, const float *rgb, float *composite, int32_t n_calls, int32_t frame,
                       int32_t first_line, int32_t k0, void *stream);
int cm_am_demodulate_run(const cm_am_plan *plan, const float *composite, float *rgb, int32_t n_calls, int32_t frame,
                         int32_t first_line, int32_t k0, void *stream);
/* NiirModem(noise_level != 0) (niir.py:45-46; HueCorrectingNiirModem: niir.py:193-194): the encoder perturbs the hue with
 * (numpy.random.random_sample(W) - 0.5) * noise_level, drawn for db, then for dr, once per modulate() call.  The caller draws
 * them in the reference's call order - the order of the flattened [frame][field][call] list, the warm-up calls of
 * modulation_delay included - and passes them as noise [calls][2][width] float32 (device memory); everything else as above.
 * (modulate_components of the plain NiirModem adds no noise: niir.py:82-83.) */
int cm_am_modulate_frames_noise(const cm_am_plan *plan, const float *rgb, const float *noise, float *composite, int64_t n_frames,
                                int64_t first_frame, void *stream);
int cm_am_modulate_run_noise(const cm_am_plan *plan, const float *rgb, const float *noise, float *composite, int32_t n_calls,
                             int32_t frame, int32_t first_line, int32_t k0, void *stream);

/* ---- SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem (comb.py:71-127 over pal.py:62-234) -----------------
 * Replaces SimpleCombModem.demodulate_components / demodulate (comb.py:96-122) when the backend is a PAL delay-line decoder.
 * These stacks run as two streaming kernels per batch (csrc/cm_wrap_kernels.h): the inner decoder in
 * component mode with strip_chroma = False over all calls of the batch (what the wrapper asks its backend for, comb.py:97, 101),
 * then the wrapper's own arithmetic - averaging of consecutive calls' chroma and the luma source (comb.py:102-104), the luma
 * strip by re-modulating the averaged chroma (comb.py:105-107: the backend modulator's pre-correction filter and carrier), the
 * notch (comb.py:108-110) and decode_components (comb.py:121-122).
 *   inner    plan of the wrapped decoder (PalDModem / Pal3DModem) built for the component protocol with strip_chroma = False
 *   first    plan of the plain decoder (PalSModem, components, strip_chroma = False) when the inner decoder takes call 0 of a
 *            run from it (comb.py:48-49: cm_plan_desc.first_is_plain), else NULL
 *   backend  plan of the plain modem (PalSModem, components): its modulator constants are used
 * The plans' lane tables must cover line numbers up to height - 1 + 2 (inner demodulation_delay + own_delay). */
typedef struct {
    int32_t own_delay;      /* 1: Simple3DCombModem / SimpleCombModem(delay=True) (comb.py:74, 126) */
    int32_t minavg;         /* 1: avg=comb.minavg (comb.py:13-15), 0: comb.avg (comb.py:9-10), 2: the caller averaged the component buffer
                               (cm_comb_wrap_finish_*: avg= callables) */
    int32_t strip_chroma;   /* the flag of demodulate_components (comb.py:96); 1 for demodulate */
    int32_t reserved;
    cm_iir_desc notch;      /* the wrapper's notch= (comb.py:18-20, 86-88): one section, shift 0; n_sections = 0: none */
    double matrix[9];       /* decode_components (comb.py:121-122), row major; identity for the component protocol */
} cm_comb_wrap_desc;
/* composite [F][H][W] float32 -> rgb [F][3][H][W] float32, the row schedule of image.py:75-83 (device pointers).  Stream-ordered scratch of the
 * call's lifetime: the component buffer of up to 2 GiB (the batch is walked in chunks of that many frames), with bytes at the boundary the
 * level-decoded composite of one chunk beside it (+ a third). */
int cm_comb_wrap_demodulate_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                                   const float *composite, float *rgb, int64_t n_frames, int64_t first_frame, void *stream);
/* the same with the ImageModem byte boundary (image.py:58-84): 'L' bytes [F][H][W] -> interleaved 'RGB' bytes [F][H][W][3] */
int cm_comb_wrap_demodulate_frames_u8(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                                      const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames, int64_t first_frame, void *stream);
/* The two entry points above with a fourth plan for long batches around PalDModem: `fused` = the PAL-D front end with TWO lines of history
 * (cm_plan_desc: pipeline CM_PIPE_PAL_D, depth 2, first_is_plain 0, skip_calls 2, the wrapper's average / notch / delay folded into its lane
 * tables), which decodes every call k >= 2 of every run in ONE pass over the frames - no component scratch, 16 instead of 40 bytes per pixel
 * through HBM; the calls k < 2 (the top four rows of a frame: comb.py:97-99 and the first average, which mix in the plain decode) still go
 * through the composition.  Same results (both halves are the reference's arithmetic, comb.py:96-113 over pal.py:79-127).  Falls back to
 * the composition for short batches (the scan kernels' regime), pinned small-batch modes, heights < 8, widths that are not multiples
 * of 4, and when `fused` is NULL. */
int cm_comb_wrap_demodulate_frames_fused(const cm_plan *fused, const cm_plan *inner, const cm_plan *first, const cm_plan *backend,
                                         const cm_comb_wrap_desc *wrap, const float *composite, float *rgb, int64_t n_frames,
                                         int64_t first_frame, void *stream);
int cm_comb_wrap_demodulate_frames_fused_u8(const cm_plan *fused, const cm_plan *inner, const cm_plan *first, const cm_plan *backend,
                                            const cm_comb_wrap_desc *wrap, const uint8_t *composite8, uint8_t *rgb8, int64_t n_frames,
                                            int64_t first_frame, void *stream);
/* One run of n_calls consecutive calls as cm_demodulate_run: rows [n][W] -> [n][3][W]; a run submitted with k0 > 0 carries one
 * call of history in front (inner depth + 1 calls before the first wanted result). */
int cm_comb_wrap_demodulate_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                                const float *composite, float *rgb, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0,
                                void *stream);

/* FilterFunction.__call__ (/root/reference/color_modem/utils.py:28-36) on rows of float64 in device memory: y = lfilter(b, a, x padded
 * with `shift` copies of its last sample)[shift:] for shift > 0, lfilter(b, a, -shift copies of the first sample + x)[:shift] for
 * shift < 0, lfilter(b, a, x) for 0 - scipy.signal.lfilter's own recurrence (transposed direct form II, zero initial state, a[0]
 * normalised, unfused float64), one lane per row: a callable for REPL use (README.md:8-10 of the reference), not a throughput path.
 * x, y: [n_rows][width] doubles on the current device, not overlapping; b, a: host arrays of up to CM_FILTER_MAX_TAPS coefficients. */
#define CM_FILTER_MAX_TAPS 25
int cm_filter_rows_f64(const double *b, int32_t n_b, const double *a, int32_t n_a, int32_t shift, const double *x, double *y,
                       int64_t n_rows, int32_t width, void *stream);

/* The luma notch of a comb decoder as a pass of its own, for notch= values whose FilterFunction shift is not 0 (comb.py:18-20 with utils.py:9-26:
 * round(group delay at DC) is 1 for q = 1.0, and other values, also negative ones, below that; the fused kernels carry the notch at shift 0 only).
 * yuv_in: what the decoder returns in component form WITHOUT its notch, [group][3][rows_per_group][width] floats; yuv_out (another buffer)
 * receives decode_components(notch(y), u, v) (comb.py:54-55, 108-110, 121-122; pal.py:225-228): the luma rows with index >= skip_rows of
 * every group through FilterFunction.__call__ (utils.py:28-36, float64 inside), the other rows as they are (the first call of a run is never
 * notched: comb.py:48-49, 97-99), then `matrix` (row major; identity for the component protocol). */
int cm_notch_luma_f32(const double *b, int32_t n_b, const double *a, int32_t n_a, int32_t shift, const float *yuv_in, float *yuv_out,
                      int64_t n_groups, int64_t rows_per_group, int32_t width, int32_t skip_rows, const double *matrix, void *stream);

/* avg= callables (comb.py:72, 81-84: SimpleCombModem(avg=f) combines the chroma of consecutive calls with the caller's own function,
 * comb.py:103-104).  The composition cut in two: `components` receives what the inner decoder returns for every call of every run,
 * [frame][call][3 = y, u, v][W] floats in call order (frames entry points: cm_comb_wrap_calls_per_frame() calls per frame - both fields, each
 * with its delay calls; run entry points: [n_calls][3][W]); the caller replaces (u, v) of every call k >= 1 of a run by f(previous call's,
 * this call's) - taken from the buffer as it was filled - and hands the buffer to the second half with cm_comb_wrap_desc.minavg = 2, which
 * does the rest of comb.py:101-110 (luma source, re-modulation, strip, notch, matrix).  Widths that are multiples of 4; float rows only. */
int cm_comb_wrap_calls_per_frame(const cm_plan *inner, const cm_comb_wrap_desc *wrap);
int cm_comb_wrap_components_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                                   const float *composite, float *components, int64_t n_frames, int64_t first_frame, void *stream);
int cm_comb_wrap_finish_frames(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                               float *components, float *rgb, int64_t n_frames, int64_t first_frame, void *stream);
int cm_comb_wrap_components_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                                const float *composite, float *components, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0,
                                void *stream);
int cm_comb_wrap_finish_run(const cm_plan *inner, const cm_plan *first, const cm_plan *backend, const cm_comb_wrap_desc *wrap,
                            float *components, float *rgb, int32_t n_calls, int32_t frame, int32_t first_line, int32_t k0, void *stream);

/* Every compute entry point checks that the plan's device is the current one and that both image buffers are memory the HIP
 * runtime knows as accessible from it (CM_ERR_INVALID for another GPU's memory, pageable host memory and pointers the runtime
 * cannot classify; pinned / mapped host memory and managed memory pass).  cm_set_pointer_check(0) switches the pointer
 * classification off for the process - for allocators the runtime does not know - and cm_set_pointer_check(1) on again. */
void cm_set_pointer_check(int32_t on);
/* Small batches.  The streaming kernels give a scan line to a lane, so one launch lasts as long as one row takes however few
 * rows there are.  Below CM_SCAN_MAX_CALLS calls per launch (a few frames; the per-row protocol) the library therefore runs
 * the row-parallel kernel - one wavefront per scan line, the recursive filters as a scan over the lanes (csrc/cm_scan_kernels.h) -
 * where the plan's shape fits it (rows up to ~2000 samples, SECAM decoders below 1280; floats or bytes at the boundary; encoders up to
 * 40000 calls), else it cuts the rows into segments that workgroups walk side by side.  Results agree with the row walk to ~1e-7 of full scale (tests: test_small_batch_modes).  This switch pins
 * one of the three for tests and measurements; CM_ERR_UNSUPPORTED when the plan has no scan kernel.  It is the one call that changes a plan
 * after its creation (an atomic field: safe beside running calls, which pick the mode up at their next launch); production code leaves it alone. */
enum { CM_SMALL_BATCH_AUTO = 0, CM_SMALL_BATCH_ROWS = 1, CM_SMALL_BATCH_SEGMENTS = 2, CM_SMALL_BATCH_SCAN = 3 };
int cm_plan_set_small_batch(const cm_plan *plan, int32_t mode);
/* The same for a Proto-SECAM plan (csrc/cm_am_scan_kernels.h: protosecam.py:74-112 with one wavefront per scan line, rows up to
 * ~1000 samples, floats or bytes at the boundary) or a NIIR plan (niir.py:78-164 the same way); CM_SMALL_BATCH_SEGMENTS does not exist there. */
int cm_am_plan_set_small_batch(const cm_am_plan *plan, int32_t mode);
/* Name, main-loop instruction mix and launch geometry of the dominant kernel of the last
 * cm_demodulate_frames call on this plan (for bench.py / profiling); returns bytes written. */
int cm_plan_describe(const cm_plan *plan, char *buf, int32_t buf_len);

#ifdef __cplusplus
}
#endif
#endif

// cm_sim.cpp - TEST INFRASTRUCTURE.  Runs the streaming stages of color_modem_amd/csrc/cm_stages.h
// on the host (T = double: checks the schedule/index logic against the oracle at ~1e-12;
// T = float: predicts the float32 rounding error of the device kernels).  Never used by the
// product path.
#include <cstring>
#include <type_traits>
#include <string>
#include <vector>

#include "../../color_modem_amd/csrc/cm_plan.h"

using namespace cm;

static thread_local std::string g_err;

template <typename T>
struct RowCtx {
    LaneK<T> lk;
    std::vector<T> x, xl;  // own input row, luma source row
    bool plain;            // produced by the band-stop (first line) decoder
};

template <typename T>
static const double *lane_entry(const cm_lane_table &tb, int frame, int regime, int line) {
    int f = ((frame % tb.frame_cycle) + tb.frame_cycle) % tb.frame_cycle;
    return tb.table + (((size_t)f * 3 + regime) * tb.n_lines + line) * CM_LANE_DOUBLES;
}

// PIPE: 0 = QAM front, 1 = PAL-D front.  bsf: luma from the band-stop path.
template <typename T, class S>
static int run_generic(const cm_plan_desc &d, bool pald, bool bsf, const cm_lane_table &tb, const std::vector<int> &calls,
                       const double *comp, double *rgb, int n_calls, int frame, int first_line, int k0, bool mid_fast) {
    DemodK<T, S> k;
    DemodScales sc;
    if (!build_demod_k<T, S>(d, pald, bsf, k, sc, g_err)) return CM_ERR_UNSUPPORTED;
    const int W = d.width;
    std::vector<T> car = build_carrier<T>(d.carrier_phase_step, W);
    auto carrier = [&](int m, T *out2) {
        if (m < 0) m = 0;
        if (m > 2 * W - 1) m = 2 * W - 1;
        out2[0] = car[2 * m];
        out2[1] = car[2 * m + 1];
    };
    const int n = n_calls;
    std::vector<LaneK<T>> lk(n);
    std::vector<int> dy(n);
    for (int i = 0; i < n; ++i) {
        int kk = k0 + i, regime = kk < 2 ? kk : 2, line = first_line + 2 * i;
        if (line < 0 || line >= tb.n_lines) { g_err = "line outside the lane table"; return CM_ERR_INVALID; }
        lk[i] = convert_lane<T>(lane_entry<T>(tb, frame, regime, line), sc);
        dy[i] = (tb.luma_from_prev >> regime) & 1;
    }
    auto xin = [&](int i, int s) -> T {
        if (i < 0 || s < 0 || s >= W) return T(0);
        return T(comp[(size_t)i * W + s]);
    };
    typedef SysPal SP_;  // the PAL-D front end only exists for even shifts
    typedef typename std::conditional<S::ODD_E || S::ODD_L, SP_, S>::type SD;
    std::vector<PalDFront<T, SD>> fp(pald ? n : 0);
    std::vector<QamFront<T, S, true>> fq(n);
    std::vector<DemodBack<T, S, 2>> back(n);
    std::vector<FrontLatch<T>> fla(n);
    std::vector<BackLatch<T>> bla(n);
    for (int i = 0; i < n; ++i) { fla[i].reset(); bla[i].reset(); }
    for (int i = 0; i < n; ++i) { if (pald) fp[i].reset(); fq[i].reset(); back[i].reset(); }
    const int lat_front = pald ? (10 + k.q_e + 9 + 10 + k.q_l + 9) : QamFront<T, S, true>::latency(k);
    const int lat_luma = QamFront<T, S, true>::luma_latency(k);
    const int lat_total = lat_front + k.s_p;
    const int steps = W + lat_total;
    // histories (the device keeps these in small register windows / an LDS ring)
    std::vector<std::vector<T>> e_hist(n, std::vector<T>(steps + 64, T(0)));
    std::vector<std::vector<T>> u_hist(n, std::vector<T>(steps + 64, T(0))), v_hist(n, std::vector<T>(steps + 64, T(0)));
    std::vector<std::vector<T>> y_hist(n, std::vector<T>(steps + 64, T(0)));
    std::vector<Pair<T>> base(n);
    for (int t = 0; t < steps; ++t) {
        // the device runs the EDGE-free body where no stage touches a row boundary
        bool edge = !(mid_fast && t >= lat_total + 1 && t < W + 8);
        const int n1 = t - 10;
        for (int i = 0; i < n; ++i) {
            T x_now = xin(i, t), x_d10 = xin(i, t - 10);
            if (pald) {
                const int n3 = n1 - k.q_e - 9, n4 = n3 - 10;
                T cr[4];
                carrier(2 * n4, cr);
                carrier(2 * n4 + 1, cr + 2);
                T e_d10 = (n3 - 10 >= 0) ? e_hist[i][n3 - 10] : T(0);
                T e_out;
                const DemodK<T, SD> &kd = reinterpret_cast<const DemodK<T, SD> &>(k);  // same type whenever pald
                base[i] = edge ? fp[i].template step<true>(kd, fla[i], t, x_now, x_d10, e_d10, cr, e_out)
                               : fp[i].template step<false>(kd, fla[i], t, x_now, x_d10, e_d10, cr, e_out);
                if (n3 >= 0) e_hist[i][n3] = e_out;
            } else {
                const int n2 = n1 - k.q_e;
                T cr[4];
                carrier(2 * n2, cr);
                carrier(2 * n2 + 1, cr + 2);
                T luma = T(0);
                base[i] = edge ? fq[i].template step<true>(k, fla[i], t, x_now, x_d10, cr, luma)
                               : fq[i].template step<false>(k, fla[i], t, x_now, x_d10, cr, luma);
                const int nl = t - lat_luma;
                if (nl >= 0) y_hist[i][nl] = luma;
            }
        }
        const int n6 = t - lat_front, n7 = n6 - k.s_p;
        for (int i = 0; i < n; ++i) {
            Pair<T> z{T(0), T(0)};
            T u, v;
            back[i].combine(lk[i], base[i], i >= 1 ? base[i - 1] : z, i >= 2 ? base[i - 2] : z, u, v);
            if (n6 >= 0) { u_hist[i][n6] = u; v_hist[i][n6] = v; }
            T u_d = n7 >= 0 ? u_hist[i][n7] : T(0), v_d = n7 >= 0 ? v_hist[i][n7] : T(0);
            T y_src = bsf ? (n7 >= 0 ? y_hist[i][n7] : T(0)) : xin(i - dy[i], n7);
            T cr[2];
            carrier(2 * n7, cr);
            Rgb<T> o = edge ? back[i].template step<true>(k, lk[i], bla[i], n6, u, v, u_d, v_d, y_src, cr)
                            : back[i].template step<false>(k, lk[i], bla[i], n6, u, v, u_d, v_d, y_src, cr);
            bool wanted = false;
            for (int c : calls) wanted |= (c == i);
            if (wanted && n7 >= 0 && n7 < W) {
                rgb[((size_t)i * 3 + 0) * W + n7] = (double)o.r;
                rgb[((size_t)i * 3 + 1) * W + n7] = (double)o.g;
                rgb[((size_t)i * 3 + 2) * W + n7] = (double)o.b;
            }
        }
    }
    return CM_OK;
}

template <typename T>
static int run_dispatch(const cm_plan_desc &d, bool pald, bool bsf, const cm_lane_table &tb, const std::vector<int> &calls,
                        const double *comp, double *rgb, int n_calls, int frame, int first_line, int k0, bool mid_fast) {
    SysSignature want = signature_wanted(d, pald);
    if (!bsf) { want.nr = 0; want.odd_r = 0; }
    auto match = [&](SysSignature have) {
        if (!bsf) { have.nr = 0; have.odd_r = 0; }
        return same_signature(want, have);
    };
    if (match(signature_of<SysPal>()))
        return run_generic<T, SysPal>(d, pald, bsf, tb, calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
    if (!pald && match(signature_of<SysNtsc>()))
        return run_generic<T, SysNtsc>(d, pald, bsf, tb, calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
    if (!pald && match(signature_of<SysNtscI>()))
        return run_generic<T, SysNtscI>(d, pald, bsf, tb, calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
    if (!pald && match(signature_of<SysNtscA>()))
        return run_generic<T, SysNtscA>(d, pald, bsf, tb, calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
    if (fits_any(want))   // the run-time shape: identity-padded cascades, shift parities read at run time
        return run_generic<T, SysAny>(d, pald, bsf, tb, calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
    g_err = "no kernel instance for this filter set";
    return CM_ERR_UNSUPPORTED;
}

template <typename T>
static int sim_run(const cm_plan_desc *d, const double *comp, double *rgb, int n_calls, int frame, int first_line, int k0,
                   int mid_fast) {
    std::vector<int> main_calls, first_calls;
    for (int i = 0; i < n_calls; ++i) {
        if (d->first_is_plain && k0 + i == 0) first_calls.push_back(i); else main_calls.push_back(i);
    }
    int rc = CM_OK;
    if (!main_calls.empty()) {
        bool pald = d->pipeline == CM_PIPE_PAL_D;
        bool bsf = d->main_luma_bandstop != 0;  // plain decoders: every call uses the band-stop
        rc = run_dispatch<T>(*d, pald, bsf, d->demod_main, main_calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast != 0);
        if (rc) return rc;
    }
    if (!first_calls.empty()) rc = run_dispatch<T>(*d, false, true, d->demod_first, first_calls, comp, rgb, n_calls, frame, first_line, k0, mid_fast != 0);
    return rc;
}

// ---- SECAM decoder -------------------------------------------------------------------------------
template <typename T>
static int sim_secam_run(const cm_plan_desc *d, const double *comp, double *rgb, int n_calls, int frame, int first_line, int k0) {
    SecamDemodK<T> k;
    if (!build_secam_demod_k<T>(*d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const int W = d->width, P = k.preroll, Lc = W + P;
    std::vector<T> fm = build_fm_reference<T>(d->secam.fm_fc, Lc);
    std::vector<T> dc = build_fm_dc<T>(*d, Lc);
    const cm_lane_table &tb = d->demod_main;
    std::vector<SecamDemodLaneK<T>> lk(n_calls);
    for (int i = 0; i < n_calls; ++i) {
        int kk = k0 + i, regime = kk < 2 ? kk : 2, line = first_line + 2 * i;
        if (line < 0 || line >= tb.n_lines) { g_err = "line outside the lane table"; return CM_ERR_INVALID; }
        lk[i] = convert_secam_demod_lane<T>(lane_entry<T>(tb, frame, regime, line), d->secam);
    }
    std::vector<SecamDemod<T>> st(n_calls);
    for (auto &x : st) x.reset();
    const int lat = SecamDemod<T>::latency(k);
    const int steps = Lc + lat + 2;
    // the float32 kernels run the band-pass + bell of their guarded bodies in float64 (cm_stages.h: SecamBp64)
    SecamBp64 e64;
    if (!build_secam_bp64(*d, e64, g_err)) return CM_ERR_UNSUPPORTED;
    int xb0, xb1;
    secam_mid_bounds(W, P, lat, e64, xb0, xb1);
    const bool f32 = std::is_same<T, float>::value;
    std::vector<std::vector<T>> ch_hist(n_calls, std::vector<T>(steps + 64, T(0)));
    std::vector<T> own(n_calls), prev_own(n_calls, T(0));
    auto xin = [&](int i, int s) -> T { return (s < 0 || s >= W) ? T(0) : T(comp[(size_t)i * W + s]); };
    for (int m = 0; m < steps; ++m) {
        const int m1 = m - k.s_b, m2 = m1 - 10;
        for (int i = 0; i < n_calls; ++i) {
            T cc = m < P ? xin(i, P - m) : xin(i, m - P);
            int mc = m2 < 0 ? 0 : (m2 > Lc - 1 ? Lc - 1 : m2);
            T car[4] = {fm[4 * mc], fm[4 * mc + 1], fm[4 * mc + 2], fm[4 * mc + 3]};
            T ch_d10 = (m1 - 10 >= 0) ? ch_hist[i][m1 - 10] : T(0);
            T ch_out;
            int m4 = m - lat + P;
            m4 = m4 < 0 ? 0 : (m4 > Lc - 1 ? Lc - 1 : m4);
            const bool guarded = m - P < xb0 || m - P >= xb1;
            own[i] = st[i].chroma_step(k, lk[i], m, cc, ch_d10, car, dc[m4], ch_out, f32 && guarded ? &e64 : nullptr);
            if (m1 >= 0) ch_hist[i][m1] = ch_out;
        }
        // the back end runs one step behind (neighbour exchange), as on the device
        const int n = m - 1 - lat;
        for (int i = 0; i < n_calls; ++i) {
            T luma = st[i].luma_step(k, n, xin(i, n + k.s_y));
            Rgb<T> o = st[i].finish(k, lk[i], luma, prev_own[i], i >= 1 ? prev_own[i - 1] : T(0));
            if (n >= 0 && n < W) {
                rgb[((size_t)i * 3 + 0) * W + n] = (double)o.r;
                rgb[((size_t)i * 3 + 1) * W + n] = (double)o.g;
                rgb[((size_t)i * 3 + 2) * W + n] = (double)o.b;
            }
        }
        prev_own = own;
    }
    return CM_OK;
}

// ---- SECAM modulator (rows mode: rgb [n][3][W] -> composite [n][W]) -------------------------------------
template <typename T, typename TD>
static int sim_secam_mod(const cm_plan_desc *d, const double *rgb, double *comp, int n_calls, int frame, int first_line, int k0) {
    SecamModK<T, TD> k;
    if (!build_secam_mod_k<T, TD>(*d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const int W = d->width;
    const cm_lane_table &tb = d->mod_main;
    for (int i = 0; i < n_calls; ++i) {
        int kk = k0 + i, regime = kk < 2 ? kk : 2, line = first_line + 2 * i;
        if (line < 0 || line >= tb.n_lines) { g_err = "line outside the lane table"; return CM_ERR_INVALID; }
        SecamModLaneK<T, TD> lk = convert_secam_mod_lane<T, TD>(lane_entry<T>(tb, frame, regime, line));
        SecamMod<T, TD> st;
        st.reset();
        std::vector<T> luma(W + 16, T(0));
        for (int n = 0; n < W + k.s_p; ++n) {
            T comp3[2][3] = {{0, 0, 0}, {0, 0, 0}};  // own row, previous call's row
            for (int which = 0; which < 2; ++which) {
                int row = i - which < 0 ? i : i - which;
                if (n < W)
                    for (int c = 0; c < 3; ++c) {
                        T acc = T(0);
                        for (int p = 0; p < 3; ++p) acc += k.e[c][p] * T(rgb[((size_t)row * 3 + p) * W + n]);
                        comp3[which][c] = acc;
                    }
            }
            T y = lk.wy0 * comp3[0][0] + lk.wy1 * comp3[1][0];
            int sel = lk.own_is_db != T(0) ? 2 : 1;
            T dsel = lk.wc0 * comp3[0][sel] + lk.wc1 * comp3[1][sel];
            if (n < W) luma[n] = y;
            int n7 = n - k.s_p;
            T out = st.step(k, lk, n, n7 >= 0 ? luma[n7] : T(0), dsel);
            if (n7 >= 0 && n7 < W) comp[(size_t)i * W + n7] = (double)out;
        }
    }
    return CM_OK;
}

extern "C" {
int cm_sim_secam_modulate_run_f64(const cm_plan_desc *d, const double *rgb, double *comp, int n_calls, int frame, int first_line, int k0) {
    return sim_secam_mod<double, double>(d, rgb, comp, n_calls, frame, first_line, k0);
}
int cm_sim_secam_modulate_run_f32(const cm_plan_desc *d, const double *rgb, double *comp, int n_calls, int frame, int first_line, int k0) {
    return sim_secam_mod<float, double>(d, rgb, comp, n_calls, frame, first_line, k0);
}
}

extern "C" {
int cm_sim_secam_demodulate_run_f64(const cm_plan_desc *d, const double *comp, double *rgb, int n_calls, int frame, int first_line, int k0) {
    return sim_secam_run<double>(d, comp, rgb, n_calls, frame, first_line, k0);
}
int cm_sim_secam_demodulate_run_f32(const cm_plan_desc *d, const double *comp, double *rgb, int n_calls, int frame, int first_line, int k0) {
    return sim_secam_run<float>(d, comp, rgb, n_calls, frame, first_line, k0);
}
}

extern "C" {
const char *cm_sim_last_error(void) { return g_err.c_str(); }
int cm_sim_demodulate_run_f64(const cm_plan_desc *d, const double *comp, double *rgb, int n_calls, int frame, int first_line,
                              int k0, int mid_fast) {
    return sim_run<double>(d, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
}
int cm_sim_demodulate_run_f32(const cm_plan_desc *d, const double *comp, double *rgb, int n_calls, int frame, int first_line,
                              int k0, int mid_fast) {
    return sim_run<float>(d, comp, rgb, n_calls, frame, first_line, k0, mid_fast);
}
}

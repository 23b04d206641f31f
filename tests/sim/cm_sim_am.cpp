// cm_sim_am.cpp - TEST INFRASTRUCTURE.  Runs the streaming stages of color_modem_amd/csrc/cm_am_stages.h on the host
// (T = double: checks the schedule / index logic against the numpy oracle at ~1e-12; T = float: predicts the float32
// rounding error of the device kernels).  One "lane" per call of a run, exactly the per-step protocol of the kernels in
// cm_am_kernels.h, with the kernels' LDS delay rings replaced by plain arrays.  Never used by the product path.
#include <cstring>
#include <string>
#include <vector>

#include "../../color_modem_amd/csrc/cm_am_plan.h"

using namespace cm;

static thread_local std::string g_err;

extern "C" const char *am_sim_last_error() { return g_err.c_str(); }

// One run of n_calls consecutive calls (lines first_line, first_line + 2, ...), the first being the k0-th since a reset.
// comp [n_calls][W] -> rgb [n_calls][3][W].  Call 0 with k0 > 0 lacks its history (returns what zero history gives).
template <typename T>
static int proto_demod_run(const cm_am_desc &d, const double *comp, double *rgb, int n_calls, long long frame, int first_line, int k0) {
    ProtoDemodK<T> k;
    if (!build_proto_demod_k<T>(d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const AmLine ln = am_line(d);
    const int W = d.width;
    const int lat_c = ProtoDemod<T>::lat_chroma(k), lat_y = ProtoDemod<T>::lat_luma(k);
    if (lat_c < lat_y) { g_err = "chroma path shorter than the luma path"; return CM_ERR_UNSUPPORTED; }
    std::vector<std::vector<T>> chroma(n_calls, std::vector<T>(W)), luma(n_calls, std::vector<T>(W));
    for (int i = 0; i < n_calls; ++i) {
        ProtoDemod<T> st;
        st.reset();
        for (int t = 0; t < W + lat_c; ++t) {
            T l, c;
            st.step(k, t, t < W ? T(comp[(size_t)i * W + t]) : T(0), l, c);
            if (t - lat_y >= 0 && t - lat_y < W) luma[i][t - lat_y] = l;
            if (t - lat_c >= 0 && t - lat_c < W) chroma[i][t - lat_c] = c;
        }
    }
    for (int i = 0; i < n_calls; ++i) {
        const int line = first_line + 2 * i;
        const bool alt = ln.alternate(frame, line);
        const bool have_prev = (k0 + i) > 0 && i > 0;
        for (int n = 0; n < W; ++n) {
            const T prev = have_prev ? chroma[i - 1][n] : T(0);
            const T dr = alt ? prev : chroma[i][n], db = alt ? chroma[i][n] : prev;      // protosecam.py:105-108
            for (int p = 0; p < 3; ++p)
                rgb[((size_t)i * 3 + p) * W + n] = (double)(k.m[p][0] * luma[i][n] + k.m[p][1] * dr + k.m[p][2] * db);
        }
    }
    return CM_OK;
}

// rgb [n_calls][3][W] -> comp [n_calls][W].  averaging: ColorAveragingModem (comb.py:141-152): call i modulates line - 2
// with the previous call's luma and the mean of both calls' colour-difference signals.
template <typename T>
static int proto_mod_run(const cm_am_desc &d, const double *rgb, double *comp, int n_calls, long long frame, int first_line, int k0) {
    ProtoModK<T> k;
    if (!build_proto_mod_k<T>(d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const AmLine ln = am_line(d);
    const int W = d.width;
    const int lat_y = ProtoMod<T>::lat_luma(k), lat_c = ProtoMod<T>::lat_chroma(k);
    const int lat = lat_y > lat_c ? lat_y : lat_c;
    for (int i = 0; i < n_calls; ++i) {
        const int call_line = first_line + 2 * i;
        const int line = d.averaging ? call_line - 2 : call_line;        // the line that is modulated (comb.py:152)
        const bool alt = ln.alternate(frame, line);
        const double phi = ln.start_phase(frame, line);
        const bool have_prev = (k0 + i) > 0 && i > 0;
        auto comps = [&](int row, int n, T &y, T &dr, T &db) {
            const double *r = rgb + ((size_t)row * 3) * W;
            const T R = T(r[n]), G = T(r[W + n]), B = T(r[2 * W + n]);
            y = k.e[0][0] * R + k.e[0][1] * G + k.e[0][2] * B;
            dr = k.e[1][0] * R + k.e[1][1] * G + k.e[1][2] * B;
            db = k.e[2][0] * R + k.e[2][1] * G + k.e[2][2] * B;
        };
        auto source = [&](int n, T &y, T &dd) {       // (luma, d) of the modulated line at sample n (zero outside the row)
            y = dd = T(0);
            if (n < 0 || n >= W) return;
            T y0, dr0, db0;
            comps(i, n, y0, dr0, db0);
            if (d.averaging) {
                T y1 = y0, dr1 = dr0, db1 = db0;
                if (have_prev) comps(i - 1, n, y1, dr1, db1);
                y = y1;                                                  // comb.py:147: the previous call's luma
                dr0 = T(0.5) * (dr0 + dr1);
                db0 = T(0.5) * (db0 + db1);
            } else {
                y = y0;
            }
            dd = alt ? db0 : dr0;                                        // protosecam.py:75-78
        };
        ProtoMod<T> st;
        st.reset();
        for (int t = 0; t < W + lat; ++t) {
            const int i_c = t - (lat - lat_c), i_y = t - (lat - lat_y);
            T yc, dc, yy, dy;
            source(i_c, yc, dc);
            source(i_y, yy, dy);
            T lo, co;
            st.step(k, i_c, dc, i_y, yy, lo, co);
            const int n = t - lat;
            if (n >= 0 && n < W) {
                const double ph = phi + (double)n * d.carrier_phase_step;
                comp[(size_t)i * W + n] = (double)(lo + T(std::cos(ph)) * co);
            }
        }
    }
    return CM_OK;
}

extern "C" int am_sim_demod_run(const cm_am_desc *d, int use_float, const double *comp, double *rgb, int n_calls, long long frame,
                                int first_line, int k0) {
    if (d->kind == CM_AM_PROTO_SECAM)
        return use_float ? proto_demod_run<float>(*d, comp, rgb, n_calls, frame, first_line, k0)
                         : proto_demod_run<double>(*d, comp, rgb, n_calls, frame, first_line, k0);
    g_err = "kind not simulated";
    return CM_ERR_UNSUPPORTED;
}
extern "C" int am_sim_mod_run(const cm_am_desc *d, int use_float, const double *rgb, double *comp, int n_calls, long long frame,
                              int first_line, int k0) {
    if (d->kind == CM_AM_PROTO_SECAM)
        return use_float ? proto_mod_run<float>(*d, rgb, comp, n_calls, frame, first_line, k0)
                         : proto_mod_run<double>(*d, rgb, comp, n_calls, frame, first_line, k0);
    g_err = "kind not simulated";
    return CM_ERR_UNSUPPORTED;
}

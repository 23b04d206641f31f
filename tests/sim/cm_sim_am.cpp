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
            st.step(k, t, t < W ? T(comp[(size_t)i * W + t]) : T(0), (t >= 10 && t - 10 < W) ? T(comp[(size_t)i * W + t - 10]) : T(0), l, c);
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
            T yc, dc, yy, dy, yd, dd10;
            source(i_c, yc, dc);
            source(i_y, yy, dy);
            source(i_y - kAmHalf, yd, dd10);
            T lo, co;
            st.step(k, i_c, dc, i_y, yy, yd, lo, co);
            const int n = t - lat;
            if (n >= 0 && n < W) {
                const double ph = phi + (double)n * d.carrier_phase_step;
                comp[(size_t)i * W + n] = (double)(lo + T(std::cos(ph)) * co);
            }
        }
    }
    return CM_OK;
}


// ---- NIIR ---------------------------------------------------------------------------------------------------------------
template <typename T>
static int niir_demod_run(const cm_am_desc &d, const double *comp, double *rgb, int n_calls, long long frame, int first_line, int k0,
                          bool strip) {
    NiirDemodK<T> k;
    if (!build_niir_demod_k<T>(d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const AmLine ln = am_line(d);
    const int W = d.width;
    const int lat = 2 * kAmHalf + 1 + k.gb.q + k.gl.q;
    const int steps = W + lat;
    struct Tri { T v[3]; };
    std::vector<std::vector<Tri>> P(n_calls, std::vector<Tri>(steps)), S(n_calls, std::vector<Tri>(steps));
    auto run_front = [&](int i) {
        NiirFront<T> f;
        f.reset();
        std::vector<Tri> mh(steps);
        for (int t = 0; t < steps; ++t) {
            T m[3], sv[3];
            f.step(k, t, t < W ? T(comp[(size_t)i * W + t]) : T(0), (t >= 10 && t - 10 < W) ? T(comp[(size_t)i * W + t - 10]) : T(0), m, sv);
            for (int j = 0; j < 3; ++j) mh[t].v[j] = m[j];
            T md[3] = {T(0), T(0), T(0)};
            if (t - k.gl.q >= 0) for (int j = 0; j < 3; ++j) md[j] = mh[t - k.gl.q].v[j];
            const int n3 = t - kAmHalf - k.gb.q - k.gl.q;
            T p[3];
            niir_phasemod(k, n3, md, sv, p);
            for (int j = 0; j < 3; ++j) { P[i][t].v[j] = p[j]; S[i][t].v[j] = sv[j]; }
        }
    };
    for (int i = 0; i < n_calls; ++i) run_front(i);
    for (int i = 0; i < n_calls; ++i) {
        const int line = first_line + 2 * i;
        const NiirLineK<T> lk = niir_line_k<T>(d, ln, frame, line);
        std::vector<Tri> prev(steps);
        if (k0 + i == 0) {          // niir.py:107-110: the reference carrier of line - 2, band-passed, not normalised
            const double phi = ln.start_phase(frame, line - 2);
            const bool palt = ln.alternate(frame, line - 2);
            NiirSyn<T> sy;
            sy.reset();
            std::vector<Tri> mh(steps);
            for (int t = 0; t < steps; ++t) {
                T x = T(0), xd = T(0);
                if (t < W) x = T((palt ? -1.0 : 1.0) * std::sin(phi + (double)t * d.carrier_phase_step));
                if (t >= 10 && t - 10 < W) xd = T((palt ? -1.0 : 1.0) * std::sin(phi + (double)(t - 10) * d.carrier_phase_step));
                T m[3];
                sy.step(k, t, x, xd, m);
                for (int j = 0; j < 3; ++j) mh[t].v[j] = m[j];
                for (int j = 0; j < 3; ++j) prev[t].v[j] = t - k.gl.q >= 0 ? k.g_b * mh[t - k.gl.q].v[j] : T(0);
            }
        } else if (i > 0) {
            prev = P[i - 1];
        } else {
            for (auto &x : prev) x.v[0] = x.v[1] = x.v[2] = T(0);
        }
        NiirBack<T> b;
        b.reset();
        for (int t = 0; t < steps; ++t) {
            const int n3 = t - kAmHalf - k.gb.q - k.gl.q;
            const NiirOut<T> o = b.step(k, n3, P[i][t].v, prev[t].v, S[i][t].v, lk.alt);
            const int n = t - lat;
            if (n >= 0 && n < W) {
                const Rgb<T> c = niir_finish(k, lk, o, T(comp[(size_t)i * W + n]), strip);
                rgb[((size_t)i * 3 + 0) * W + n] = (double)c.r;
                rgb[((size_t)i * 3 + 1) * W + n] = (double)c.g;
                rgb[((size_t)i * 3 + 2) * W + n] = (double)c.b;
            }
        }
    }
    return CM_OK;
}

// The precision split of the device decoder (round 4; cm_am_stages.h: NiirHue): front end, phasor quotient, hue products and their two
// decimators in TP, the saturation / re-modulation decimators and niir_finish in T.  <double, float> is what the kernels compute.
template <typename TP, typename T>
static int niir_demod_run_split(const cm_am_desc &d, const double *comp, double *rgb, int n_calls, long long frame, int first_line, int k0,
                                bool strip) {
    NiirDemodK<TP> kd;
    NiirDemodK<T> k;
    if (!build_niir_demod_k<TP>(d, kd, g_err) || !build_niir_demod_k<T>(d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const AmLine ln = am_line(d);
    const int W = d.width;
    const int lat = 2 * kAmHalf + 1 + k.gb.q + k.gl.q;
    const int steps = W + lat;
    struct Tri { TP v[3]; };
    std::vector<std::vector<Tri>> P(n_calls, std::vector<Tri>(steps)), S(n_calls, std::vector<Tri>(steps));
    auto x_at = [&](int i, int t) { return (t >= 0 && t < W) ? TP(comp[(size_t)i * W + t]) : TP(0); };
    for (int i = 0; i < n_calls; ++i) {
        NiirFront<TP> f;
        f.reset();
        std::vector<Tri> mh(steps);
        for (int t = 0; t < steps; ++t) {
            TP m[3], sv[3];
            f.step(kd, t, x_at(i, t), x_at(i, t - kAmHalf), m, sv);
            for (int j = 0; j < 3; ++j) mh[t].v[j] = m[j];
            TP md[3] = {TP(0), TP(0), TP(0)};
            if (t - kd.gl.q >= 0) for (int j = 0; j < 3; ++j) md[j] = mh[t - kd.gl.q].v[j];
            TP p[3];
            niir_phasemod(kd, t - kAmHalf - kd.gb.q - kd.gl.q, md, sv, p);
            for (int j = 0; j < 3; ++j) { P[i][t].v[j] = p[j]; S[i][t].v[j] = sv[j]; }
        }
    }
    for (int i = 0; i < n_calls; ++i) {
        const int line = first_line + 2 * i;
        const NiirLineK<T> lk = niir_line_k<T>(d, ln, frame, line);
        std::vector<Tri> prev(steps);
        if (k0 + i == 0) {          // niir.py:107-110
            const double phi = ln.start_phase(frame, line - 2);
            const double sg = ln.alternate(frame, line - 2) ? -1.0 : 1.0;
            auto xs = [&](int t) { return (t >= 0 && t < W) ? TP(sg * std::sin(phi + (double)t * d.carrier_phase_step)) : TP(0); };
            NiirSyn<TP> sy;
            sy.reset();
            std::vector<Tri> mh(steps);
            for (int t = 0; t < steps; ++t) {
                TP m[3];
                sy.step(kd, t, xs(t), xs(t - kAmHalf), m);
                for (int j = 0; j < 3; ++j) mh[t].v[j] = m[j];
                for (int j = 0; j < 3; ++j) prev[t].v[j] = t - kd.gl.q >= 0 ? kd.g_b * mh[t - kd.gl.q].v[j] : TP(0);
            }
        } else if (i > 0) {
            prev = P[i - 1];
        } else {
            for (auto &x : prev) x.v[0] = x.v[1] = x.v[2] = TP(0);
        }
        NiirHue<TP> hue;
        Dn3<T> dn_sat, dn_sc, dn_cc;
        hue.reset(); dn_sat.reset(); dn_sc.reset(); dn_cc.reset();
        T s1[3] = {T(0), T(0), T(0)};
        for (int t = 0; t < steps; ++t) {
            const int n3 = t - kAmHalf - kd.gb.q - kd.gl.q;
            TP sp, cp, car[3], acar[3];
            hue.step(kd.taps, kd.alt_scale, W, n3, P[i][t].v, prev[t].v, lk.alt, sp, cp, car, acar);
            T cf[3], af[3];
            for (int j = 0; j < 3; ++j) { cf[j] = T(car[j]); af[j] = T(acar[j]); }
            NiirOut<T> o;
            o.sinphi = T(sp);
            o.cosphi = T(cp);
            o.sat = k.sat_gain * dn_sat.push(k.taps, s1);
            o.sinc = k.third * dn_sc.push(k.taps, cf);
            o.cosc = k.third * dn_cc.push(k.taps, af);
            for (int j = 0; j < 3; ++j) s1[j] = T(S[i][t].v[j]);
            const int n = t - lat;
            if (n >= 0 && n < W) {
                const Rgb<T> c = niir_finish(k, lk, o, T(comp[(size_t)i * W + n]), strip);
                rgb[((size_t)i * 3 + 0) * W + n] = (double)c.r;
                rgb[((size_t)i * 3 + 1) * W + n] = (double)c.g;
                rgb[((size_t)i * 3 + 2) * W + n] = (double)c.b;
            }
        }
    }
    return CM_OK;
}

template <typename T>
static int niir_mod_run(const cm_am_desc &d, const double *rgb, double *comp, int n_calls, long long frame, int first_line, int k0) {
    NiirModK<T> k;
    if (!build_niir_mod_k<T>(d, k, g_err)) return CM_ERR_UNSUPPORTED;
    const AmLine ln = am_line(d);
    const int W = d.width;
    for (int i = 0; i < n_calls; ++i) {
        const int call_line = first_line + 2 * i;
        const int line = d.averaging ? call_line - 2 : call_line;        // niir.py:202
        const bool alt = ln.alternate(frame, line);
        const double phi = ln.start_phase(frame, line);
        const bool have_prev = (k0 + i) > 0 && i > 0;
        auto comps = [&](int row, int n, T &y, T &db, T &dr) {
            const double *r = rgb + ((size_t)row * 3) * W;
            const T R = T(r[n]), G = T(r[W + n]), B = T(r[2 * W + n]);
            y = k.e[0][0] * R + k.e[0][1] * G + k.e[0][2] * B;
            db = k.e[1][0] * R + k.e[1][1] * G + k.e[1][2] * B;
            dr = k.e[2][0] * R + k.e[2][1] * G + k.e[2][2] * B;
        };
        NiirMod<T> st;
        st.reset();
        std::vector<T> lumas(W + k.s_c + 1, T(0));
        for (int t = 0; t < W + k.s_c; ++t) {
            T y = T(0), db = T(0), dr = T(0);
            if (t < W) {
                comps(i, t, y, db, dr);
                const double *px = rgb + ((size_t)i * 3) * W, *pp = rgb + ((size_t)(have_prev ? i - 1 : i) * 3) * W;
                if (d.averaging) {
                    T py = y, pdb = db, pdr = dr;
                    if (have_prev) comps(i - 1, t, py, pdb, pdr);
                    T odb, odr;
                    niir_hue_pixel(k.ed, (float)px[t], (float)px[W + t], (float)px[2 * W + t], (float)pp[t], (float)pp[W + t], (float)pp[2 * W + t], db, dr,
                                   pdb, pdr, T(0), T(0), odb, odr);
                    y = py;
                    db = odb;
                    dr = odr;
                } else {
                    niir_offset_pixel(k.ed, (float)px[t], (float)px[W + t], (float)px[2 * W + t], db, dr, T(0), T(0), false);
                }
                lumas[t] = y;
            }
            const int n = t - k.s_c;
            const double ph = phi + (double)n * d.carrier_phase_step;
            const T c = st.step(k, t, db, dr, alt, T(std::sin(ph)), T(std::cos(ph)));
            if (n >= 0 && n < W) comp[(size_t)i * W + n] = (double)(lumas[n] + c);
        }
    }
    return CM_OK;
}

extern "C" int am_sim_demod_run(const cm_am_desc *d, int use_float, const double *comp, double *rgb, int n_calls, long long frame,
                                int first_line, int k0) {
    if (d->kind == CM_AM_PROTO_SECAM)
        return use_float ? proto_demod_run<float>(*d, comp, rgb, n_calls, frame, first_line, k0)
                         : proto_demod_run<double>(*d, comp, rgb, n_calls, frame, first_line, k0);
    if (d->kind == CM_AM_NIIR) {    // use_float bit 1: strip_chroma = False; bit 2: the device's precision split (hue path float64, the rest float32)
        if (use_float & 4) return niir_demod_run_split<double, float>(*d, comp, rgb, n_calls, frame, first_line, k0, !(use_float & 2));
        return (use_float & 1) ? niir_demod_run<float>(*d, comp, rgb, n_calls, frame, first_line, k0, !(use_float & 2))
                               : niir_demod_run<double>(*d, comp, rgb, n_calls, frame, first_line, k0, !(use_float & 2));
    }
    g_err = "kind not simulated";
    return CM_ERR_UNSUPPORTED;
}
extern "C" int am_sim_mod_run(const cm_am_desc *d, int use_float, const double *rgb, double *comp, int n_calls, long long frame,
                              int first_line, int k0) {
    if (d->kind == CM_AM_PROTO_SECAM)
        return use_float ? proto_mod_run<float>(*d, rgb, comp, n_calls, frame, first_line, k0)
                         : proto_mod_run<double>(*d, rgb, comp, n_calls, frame, first_line, k0);
    if (d->kind == CM_AM_NIIR)
        return use_float ? niir_mod_run<float>(*d, rgb, comp, n_calls, frame, first_line, k0)
                         : niir_mod_run<double>(*d, rgb, comp, n_calls, frame, first_line, k0);
    g_err = "kind not simulated";
    return CM_ERR_UNSUPPORTED;
}

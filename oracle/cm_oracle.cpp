/*
 * cm_oracle.cpp - CPU oracle for the color_modem hot path.  TEST INFRASTRUCTURE ONLY.
 * See cm_oracle.h for scope and parity status (pinned against reference-generated goldens).
 *
 * The class layout mirrors the reference so that each method can name the lines it restates
 * ("ref:" comments, paths relative to the reference repository root).  float64 throughout,
 * like the reference.
 */
#include "cm_oracle.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef std::vector<double> vec;
const double PI = 3.141592653589793238462643383279502884;
const double TWO_PI = 2.0 * PI;

thread_local std::string g_error;

/* Python's float % for a positive divisor: result in [0, m). */
inline double pymod(double x, double m) {
    double r = std::fmod(x, m);
    if (r != 0.0 && r < 0.0) r += m;
    return r;
}
/* Python's int // and % (floor semantics). */
inline int floordiv(int a, int b) {
    int q = a / b, r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) --q;
    return q;
}
inline int floormod(int a, int b) {
    int r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) r += b;
    return r;
}

/* ---- scipy.signal.firwin(41, 0.5, window=('kaiser', 5.0)) --------------------------------
 * ref: scipy.signal.resample_poly designs this filter on every call (qam.py:35,37,45,53,54,57;
 * pal.py:72,77; secam.py:136,149 all use up/down = 2 -> half_len 20, 41 taps, cutoff 0.5). */
double bessel_i0(double x) {
    double q = 0.25 * x * x, term = 1.0, sum = 1.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / (double(k) * double(k));
        sum += term;
        if (term < 1e-19 * sum) break;
    }
    return sum;
}

struct ResampleFir {
    double h[41];
    ResampleFir() {
        const int M = 41;
        const double alpha = 0.5 * (M - 1), beta = 5.0;
        double s = 0.0;
        for (int n = 0; n < M; ++n) {
            double m = n - alpha;
            double arg = 0.5 * m;
            double sinc = (arg == 0.0) ? 1.0 : std::sin(PI * arg) / (PI * arg);
            double r = (n - alpha) / alpha;
            double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
            h[n] = 0.5 * sinc * w;
            s += h[n];
        }
        for (int n = 0; n < M; ++n) h[n] /= s;
    }
};
const ResampleFir &fir() {
    static ResampleFir f;
    return f;
}

/* resample_poly(x, up=2, down=1): y[m] = sum_k 2 h[k] xu[m + 20 - k], xu = zero-stuffed x,
 * zero-extended; length 2n (checked against scipy in tests). */
vec up2(const vec &x) {
    const double *h = fir().h;
    const int n = (int)x.size();
    vec y(2 * (size_t)n, 0.0);
    for (int m = 0; m < 2 * n; ++m) {
        double acc = 0.0;
        for (int k = 0; k < 41; ++k) {
            int j = m + 20 - k; /* index into the zero-stuffed signal */
            if (j < 0 || j >= 2 * n || (j & 1)) continue;
            acc += 2.0 * h[k] * x[j >> 1];
        }
        y[m] = acc;
    }
    return y;
}
/* resample_poly(z, up=1, down=2): y[i] = sum_k h[k] z[2i + 20 - k], zero-extended; len ceil(n/2). */
vec dn2(const vec &z) {
    const double *h = fir().h;
    const int n = (int)z.size();
    const int nout = n / 2 + (n % 2);
    vec y((size_t)nout, 0.0);
    for (int i = 0; i < nout; ++i) {
        double acc = 0.0;
        for (int k = 0; k < 41; ++k) {
            int j = 2 * i + 20 - k;
            if (j < 0 || j >= n) continue;
            acc += h[k] * z[j];
        }
        y[i] = acc;
    }
    return y;
}

/* scipy.signal.lfilter(b, a, x): transposed direct form II, zero initial state. */
vec lfilter(const double *b_in, int nb, const double *a_in, int na, const vec &x) {
    int n = std::max(nb, na);
    vec b(n, 0.0), a(n, 0.0), z(n, 0.0);
    for (int i = 0; i < nb; ++i) b[i] = b_in[i] / a_in[0];
    for (int i = 0; i < na; ++i) a[i] = a_in[i] / a_in[0];
    vec y(x.size());
    for (size_t t = 0; t < x.size(); ++t) {
        double xi = x[t];
        double yi = b[0] * xi + z[0];
        for (int i = 1; i < n; ++i) z[i - 1] = b[i] * xi - a[i] * yi + (i + 1 < n ? z[i] : 0.0);
        y[t] = yi;
    }
    return y;
}

/* ref: utils.py:9-36 FilterFunction (construction is done by the caller). */
struct Filter {
    orc_filter_t f;
    bool present() const { return f.present != 0; }
    vec operator()(const vec &x) const {
        if (f.shift == 0) return lfilter(f.b, f.nb, f.a, f.na, x); /* utils.py:29-30 */
        if (f.shift > 0) {                                         /* utils.py:31-33 */
            vec ext(x);
            ext.insert(ext.end(), (size_t)f.shift, x.back());
            vec y = lfilter(f.b, f.nb, f.a, f.na, ext);
            return vec(y.begin() + f.shift, y.end());
        }
        vec ext((size_t)(-f.shift), x.front()); /* utils.py:34-36 */
        ext.insert(ext.end(), x.begin(), x.end());
        vec y = lfilter(f.b, f.nb, f.a, f.na, ext);
        y.resize(x.size());
        return y;
    }
};

/* numpy.linspace(start, start + n*step, n, endpoint=False): start + k * ((stop - start) / n) */
inline double ramp(double start, double step, int n, int k) {
    double stop = start + n * step;
    double delta = (stop - start) / n;
    return start + k * delta;
}

/* ---- line.py ------------------------------------------------------------------------------ */
struct LineConfig {
    orc_desc_t d;
    int line_shift; /* line.py:55 */
    double fs;      /* line.py:53 */
    explicit LineConfig(const orc_desc_t &desc) : d(desc) {
        int active = d.odd_last - d.odd_first + d.even_last - d.even_first + 2; /* line.py:24-26 */
        line_shift = floordiv(active - d.height, 2);
        fs = d.frame_rate * d.total_lines * d.width * d.total_width_factor;
    }
    int analog_line(int digital_line) const { /* line.py:57-62 */
        int adjusted = digital_line + line_shift;
        if (floormod(adjusted, 2) == 0) return d.even_first + floordiv(adjusted, 2);
        return d.odd_first + floordiv(adjusted, 2);
    }
    bool is_alternate_line(int frame, int line) const { /* line.py:64-65 */
        return floormod(analog_line(line), 2) == floormod(frame, 2);
    }
};

struct YUV {
    vec y, u, v;
};
struct RGB {
    vec r, g, b;
};

/* ---- qam.py:10-58 QamColorModem ----------------------------------------------------------- */
struct Qam {
    double cps;
    Filter pre, extract2x, remove2x, demod_lp;
    vec modulate_chroma(double start_phase, const vec &u_in, const vec &v_in) const { /* qam.py:20-26 */
        vec u = pre(u_in), v = pre(v_in);
        int n = (int)u.size();
        vec out(n);
        for (int k = 0; k < n; ++k) {
            double p = pymod(ramp(start_phase, 2.0 * cps, n, k), TWO_PI);
            out[k] = std::sin(p) * u[k] + std::cos(p) * v[k];
        }
        return out;
    }
    vec modulate(double start_phase, const vec &y, const vec &u, const vec &v) const { /* qam.py:28-32 */
        vec c = modulate_chroma(start_phase, u, v);
        for (size_t k = 0; k < c.size(); ++k) c[k] = y[k] + c[k];
        return c;
    }
    vec extract_chroma(const vec &composite) const { /* qam.py:34-37 */
        return dn2(extract2x(up2(composite)));
    }
    YUV demodulate(double start_phase, const vec &composite, bool strip_chroma) const { /* qam.py:43-58 */
        double shifted = start_phase + extract2x.f.phase_shift;
        vec c2 = up2(composite);
        vec ch = extract2x(c2);
        int n = (int)ch.size();
        vec u2(n), v2(n);
        for (int k = 0; k < n; ++k) {
            double p = pymod(ramp(shifted, cps, n, k), TWO_PI);
            u2[k] = 2.0 * std::sin(p) * ch[k];
            v2[k] = 2.0 * std::cos(p) * ch[k];
        }
        YUV out;
        out.u = dn2(demod_lp(u2));
        out.v = dn2(demod_lp(v2));
        out.y = strip_chroma ? dn2(remove2x(c2)) : composite;
        return out;
    }
};

/* ---- the duck-typed modem protocol -------------------------------------------------------- */
struct Modem {
    int modulation_delay = 0, demodulation_delay = 0;
    virtual ~Modem() {}
    virtual vec modulate(int frame, int line, const vec &r, const vec &g, const vec &b) = 0;
    virtual RGB demodulate(int frame, int line, const vec &composite) = 0;
    /* component-level members used by the wrappers (comb.py) */
    virtual bool has_components() const { return false; }
    virtual YUV demodulate_components(int, int, const vec &, bool) { return YUV(); }
    virtual vec modulate_components(int, int, const vec &, const vec &, const vec &) { return vec(); }
    virtual YUV encode_components(const vec &, const vec &, const vec &) const { return YUV(); }
    virtual RGB decode_components(const vec &, const vec &, const vec &) const { return RGB(); }
};

/* ---- utils.py:67-88 ConstantFrequencyCarrier + qam.py:61-72 AbstractQamColorModem --------- */
struct QamBackend : Modem {
    LineConfig lc;
    Qam qam;
    double fsc;
    int frame_cycle;
    explicit QamBackend(const orc_desc_t &d) : lc(d), fsc(d.fsc), frame_cycle(d.frame_cycle) {
        qam.cps = d.carrier_phase_step;
        qam.pre.f = d.filters[ORC_F_QAM_PRE];
        qam.extract2x.f = d.filters[ORC_F_QAM_EXTRACT2X];
        qam.remove2x.f = d.filters[ORC_F_QAM_REMOVE2X];
        qam.demod_lp.f = d.filters[ORC_F_QAM_DEMOD_LP];
    }
    bool has_components() const override { return true; }
    double line_shift() const { /* utils.py:68-71 */
        return TWO_PI * pymod(fsc / (lc.d.frame_rate * lc.d.total_lines), 1.0);
    }
    double frame_shift() const { /* utils.py:73-75 */
        return TWO_PI * pymod(fsc / lc.d.frame_rate, 1.0);
    }
    double start_phase(int frame, int line) const { /* utils.py:82-88 */
        int reference_line = std::min(lc.d.odd_first, lc.d.even_first);
        frame = floormod(frame, frame_cycle);
        double fshift = pymod(frame * frame_shift(), TWO_PI);
        double lshift = pymod((lc.analog_line(line) - reference_line) * line_shift(), TWO_PI);
        return pymod(fshift + lshift, TWO_PI);
    }
    vec modulate(int frame, int line, const vec &r, const vec &g, const vec &b) override { /* qam.py:68-69 */
        YUV c = encode_components(r, g, b);
        return modulate_components(frame, line, c.y, c.u, c.v);
    }
    RGB demodulate(int frame, int line, const vec &composite) override { /* qam.py:71-72 */
        YUV c = demodulate_components(frame, line, composite, true);
        return decode_components(c.y, c.u, c.v);
    }
};

/* ---- color/pal.py:28-59 PalSModem --------------------------------------------------------- */
struct PalS : QamBackend {
    explicit PalS(const orc_desc_t &d) : QamBackend(d) {}
    YUV encode_components(const vec &r, const vec &g, const vec &b) const override { /* pal.py:33-38 */
        size_t n = r.size();
        YUV o{vec(n), vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            o.y[k] = 0.299 * r[k] + 0.587 * g[k] + 0.114 * b[k];
            o.u[k] = -0.147407 * r[k] - 0.289391 * g[k] + 0.436798 * b[k];
            o.v[k] = 0.614777 * r[k] - 0.514799 * g[k] - 0.099978 * b[k];
        }
        return o;
    }
    RGB decode_components(const vec &y, const vec &u, const vec &v) const override { /* pal.py:41-46 */
        size_t n = y.size();
        RGB o{vec(n), vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            o.r[k] = y[k] + 1.140250855188141 * v[k];
            o.g[k] = y[k] - 0.5808092090310976 * v[k] - 0.3939307027516405 * u[k];
            o.b[k] = y[k] + 2.028397565922921 * u[k];
        }
        return o;
    }
    vec modulate_components(int frame, int line, const vec &y, const vec &u, const vec &v) override { /* pal.py:48-52 */
        double sp = start_phase(frame, line);
        if (lc.is_alternate_line(frame, line)) {
            vec nv(v);
            for (double &x : nv) x = -x;
            return qam.modulate(sp, y, u, nv);
        }
        return qam.modulate(sp, y, u, v);
    }
    YUV demodulate_components(int frame, int line, const vec &composite, bool strip) override { /* pal.py:54-59 */
        double sp = start_phase(frame, line);
        YUV c = qam.demodulate(sp, composite, strip);
        if (lc.is_alternate_line(frame, line))
            for (double &x : c.v) x = -x;
        return c;
    }
};

/* ---- color/ntsc.py:23-49 NtscModem -------------------------------------------------------- */
struct Ntsc : QamBackend {
    explicit Ntsc(const orc_desc_t &d) : QamBackend(d) {}
    YUV encode_components(const vec &r, const vec &g, const vec &b) const override { /* ntsc.py:28-33 */
        size_t n = r.size();
        YUV o{vec(n), vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            o.y[k] = 0.3 * r[k] + 0.59 * g[k] + 0.11 * b[k];
            o.u[k] = -0.1476019510016258 * r[k] - 0.2893575108184752 * g[k] + 0.436959461820101 * b[k];
            o.v[k] = 0.6183717846575098 * r[k] - 0.5185533057776567 * g[k] - 0.099818478879853 * b[k];
        }
        return o;
    }
    RGB decode_components(const vec &y, const vec &u, const vec &v) const override { /* ntsc.py:36-41 */
        size_t n = y.size();
        RGB o{vec(n), vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            o.r[k] = 0.9999999999999998 * y[k] + 1.133735501874552 * v[k] + 0.007249535771601484 * u[k];
            o.g[k] = y[k] - 0.5766784873222262 * v[k] - 0.3834753199055935 * u[k];
            o.b[k] = y[k] + 0.001087790524980047 * v[k] + 2.037050709207452 * u[k];
        }
        return o;
    }
    vec modulate_components(int frame, int line, const vec &y, const vec &u, const vec &v) override { /* ntsc.py:43-45 */
        return qam.modulate(start_phase(frame, line), y, u, v);
    }
    YUV demodulate_components(int frame, int line, const vec &composite, bool strip) override { /* ntsc.py:47-49 */
        return qam.demodulate(start_phase(frame, line), composite, strip);
    }
};

/* ---- comb.py:23-68 AbstractCombModem ------------------------------------------------------ */
struct AbstractComb : Modem {
    std::unique_ptr<QamBackend> backend;
    int last_frame = -1, last_line = -1;
    bool have_last = false;
    vec last_composite;
    Filter notch{}; /* comb.py:29-31 */
    explicit AbstractComb(QamBackend *b) : backend(b) {}
    bool has_components() const override { return true; }
    virtual YUV demodulate_components_combed(int frame, int line, const vec &last, const vec &curr) = 0;
    vec modulate_components(int frame, int line, const vec &y, const vec &u, const vec &v) override { /* comb.py:41-42 */
        return backend->modulate_components(frame, line, y, u, v);
    }
    vec modulate(int frame, int line, const vec &r, const vec &g, const vec &b) override { /* comb.py:44-45 */
        return backend->modulate(frame, line, r, g, b);
    }
    YUV demodulate_components(int frame, int line, const vec &composite, bool strip) override { /* comb.py:47-59 */
        YUV c;
        if (frame != last_frame || line != last_line + 2 || !have_last) {
            c = backend->demodulate_components(frame, line, composite, strip);
        } else {
            c = demodulate_components_combed(frame, line, last_composite, composite);
            if (strip) {
                vec zeros(composite.size(), 0.0);
                vec m = backend->modulate_components(frame, line, zeros, c.u, c.v);
                for (size_t k = 0; k < m.size(); ++k) c.y[k] = c.y[k] - m[k];
                if (notch.present()) c.y = notch(c.y); /* comb.py:54-55 */
            }
        }
        last_frame = frame;
        last_line = line;
        last_composite = composite;
        have_last = true;
        return c;
    }
    YUV encode_components(const vec &r, const vec &g, const vec &b) const override { return backend->encode_components(r, g, b); }
    RGB decode_components(const vec &y, const vec &u, const vec &v) const override { return backend->decode_components(y, u, v); }
    RGB demodulate(int frame, int line, const vec &composite) override { /* comb.py:67-68 */
        YUV c = demodulate_components(frame, line, composite, true);
        return backend->decode_components(c.y, c.u, c.v);
    }
};

/* ---- color/pal.py:62-127 PalDModem -------------------------------------------------------- */
struct PalD : AbstractComb {
    double sin_factor, cos_factor;
    Filter filter;
    explicit PalD(const orc_desc_t &d) : AbstractComb(new PalS(d)) {
        notch.f = d.filters[ORC_F_COMB_NOTCH];
        sin_factor = std::sin(0.5 * backend->line_shift()); /* pal.py:65 */
        cos_factor = std::cos(0.5 * backend->line_shift()); /* pal.py:66 */
        filter.f = d.filters[ORC_F_PALD_LP];                /* pal.py:67-69 */
    }
    vec demodulate_am(const vec &data, double start_phase) const { /* pal.py:71-77 */
        vec d2 = up2(data);
        int n = (int)d2.size();
        for (int k = 0; k < n; ++k) {
            double p = pymod(ramp(start_phase, backend->qam.cps, n, k), TWO_PI);
            d2[k] *= std::sin(p);
        }
        return dn2(filter(d2));
    }
    YUV demodulate_components_combed(int frame, int line, const vec &last, const vec &curr) override { /* pal.py:79-127 */
        double diff_phase = pymod(backend->start_phase(frame, line) + backend->qam.extract2x.f.phase_shift -
                                      0.5 * backend->line_shift(),
                                  TWO_PI);
        size_t n = curr.size();
        vec s(n), d(n);
        for (size_t k = 0; k < n; ++k) {
            s[k] = curr[k] + last[k];
            d[k] = curr[k] - last[k];
        }
        vec sumsig = backend->qam.extract_chroma(s);
        vec diff = backend->qam.extract_chroma(d);
        sumsig = demodulate_am(sumsig, diff_phase);
        diff = demodulate_am(diff, pymod(diff_phase + 0.5 * PI, TWO_PI));
        YUV c{curr, vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            c.u[k] = diff[k] * sin_factor + sumsig[k] * cos_factor;
            c.v[k] = diff[k] * cos_factor - sumsig[k] * sin_factor;
        }
        if (backend->lc.is_alternate_line(frame, line))
            for (double &x : c.v) x *= -1.0;
        return c;
    }
};

inline double avg_fn(double a, double b) { return 0.5 * (a + b); } /* comb.py:9-10 */
inline double minavg_fn(double a, double b) {                       /* comb.py:13-15 */
    double sign = (1.0 - (std::signbit(a) ? 1.0 : 0.0)) - (std::signbit(b) ? 1.0 : 0.0);
    return sign * std::min(std::fabs(a), std::fabs(b));
}

/* ---- color/pal.py:130-234 Pal3DModem (use_sin/use_cos defaults, avg or minavg) ------------- */
struct Pal3D : PalD {
    bool use_sin = true, use_cos = true, use_minavg;
    double sin_sum_factor = 0, cos_u_factor = 0, cos_v_factor = 0;
    bool have_last_diff = false, have_last_demodulated = false;
    vec last_diff;
    YUV last_demodulated;
    Pal3D(const orc_desc_t &d) : PalD(d), use_minavg((d.use_minavg & 1) != 0) {
        if (d.pal3d_disable & 1) use_sin = false;
        if (d.pal3d_disable & 2) use_cos = false;
        double lssin = std::sin(backend->line_shift()); /* pal.py:154-156 */
        if (std::fabs(lssin) < 0.1) use_sin = false;
        double lscos = std::cos(backend->line_shift()); /* pal.py:158-160 */
        if (std::fabs(lscos) > 0.9) use_cos = false;
        demodulation_delay = (use_cos || use_sin) ? 1 : 0; /* pal.py:162 */
        if (use_sin) sin_sum_factor = 0.5 / lssin;         /* pal.py:167-168 */
        if (use_cos) {                                     /* pal.py:170-172 */
            cos_u_factor = -0.5 / (1.0 - lscos);
            cos_v_factor = -0.5 / (1.0 + lscos);
        }
    }
    double av(double a, double b) const { return use_minavg ? minavg_fn(a, b) : avg_fn(a, b); }
    YUV demodulate_components(int frame, int line, const vec &composite, bool strip) override { /* pal.py:180-234 */
        if (!(use_sin || use_cos)) return PalD::demodulate_components(frame, line, composite, strip);
        if (frame != last_frame || line != last_line + 2) { /* pal.py:191-195 */
            have_last_diff = false;
            last_demodulated = AbstractComb::demodulate_components(frame, line, composite, false);
            have_last_demodulated = true;
            return last_demodulated;
        }
        size_t n = composite.size();
        vec curr_diff(n);
        for (size_t k = 0; k < n; ++k) curr_diff[k] = composite[k] - last_composite[k]; /* pal.py:198 */
        YUV c;
        if (!have_last_diff) { /* pal.py:199-201 */
            c = last_demodulated;
            have_last_demodulated = false;
        } else { /* pal.py:202-223 */
            vec ss(n), ds(n);
            for (size_t k = 0; k < n; ++k) {
                ss[k] = curr_diff[k] + last_diff[k];
                ds[k] = curr_diff[k] - last_diff[k];
            }
            double sp = backend->start_phase(frame, line - 2);
            YUV sumsig = backend->qam.demodulate(sp, ss, false);
            YUV diffsig = backend->qam.demodulate(sp, ds, false);
            c.u.resize(n);
            c.v.resize(n);
            for (size_t k = 0; k < n; ++k) {
                if (use_sin && use_cos) {
                    c.u[k] = av(sumsig.v[k] * sin_sum_factor, diffsig.u[k] * cos_u_factor);
                    c.v[k] = av(sumsig.u[k] * sin_sum_factor, diffsig.v[k] * cos_v_factor);
                } else if (use_sin) {
                    c.u[k] = sin_sum_factor * sumsig.v[k];
                    c.v[k] = sin_sum_factor * sumsig.u[k];
                } else {
                    c.u[k] = cos_u_factor * diffsig.u[k];
                    c.v[k] = cos_v_factor * diffsig.v[k];
                }
            }
            if (backend->lc.is_alternate_line(frame, line - 2))
                for (double &x : c.v) x *= -1.0;
            c.y = last_composite;
        }
        if (strip) { /* pal.py:225-228 */
            vec zeros(n, 0.0);
            vec m = backend->modulate_components(frame, line - 2, zeros, c.u, c.v);
            for (size_t k = 0; k < n; ++k) c.y[k] = c.y[k] - m[k];
            if (notch.present()) c.y = notch(c.y); /* pal.py:227-228 */
        }
        last_frame = frame;
        last_line = line;
        last_composite = composite;
        last_diff = curr_diff;
        have_last_diff = true;
        return c;
    }
};

/* ---- color/ntsc.py:52-82 NtscCombModem ---------------------------------------------------- */
struct NtscComb : AbstractComb {
    double factor;
    bool finite;
    explicit NtscComb(const orc_desc_t &d) : AbstractComb(new Ntsc(d)) {
        notch.f = d.filters[ORC_F_COMB_NOTCH];
        double sine = std::sin(backend->line_shift() * 0.5); /* ntsc.py:55-59 */
        finite = std::fabs(sine) > 0.05;
        factor = finite ? 0.5 / sine : INFINITY;
    }
    YUV demodulate_components_combed(int frame, int line, const vec &last, const vec &curr) override { /* ntsc.py:61-82 */
        if (!finite) return backend->demodulate_components(frame, line, curr, false);
        double diff_phase = backend->start_phase(frame, line) - 0.5 * backend->line_shift();
        if (diff_phase < 0.0) diff_phase += TWO_PI;
        size_t n = curr.size();
        vec diff(n);
        for (size_t k = 0; k < n; ++k) diff[k] = curr[k] - last[k];
        YUV q = backend->qam.demodulate(diff_phase, diff, false);
        YUV c{curr, q.v, q.u}; /* ntsc.py:79: "_, v, u = ..." */
        for (double &x : c.u) x *= factor;
        for (double &x : c.v) x *= -factor;
        return c;
    }
};

/* ---- comb.py:71-127 SimpleCombModem / Simple3DCombModem ----------------------------------- */
struct SimpleComb : Modem {
    std::unique_ptr<Modem> backend;
    int own_delay;
    bool use_minavg;
    int last_frame = -1, last_line = -1;
    YUV last_demodulated;
    Filter notch{}; /* comb.py:86-88 */
    SimpleComb(Modem *b, bool delay, bool minavg, const orc_filter_t &nf) : backend(b), own_delay(delay ? 1 : 0), use_minavg(minavg) {
        notch.f = nf;
        modulation_delay = backend->modulation_delay;                 /* comb.py:75 */
        demodulation_delay = backend->demodulation_delay + own_delay; /* comb.py:76 */
    }
    bool has_components() const override { return true; }
    double av(double a, double b) const { return use_minavg ? minavg_fn(a, b) : avg_fn(a, b); }
    vec modulate_components(int frame, int line, const vec &y, const vec &u, const vec &v) override {
        return backend->modulate_components(frame, line, y, u, v);
    }
    vec modulate(int frame, int line, const vec &r, const vec &g, const vec &b) override {
        return backend->modulate(frame, line, r, g, b);
    }
    YUV demodulate_components(int frame, int line, const vec &composite, bool strip) override { /* comb.py:96-113 */
        YUV curr = backend->demodulate_components(frame, line, composite, false);
        YUV c;
        if (frame != last_frame || line != last_line + 2) {
            c = curr;
        } else {
            size_t n = composite.size();
            c.y = own_delay ? last_demodulated.y : curr.y;
            c.u.resize(n);
            c.v.resize(n);
            for (size_t k = 0; k < n; ++k) {
                c.u[k] = av(last_demodulated.u[k], curr.u[k]);
                c.v[k] = av(last_demodulated.v[k], curr.v[k]);
            }
            if (strip) {
                vec zeros(n, 0.0);
                vec m = backend->modulate_components(frame, line - 2 * (own_delay - modulation_delay), zeros, c.u, c.v);
                for (size_t k = 0; k < n; ++k) c.y[k] = c.y[k] - m[k];
                if (notch.present()) c.y = notch(c.y); /* comb.py:109-110 */
            }
        }
        last_frame = frame;
        last_line = line;
        last_demodulated = curr;
        return c;
    }
    YUV encode_components(const vec &r, const vec &g, const vec &b) const override { return backend->encode_components(r, g, b); }
    RGB decode_components(const vec &y, const vec &u, const vec &v) const override { return backend->decode_components(y, u, v); }
    RGB demodulate(int frame, int line, const vec &composite) override { /* comb.py:121-122 */
        YUV c = demodulate_components(frame, line, composite, true);
        return backend->decode_components(c.y, c.u, c.v);
    }
};

/* ---- comb.py:130-167 ColorAveragingModem -------------------------------------------------- */
struct ColorAveraging : Modem {
    std::unique_ptr<Modem> backend;
    int last_frame = -1, last_line = -1;
    bool have_last = false;
    vec last_y, last_u, last_v;
    explicit ColorAveraging(Modem *b) : backend(b) {
        modulation_delay = backend->modulation_delay + 1; /* comb.py:133 */
        demodulation_delay = backend->demodulation_delay; /* comb.py:134 */
    }
    bool has_components() const override { return backend->has_components(); }
    vec modulate_components(int frame, int line, const vec &y_in, const vec &u_in, const vec &v_in) override { /* comb.py:141-152 */
        if (frame != last_frame || line != last_line + 2 || !have_last) {
            last_y = y_in;
            last_u = u_in;
            last_v = v_in;
            have_last = true;
        }
        vec y = last_y;
        last_y = y_in;
        size_t n = y_in.size();
        vec u(n), v(n);
        for (size_t k = 0; k < n; ++k) {
            u[k] = 0.5 * (u_in[k] + last_u[k]);
            v[k] = 0.5 * (v_in[k] + last_v[k]);
        }
        last_u = u_in;
        last_v = v_in;
        last_frame = frame;
        last_line = line;
        return backend->modulate_components(frame, line - 2, y, u, v);
    }
    vec modulate(int frame, int line, const vec &r, const vec &g, const vec &b) override { /* comb.py:154-155 */
        YUV c = backend->encode_components(r, g, b);
        return modulate_components(frame, line, c.y, c.u, c.v);
    }
    YUV demodulate_components(int frame, int line, const vec &composite, bool strip) override {
        return backend->demodulate_components(frame, line, composite, strip);
    }
    RGB demodulate(int frame, int line, const vec &composite) override { return backend->demodulate(frame, line, composite); }
    YUV encode_components(const vec &r, const vec &g, const vec &b) const override { return backend->encode_components(r, g, b); }
    RGB decode_components(const vec &y, const vec &u, const vec &v) const override { return backend->decode_components(y, u, v); }
};

/* ---- color/secam.py:127-149 FmDecoder ----------------------------------------------------- */
struct FmDecoder {
    double fc;
    Filter lowpass;
    vec operator()(const vec &data) const {
        vec up = up2(data); /* secam.py:136 */
        int n = (int)up.size();
        vec co(n), si(n);
        for (int k = 0; k < n; ++k) { /* secam.py:137-140, phase not wrapped */
            double stop = (n * PI * fc) / 2.0;
            double p = 0.0 + k * ((stop - 0.0) / n);
            co[k] = up[k] * std::cos(p);
            si[k] = up[k] * std::sin(p);
        }
        co = lowpass(co);
        si = lowpass(si);
        vec ph(n);
        for (int k = 0; k < n; ++k) ph[k] = std::arg(std::complex<double>(co[k], -si[k])); /* secam.py:143-145 */
        /* numpy.unwrap (default discont = pi, period = 2 pi), secam.py:146 */
        vec unwrapped(n);
        if (n > 0) unwrapped[0] = ph[0];
        double cum = 0.0;
        for (int k = 1; k < n; ++k) {
            double dd = ph[k] - ph[k - 1];
            double ddmod = pymod(dd + PI, TWO_PI) - PI;
            if (ddmod == -PI && dd > 0) ddmod = PI;
            double corr = ddmod - dd;
            if (std::fabs(dd) < PI) corr = 0.0;
            cum += corr;
            unwrapped[k] = ph[k] + cum;
        }
        vec freq(n);
        for (int k = 0; k < n; ++k) { /* secam.py:147-148 */
            double shift = (k == 0) ? 0.0 : unwrapped[k] - unwrapped[k - 1];
            freq[k] = fc + 2.0 * shift / PI;
        }
        return dn2(freq); /* secam.py:149 */
    }
};

/* ---- color/secam.py:152-304 SecamModem ---------------------------------------------------- */
struct Secam : Modem {
    LineConfig lc;
    double fsc_dr, fsc_db, fdev_dr, fdev_db, flimit_min, flimit_max, bell_f0, m0, kn, kd;
    bool inversions[6];
    Filter pre_lp, lf_pre, lf_rev, bell, chroma_bp, luma_bs;
    FmDecoder fm;
    int last_frame = -1, last_line = -1;
    bool have_last = false;
    vec last_chroma;
    explicit Secam(const orc_desc_t &d) : lc(d) {
        fsc_dr = d.fsc_dr; fsc_db = d.fsc_db; fdev_dr = d.fdev_dr; fdev_db = d.fdev_db;
        flimit_min = d.flimit_min; flimit_max = d.flimit_max; bell_f0 = d.bell_f0;
        m0 = d.m0; kn = d.bell_kn; kd = d.bell_kd;
        const bool a[6] = {false, false, true, false, false, true};  /* secam.py:164-165 */
        const bool b[6] = {false, false, false, true, true, true};   /* secam.py:166-167 */
        for (int i = 0; i < 6; ++i) inversions[i] = d.alternate_phases ? b[i] : a[i];
        pre_lp.f = d.filters[ORC_F_SECAM_PRE_LP];
        lf_pre.f = d.filters[ORC_F_SECAM_LF_PRE];
        lf_rev.f = d.filters[ORC_F_SECAM_LF_REV];
        bell.f = d.filters[ORC_F_SECAM_BELL];
        chroma_bp.f = d.filters[ORC_F_SECAM_CHROMA_BP];
        luma_bs.f = d.filters[ORC_F_SECAM_LUMA_BS];
        fm.fc = d.fm_fc;
        fm.lowpass.f = d.filters[ORC_F_SECAM_FM_LP];
    }
    static YUV encode(const vec &r, const vec &g, const vec &b) { /* secam.py:193-200 */
        size_t n = r.size();
        YUV o{vec(n), vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            o.y[k] = 0.299 * r[k] + 0.587 * g[k] + 0.114 * b[k];
            o.u[k] = -1.333302 * r[k] + 1.116474 * g[k] + 0.216828 * b[k]; /* dr */
            o.v[k] = -0.449995 * r[k] - 0.883435 * g[k] + 1.33343 * b[k];  /* db */
        }
        return o;
    }
    static RGB decode(const vec &luma, const vec &dr, const vec &db) { /* secam.py:203-208 */
        size_t n = luma.size();
        RGB o{vec(n), vec(n), vec(n)};
        for (size_t k = 0; k < n; ++k) {
            o.r[k] = luma[k] - 0.5257623554153522 * dr[k];
            o.g[k] = luma[k] + 0.2678074007993021 * dr[k] - 0.1290417517983779 * db[k];
            o.b[k] = luma[k] + 0.6644518272425249 * db[k];
        }
        return o;
    }
    YUV encode_components(const vec &r, const vec &g, const vec &b) const override { return encode(r, g, b); }
    RGB decode_components(const vec &y, const vec &u, const vec &v) const override { return decode(y, u, v); }
    vec modulate_chroma(double start_phase, const vec &f) const { /* secam.py:240-246 */
        typedef std::complex<double> cd;
        size_t n = f.size();
        std::vector<cd> G(n);
        for (size_t k = 0; k < n; ++k) {
            double F = f[k] / bell_f0 - bell_f0 / f[k];
            G[k] = m0 * (cd(1.0, 0.0) + cd(0.0, 1.0) * kn * F) / (cd(1.0, 0.0) + cd(0.0, 1.0) * kd * F);
        }
        vec out(n);
        double cum = 0.0;
        double head = start_phase - PI * f[0] - std::arg(G[0]);
        for (size_t k = 0; k < n; ++k) {
            cum += PI * f[k]; /* numpy.cumsum: sequential */
            double phase = pymod(head + cum, TWO_PI);
            out[k] = G[k].real() * std::cos(phase) - G[k].imag() * std::sin(phase);
        }
        return out;
    }
    bool start_phase_inverted(int frame, int line) const { /* secam.py:248-256 */
        frame = floormod(frame, 6);
        int line_in_field = (floormod(line, 2) == 0) ? 23 + floordiv(line, 2) : 336 + floordiv(line, 2);
        int line_in_sequence = floormod(frame * 625 + line_in_field, 6);
        return inversions[line_in_sequence] ^ (frame % 2 == 1);
    }
    vec modulate_components(int frame, int line, const vec &luma, const vec &dr, const vec &db) override { /* secam.py:261-276 */
        bool alt = lc.is_alternate_line(frame, line);
        vec c = pre_lp(alt ? db : dr);
        if (lf_pre.present()) c = lf_pre(c);
        size_t n = c.size();
        vec f(n);
        for (size_t k = 0; k < n; ++k) {
            double v = alt ? fsc_db + fdev_db * c[k] : fsc_dr + fdev_dr * c[k];
            f[k] = std::min(std::max(v, flimit_min), flimit_max);
        }
        double sp = start_phase_inverted(frame, line) ? PI : 0.0;
        vec chroma = modulate_chroma(sp, f);
        for (size_t k = 0; k < n; ++k) chroma[k] = luma[k] + chroma[k];
        return chroma;
    }
    bool has_components() const override { return false; } /* no demodulate_components in the reference */
    vec modulate(int frame, int line, const vec &r, const vec &g, const vec &b) override { /* secam.py:258-259 */
        YUV c = encode(r, g, b);
        return modulate_components(frame, line, c.y, c.u, c.v);
    }
    RGB demodulate(int frame, int line, const vec &composite) override { /* secam.py:278-304 */
        size_t n = composite.size();
        if (frame != last_frame || line != last_line + 2 || !have_last) {
            last_chroma.assign(n, 0.0);
            have_last = true;
        }
        vec luma = luma_bs(composite);
        int pre = (int)n / 40; /* composite[1:len//40] flipped */
        vec comp;
        for (int i = pre - 1; i >= 1; --i) comp.push_back(composite[i]);
        comp.insert(comp.end(), composite.begin(), composite.end());
        vec chroma = chroma_bp(comp);
        if (bell.present()) chroma = bell(chroma);
        vec freq_all = fm(chroma);
        vec freq(freq_all.end() - n, freq_all.end());
        bool alt = lc.is_alternate_line(frame, line);
        vec c(n);
        for (size_t k = 0; k < n; ++k) {
            double f = std::min(std::max(freq[k], flimit_min), flimit_max);
            c[k] = alt ? (f - fsc_db) / fdev_db : (f - fsc_dr) / fdev_dr;
        }
        if (lf_rev.present()) c = lf_rev(c);
        vec dr, db;
        if (!alt) {
            dr = c;
            db = last_chroma;
        } else {
            dr = last_chroma;
            db = c;
        }
        last_chroma = c;
        last_frame = frame;
        last_line = line;
        return decode(luma, dr, db);
    }
};

Modem *build(const orc_desc_t &d) {
    Modem *m = nullptr;
    switch (d.kind) {
        case ORC_PAL_S: m = new PalS(d); break;
        case ORC_PAL_D: m = new PalD(d); break;
        case ORC_PAL_3D: m = new Pal3D(d); break;
        case ORC_NTSC: m = new Ntsc(d); break;
        case ORC_NTSC_COMB: m = new NtscComb(d); break;
        case ORC_SECAM: m = new Secam(d); break;
        default: g_error = "unknown modem kind"; return nullptr;
    }
    switch (d.wrapper) {
        case ORC_WRAP_NONE: break;
        case ORC_WRAP_SIMPLE_COMB:
        case ORC_WRAP_SIMPLE_3D_COMB:
            if (!m->has_components()) { /* the reference would raise AttributeError at the first row */
                delete m;
                g_error = "SimpleCombModem needs a backend with demodulate_components";
                return nullptr;
            }
            m = new SimpleComb(m, d.wrapper == ORC_WRAP_SIMPLE_3D_COMB, (d.use_minavg & 2) != 0, d.filters[ORC_F_WRAP_NOTCH]);
            break;
        case ORC_WRAP_COLOR_AVERAGING: m = new ColorAveraging(m); break;
        default:
            delete m;
            g_error = "unknown wrapper";
            return nullptr;
    }
    return m;
}

/* image.py:7-8 */
inline uint8_t as_byte(double x) {
    double c = std::max(std::min(x, 1.0), 0.0);
    return (uint8_t)std::nearbyint(255.0 * c); /* numpy.rint = round half to even (default FP mode) */
}

}  // namespace

struct orc_modem {
    orc_desc_t desc;
    std::unique_ptr<Modem> m;
};

extern "C" {

const char *orc_last_error(void) { return g_error.c_str(); }

orc_modem *orc_create(const orc_desc_t *desc) {
    for (int i = 0; i < ORC_MAX_FILTERS; ++i) {
        const orc_filter_t &f = desc->filters[i];
        if (f.present && (f.nb < 1 || f.na < 1 || f.nb > ORC_MAX_COEF || f.na > ORC_MAX_COEF)) {
            g_error = "filter coefficient count out of range";
            return nullptr;
        }
    }
    Modem *m = build(*desc);
    if (!m) return nullptr;
    orc_modem *o = new orc_modem;
    o->desc = *desc;
    o->m.reset(m);
    return o;
}
void orc_destroy(orc_modem *m) { delete m; }
int orc_modulation_delay(const orc_modem *m) { return m->m->modulation_delay; }
int orc_demodulation_delay(const orc_modem *m) { return m->m->demodulation_delay; }

int orc_modulate(orc_modem *m, int frame, int line, const double *r, const double *g, const double *b, int n,
                 double *composite) {
    vec out = m->m->modulate(frame, line, vec(r, r + n), vec(g, g + n), vec(b, b + n));
    std::copy(out.begin(), out.end(), composite);
    return 0;
}
int orc_demodulate(orc_modem *m, int frame, int line, const double *composite, int n, double *r, double *g,
                   double *b) {
    RGB out = m->m->demodulate(frame, line, vec(composite, composite + n));
    std::copy(out.r.begin(), out.r.end(), r);
    std::copy(out.g.begin(), out.g.end(), g);
    std::copy(out.b.begin(), out.b.end(), b);
    return 0;
}

/* the component-level protocol (qam.py / comb.py / secam.py *_components); returns -1 when the stack has none */
int orc_modulate_components(orc_modem *m, int frame, int line, const double *y, const double *u, const double *v, int n,
                            double *composite) {
    vec out = m->m->modulate_components(frame, line, vec(y, y + n), vec(u, u + n), vec(v, v + n));
    if (out.size() != (size_t)n) return -1;   /* the base class returns an empty row: no such member */
    std::copy(out.begin(), out.end(), composite);
    return 0;
}
int orc_demodulate_components(orc_modem *m, int frame, int line, const double *composite, int n, int strip_chroma,
                              double *y, double *u, double *v) {
    if (!m->m->has_components()) return -1;
    YUV out = m->m->demodulate_components(frame, line, vec(composite, composite + n), strip_chroma != 0);
    if (out.y.size() != (size_t)n) return -1;
    std::copy(out.y.begin(), out.y.end(), y);
    std::copy(out.u.begin(), out.u.end(), u);
    std::copy(out.v.begin(), out.v.end(), v);
    return 0;
}

/* image.py:47-55 */
int orc_modulate_frame(orc_modem *m, int frame, const double *rgb, double *composite) {
    const int W = m->desc.width, H = m->desc.height;
    const int delay = m->m->modulation_delay;
    const size_t plane = (size_t)W * H;
    auto row = [&](int c, int y) { return vec(rgb + c * plane + (size_t)y * W, rgb + c * plane + (size_t)(y + 1) * W); };
    for (int field = 0; field < 2; ++field) {
        for (int y = field; y < 2 * delay; y += 2)
            if (y < H) m->m->modulate(frame, y, row(0, y), row(1, y), row(2, y));
        for (int y = field; y < H; y += 2) {
            int iy = y + 2 * delay;
            while (iy >= H) iy -= 2;
            vec out = m->m->modulate(frame, y + 2 * delay, row(0, iy), row(1, iy), row(2, iy));
            std::copy(out.begin(), out.end(), composite + (size_t)y * W);
        }
    }
    return 0;
}
/* image.py:75-83 */
int orc_demodulate_frame(orc_modem *m, int frame, const double *composite, double *rgb) {
    const int W = m->desc.width, H = m->desc.height;
    const int delay = m->m->demodulation_delay;
    const size_t plane = (size_t)W * H;
    auto row = [&](int y) { return vec(composite + (size_t)y * W, composite + (size_t)(y + 1) * W); };
    for (int field = 0; field < 2; ++field) {
        for (int y = field; y < 2 * delay; y += 2)
            if (y < H) m->m->demodulate(frame, y, row(y));
        for (int y = field; y < H; y += 2) {
            int iy = y + 2 * delay;
            while (iy >= H) iy -= 2;
            RGB out = m->m->demodulate(frame, y + 2 * delay, row(iy));
            std::copy(out.r.begin(), out.r.end(), rgb + (size_t)y * W);
            std::copy(out.g.begin(), out.g.end(), rgb + plane + (size_t)y * W);
            std::copy(out.b.begin(), out.b.end(), rgb + 2 * plane + (size_t)y * W);
        }
    }
    return 0;
}

/* image.py:27-56 */
int orc_image_modulate(orc_modem *m, int frame, const uint8_t *rgb8, uint8_t *comp8) {
    const int W = m->desc.width, H = m->desc.height;
    std::vector<double> rgb((size_t)3 * W * H), comp((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) rgb[((size_t)c * H + y) * W + x] = rgb8[((size_t)y * W + x) * 3 + c] / 255.0;
    orc_modulate_frame(m, frame, rgb.data(), comp.data());
    for (size_t i = 0; i < comp.size(); ++i) comp8[i] = as_byte(0.6 * comp[i] + 0.2); /* image.py:16-21 */
    return 0;
}
/* image.py:58-84 */
int orc_image_demodulate(orc_modem *m, int frame, const uint8_t *comp8, uint8_t *rgb8) {
    const int W = m->desc.width, H = m->desc.height;
    std::vector<double> rgb((size_t)3 * W * H), comp((size_t)W * H);
    for (size_t i = 0; i < comp.size(); ++i) comp[i] = (5.0 * (comp8[i] / 255.0) - 1.0) / 3.0; /* image.py:24-25 */
    orc_demodulate_frame(m, frame, comp.data(), rgb.data());
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) rgb8[((size_t)y * W + x) * 3 + c] = as_byte(rgb[((size_t)c * H + y) * W + x]);
    return 0;
}

}  // extern "C"

static int run_frames(const orc_desc_t *desc, int64_t n_frames, int n_threads,
                      const std::function<void(orc_modem *, int64_t)> &body) {
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n_frames) n_threads = (int)std::max<int64_t>(1, n_frames);
    std::vector<std::thread> threads;
    std::vector<int> status(n_threads, 0);
    for (int t = 0; t < n_threads; ++t) {
        threads.emplace_back([&, t]() {
            orc_modem *m = orc_create(desc);
            if (!m) {
                status[t] = -1;
                return;
            }
            int64_t lo = n_frames * t / n_threads, hi = n_frames * (t + 1) / n_threads;
            for (int64_t f = lo; f < hi; ++f) body(m, f);
            orc_destroy(m);
        });
    }
    for (auto &th : threads) th.join();
    for (int s : status)
        if (s) return s;
    return 0;
}

extern "C" {

int orc_demodulate_frames_f32(const orc_desc_t *desc, const float *composite, float *rgb, int64_t n_frames,
                              int64_t first_frame, int n_threads) {
    const size_t plane = (size_t)desc->width * desc->height;
    return run_frames(desc, n_frames, n_threads, [&](orc_modem *m, int64_t f) {
        std::vector<double> in(composite + f * plane, composite + (f + 1) * plane), out(3 * plane);
        orc_demodulate_frame(m, (int)(first_frame + f), in.data(), out.data());
        float *dst = rgb + f * 3 * plane;
        for (size_t i = 0; i < 3 * plane; ++i) dst[i] = (float)out[i];
    });
}
int orc_modulate_frames_f32(const orc_desc_t *desc, const float *rgb, float *composite, int64_t n_frames,
                            int64_t first_frame, int n_threads) {
    const size_t plane = (size_t)desc->width * desc->height;
    return run_frames(desc, n_frames, n_threads, [&](orc_modem *m, int64_t f) {
        std::vector<double> in(rgb + f * 3 * plane, rgb + (f + 1) * 3 * plane), out(plane);
        orc_modulate_frame(m, (int)(first_frame + f), in.data(), out.data());
        float *dst = composite + f * plane;
        for (size_t i = 0; i < plane; ++i) dst[i] = (float)out[i];
    });
}

void orc_firwin41(double *h41) { std::memcpy(h41, fir().h, sizeof(double) * 41); }
void orc_resample_up2(const double *x, int n, double *y) {
    vec out = up2(vec(x, x + n));
    std::copy(out.begin(), out.end(), y);
}
int orc_resample_dn2(const double *x, int n, double *y) {
    vec out = dn2(vec(x, x + n));
    std::copy(out.begin(), out.end(), y);
    return (int)out.size();
}
void orc_filter_apply(const orc_filter_t *f, const double *x, int n, double *y) {
    Filter ff;
    ff.f = *f;
    vec out = ff(vec(x, x + n));
    std::copy(out.begin(), out.end(), y);
}
double orc_start_phase(const orc_desc_t *d, int frame, int line) {
    PalS tmp(*d); /* start_phase lives in the shared QamBackend base */
    return tmp.start_phase(frame, line);
}
int orc_analog_line(const orc_desc_t *d, int line) { return LineConfig(*d).analog_line(line); }
int orc_is_alternate_line(const orc_desc_t *d, int frame, int line) {
    return LineConfig(*d).is_alternate_line(frame, line) ? 1 : 0;
}

}  // extern "C"

// tangent_oracle.cpp -- ORACLE tooling (test infrastructure only; see gradus_oracle.h).
//
// The oracle (gradus_oracle.c, UNCHANGED source) compiled with `double` replaced by a forward-mode number that carries
// two tangent directions, ∂/∂α and ∂/∂β of the image-plane impact parameters.  Every arithmetic operation of the CPU
// restatement -- map_impact_parameters, constrain_all, the Tsit5 stages in first-order form, the dense output, the
// redshift -- then propagates the derivatives, which is what the reference obtains from ForwardDiff.jacobian around
// tracegeodesics (src/tracing/precision-solvers.jl:401-451, jacobian_∂αβ_∂gr) and from the Dual-valued integrator of
// _make_image_plane_mapper (:73-131).  It pins the product's tangent kernels (gr_tangent.hpp, gr_ray_tangent) PER RAY;
// the recorded transfer-function statistics of the reference only pin means over 114 samples.
//
// Shared with the product: nothing.  This scalar type is written here for the oracle; the integrator is the oracle's
// own (first-order form, library sin/cos, true divisions, bisection-class root find of the event).
//
// Semantics restated (third party: DiffEqBase / ForwardDiff, absent from /root/reference):
//   * comparisons, the accept test and the step-size controller read VALUES.  (DiffEqBase's ForwardDiff extension folds
//     the partials into the error norm; orct_set_norm(1) switches that on here: EEst² = Σ_i sse(ũ_i / sc_i) / (8 · 3),
//     sc_i = abstol + reltol max(|u0_i|_D, |u1_i|_D), |u|_D = sqrt(v² + a² + b²).)
//   * event time: with Dual state and a ContinuousCallback DiffEqBase promotes the time span to Dual, so the event
//     time carries derivatives: c(x(λ*; α, β)) = 0  =>  ∂λ* = -∂c|_λ / (dc/dλ).  The root find of the oracle works
//     on values (Θ is a plain number), so the state it returns carries the tangents at FIXED λ; the wrapper below adds
//     (v, a) ∂λ* with a = the geodesic right-hand side AT the end point (not the interpolant's derivative that the
//     product uses: the two differ by the interpolation error, O(h⁴)).
//
//   make -C oracle tangent   ->   oracle/libgradus_oracle_tangent.so
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct T2 {
    double v, a, b;
    T2() : v(0.0), a(0.0), b(0.0) {}
    T2(double x) : v(x), a(0.0), b(0.0) {}
    T2(long double x) : v((double)x), a(0.0), b(0.0) {}
    T2(int x) : v((double)x), a(0.0), b(0.0) {}
    T2(long x) : v((double)x), a(0.0), b(0.0) {}
    T2(long long x) : v((double)x), a(0.0), b(0.0) {}
    T2(double x, double da, double db) : v(x), a(da), b(db) {}
    template <class T> explicit operator T() const { return (T)v; }
};
static inline T2 operator+(T2 x, T2 y) { return T2(x.v + y.v, x.a + y.a, x.b + y.b); }
static inline T2 operator-(T2 x, T2 y) { return T2(x.v - y.v, x.a - y.a, x.b - y.b); }
static inline T2 operator*(T2 x, T2 y) { return T2(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b); }
static inline T2 operator/(T2 x, T2 y)
{
    const double q = x.v / y.v;
    return T2(q, (x.a - q * y.a) / y.v, (x.b - q * y.b) / y.v);
}
static inline T2 operator-(T2 x) { return T2(-x.v, -x.a, -x.b); }
static inline T2& operator+=(T2& x, T2 y) { x = x + y; return x; }
static inline T2& operator-=(T2& x, T2 y) { x = x - y; return x; }
static inline T2& operator*=(T2& x, T2 y) { x = x * y; return x; }
static inline T2& operator/=(T2& x, T2 y) { x = x / y; return x; }
#define CMP(op) static inline bool operator op(T2 x, T2 y) { return x.v op y.v; }
CMP(<) CMP(>) CMP(<=) CMP(>=) CMP(==) CMP(!=)
#undef CMP
static inline T2 sqrt(T2 x) { const double s = std::sqrt(x.v); return T2(s, 0.5 * x.a / s, 0.5 * x.b / s); }
static inline T2 cbrt(T2 x) { const double c = std::cbrt(x.v), d = c / (3.0 * x.v); return T2(c, d * x.a, d * x.b); }
static inline T2 sin(T2 x) { const double s = std::sin(x.v), c = std::cos(x.v); return T2(s, c * x.a, c * x.b); }
static inline T2 cos(T2 x) { const double s = std::sin(x.v), c = std::cos(x.v); return T2(c, -s * x.a, -s * x.b); }
static inline T2 atan(T2 x) { const double w = 1.0 / (1.0 + x.v * x.v); return T2(std::atan(x.v), w * x.a, w * x.b); }
static inline T2 atan2(T2 y, T2 x)
{
    const double n = x.v * x.v + y.v * y.v;
    return T2(std::atan2(y.v, x.v), (x.v * y.a - y.v * x.a) / n, (x.v * y.b - y.v * x.b) / n);
}
static inline T2 pow(T2 x, T2 y)
{
    // d(x^y) = y x^(y-1) dx + x^y ln x dy ; the oracle only raises to constant powers (dy = 0), the general rule is kept
    const double p = std::pow(x.v, y.v);
    const double dx = y.v * std::pow(x.v, y.v - 1.0);
    const double dy = (y.a != 0.0 || y.b != 0.0) ? p * std::log(x.v) : 0.0;
    return T2(p, dx * x.a + dy * y.a, dx * x.b + dy * y.b);
}
static inline T2 log10(T2 x) { const double d = 1.0 / (x.v * M_LN10); return T2(std::log10(x.v), d * x.a, d * x.b); }
static inline T2 fabs(T2 x) { return x.v < 0.0 ? -x : x; }
static inline T2 floor(T2 x) { return T2(std::floor(x.v)); }
static inline T2 fmax(T2 x, T2 y) { return (x.v >= y.v || y.v != y.v) ? x : y; }
static inline T2 fmin(T2 x, T2 y) { return (x.v <= y.v || y.v != y.v) ? x : y; }
static inline T2 fma(T2 x, T2 y, T2 z) { return x * y + z; }
static inline bool isnan(T2 x) { return std::isnan(x.v); }
static inline bool isfinite(T2 x) { return std::isfinite(x.v); }
static inline bool isinf(T2 x) { return std::isinf(x.v); }

#ifdef _OPENMP
#include <omp.h>
#endif
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

// 1 = the error norm over values and tangents (consulted by rms8 of the oracle through ORC_TANGENT_NORM_HOOK)
static int g_norm_with_tangents = 0;
static inline double orct_sse(const T2& x) { return x.v * x.v + x.a * x.a + x.b * x.b; }
#define ORC_TANGENT_BUILD 1

#define double T2
#include "gradus_oracle.c"
#undef double

// ---- plain-double mirrors of the oracle's C structs (the header above was compiled with double = T2) ----
struct cfg_plain {
    int32_t metric_id, disc_id;
    double params[8];
    double r_inner, r_outer, disc_r_in, disc_r_out, gtol, lambda0, lambda1, abstol, reltol, mu;
    int64_t maxiters;
    int32_t upper_hemisphere, _pad;
    double hemi_delta;
    double disc_params[4];
    const double* disc_table;
    int64_t disc_table_n;
    const double* chart_table;
    int64_t chart_table_n;
    double chart_theta0, chart_theta1, q;
    int32_t count_windings, _pad2;
    double winding_plane;
};
struct pf_plain {
    int32_t pf_id, filter_id;
    double fill, r_isco;
    int64_t n_plunge;
    const double *plunge_r, *plunge_vt, *plunge_vr, *plunge_vphi;
};

static void to_t2(const cfg_plain& p, orc_config& c)
{
    memset((void*)&c, 0, sizeof c);
    c.metric_id = p.metric_id; c.disc_id = p.disc_id;
    for (int i = 0; i < 8; ++i) c.params[i] = p.params[i];
    c.r_inner = p.r_inner; c.r_outer = p.r_outer; c.disc_r_in = p.disc_r_in; c.disc_r_out = p.disc_r_out; c.gtol = p.gtol;
    c.lambda0 = p.lambda0; c.lambda1 = p.lambda1; c.abstol = p.abstol; c.reltol = p.reltol; c.mu = p.mu;
    c.maxiters = p.maxiters; c.upper_hemisphere = p.upper_hemisphere; c.hemi_delta = p.hemi_delta;
    for (int i = 0; i < 4; ++i) c.disc_params[i] = p.disc_params[i];
    c.disc_table = nullptr; c.disc_table_n = 0;       // tabulated discs / charts are not needed for the per-ray pin
    c.chart_table = nullptr; c.chart_table_n = 0;
    c.chart_theta0 = p.chart_theta0; c.chart_theta1 = p.chart_theta1; c.q = p.q;
    c.count_windings = p.count_windings; c.winding_plane = p.winding_plane;
}

extern "C" {

void orct_set_norm(int with_tangents) { g_norm_with_tangents = with_tangents ? 1 : 0; }

// Rays given by impact parameters (α_i, β_i) from the observer x_obs; out: n x 8 doubles
//   (g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, coordinate time t of the end point, status)
// -- the record of the product's gr_ray_tangent.  g = NaN unless the ray met the geometry.  Per-ray datum-plane heights
// (DatumPlane per emission radius) may be given; NULL = cfg->disc_params[0].
int orct_ray_tangent(const void* cfg_in, const void* pf_in, const double* x_obs, const double* alpha, const double* beta,
                     const double* heights, int64_t n, double max_time, double* out)
{
    const cfg_plain& cp = *(const cfg_plain*)cfg_in;
    const pf_plain& pp = *(const pf_plain*)pf_in;
    if (cp.disc_table_n != 0 || cp.chart_table_n != 0 || pp.n_plunge != 0) return -1;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < n; ++i) {
        orc_config c;
        to_t2(cp, c);
        if (heights && c.disc_id == ORC_DISC_DATUM) c.disc_params[0] = heights[i];
        orc_pf pf;
        memset((void*)&pf, 0, sizeof pf);
        pf.pf_id = pp.pf_id; pf.filter_id = ORC_FILTER_NONE; pf.fill = pp.fill; pf.r_isco = pp.r_isco;
        T2 x[4] = { x_obs[0], x_obs[1], x_obs[2], x_obs[3] };
        const T2 al(alpha[i], 1.0, 0.0), be(beta[i], 0.0, 1.0);
        T2 v[4];
        orc_map_impact_parameters(&c, x, al, be, v);
        orc_point pt;
        orc_trace(&c, x, 0, v, 1, &pt, nullptr, 1);
        double* o = out + 8 * i;
        const bool hit = pt.status == ORC_INTERSECTED_WITH_GEOMETRY;
        if (hit) {
            // implicit differentiation of the event time (see the header comment)
            T2 u[8] = { pt.x[0], pt.x[1], pt.x[2], pt.x[3], pt.v[0], pt.v[1], pt.v[2], pt.v[3] };
            const T2 cv = disc_condition(&c, u);
            T2 ud[8];
            for (int k = 0; k < 4; ++k) { ud[k] = T2(pt.x[k].v, pt.v[k].v, 0.0); ud[4 + k] = T2(pt.v[k].v); }
            const double cdot = disc_condition(&c, ud).a;
            if (cdot != 0.0) {
                const double la = -cv.a / cdot, lb = -cv.b / cdot;
                T2 acc[4];
                orc_geodesic_equation(&c, pt.x, pt.v, acc);
                for (int k = 0; k < 4; ++k) {
                    const double vel = pt.v[k].v, ak = acc[k].v;
                    pt.x[k].a += vel * la; pt.x[k].b += vel * lb;
                    pt.v[k].a += ak * la; pt.v[k].b += ak * lb;
                }
            }
        }
        const T2 rho = pt.x[1] * fabs(sin(pt.x[2]));
        T2 g(NAN);
        if (hit) orc_apply_pf(&c, &pf, &pt, 1, T2(max_time), &g, 1);
        o[0] = g.v; o[1] = rho.v; o[2] = g.a; o[3] = g.b; o[4] = rho.a; o[5] = rho.b; o[6] = pt.x[0].v; o[7] = (double)pt.status;
    }
    return 0;
}
}

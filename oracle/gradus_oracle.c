/*
 * gradus_oracle.c -- CPU ORACLE (test infrastructure only; see gradus_oracle.h).
 *
 * Plain-C restatement of Gradus.jl's image-plane render path.  Citations are into
 * /root/reference (Gradus.jl v0.4.30).  "[3P]" marks behaviour of OrdinaryDiffEq.jl /
 * DiffEqBase.jl / ForwardDiff.jl, which are not vendored by the reference and whose
 * versions it does not pin; SURVEY.md App. A documents what is restated.
 */
#include "gradus_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------
 * Forward-mode dual numbers with two partials (d/dr, d/dθ): what ForwardDiff.Dual{Tag,
 * Float64,2} carries in metric_jacobian (auto-diff.jl:206-211) [3P].
 * ---------------------------------------------------------------------------------- */
typedef struct { double v, a, b; } d2;

static inline d2 d2_const(double x) { d2 r = { x, 0.0, 0.0 }; return r; }
static inline d2 d2_add(d2 x, d2 y) { d2 r = { x.v + y.v, x.a + y.a, x.b + y.b }; return r; }
static inline d2 d2_sub(d2 x, d2 y) { d2 r = { x.v - y.v, x.a - y.a, x.b - y.b }; return r; }
static inline d2 d2_neg(d2 x) { d2 r = { -x.v, -x.a, -x.b }; return r; }
static inline d2 d2_scale(double s, d2 x) { d2 r = { s * x.v, s * x.a, s * x.b }; return r; }
static inline d2 d2_mul(d2 x, d2 y)
{
    d2 r = { x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b };
    return r;
}
static inline d2 d2_div(d2 x, d2 y)
{
    double q = x.v / y.v, iy = 1.0 / y.v;
    d2 r = { q, (x.a - q * y.a) * iy, (x.b - q * y.b) * iy };
    return r;
}
static inline d2 d2_sin(d2 x) { double s = sin(x.v), c = cos(x.v); d2 r = { s, c * x.a, c * x.b }; return r; }
static inline d2 d2_cos(d2 x) { double s = sin(x.v), c = cos(x.v); d2 r = { c, -s * x.a, -s * x.b }; return r; }

static inline d2 d2_atan(d2 x) { double w = 1.0 / (1.0 + x.v * x.v); d2 r = { atan(x.v), w * x.a, w * x.b }; return r; }

#define NUM d2
#define N_ATAN d2_atan
#define N_VAL(x) ((x).v)
#define FN(name) name##_d2
#define N_CONST d2_const
#define N_ADD d2_add
#define N_SUB d2_sub
#define N_MUL d2_mul
#define N_DIV d2_div
#define N_SCALE d2_scale
#define N_NEG d2_neg
#define N_SIN d2_sin
#define N_COS d2_cos
#include "metrics_tmpl.h"
#undef NUM
#undef FN
#undef N_CONST
#undef N_ADD
#undef N_SUB
#undef N_MUL
#undef N_DIV
#undef N_SCALE
#undef N_NEG
#undef N_SIN
#undef N_COS
#undef N_ATAN
#undef N_VAL

/* Second-order jet in one variable (r) at fixed θ: value, d/dr, d²/dr².  Needed only for
 * the generic ISCO condition dE/dr = 0 (special-radii.jl:20-23), where the reference nests
 * ForwardDiff.derivative around CircularOrbits.energy (itself using ∂_r g). */
typedef struct { double v, d, dd; } j2;
static inline j2 j2_const(double x) { j2 r = { x, 0.0, 0.0 }; return r; }
static inline j2 j2_add(j2 x, j2 y) { j2 r = { x.v + y.v, x.d + y.d, x.dd + y.dd }; return r; }
static inline j2 j2_sub(j2 x, j2 y) { j2 r = { x.v - y.v, x.d - y.d, x.dd - y.dd }; return r; }
static inline j2 j2_neg(j2 x) { j2 r = { -x.v, -x.d, -x.dd }; return r; }
static inline j2 j2_scale(double s, j2 x) { j2 r = { s * x.v, s * x.d, s * x.dd }; return r; }
static inline j2 j2_mul(j2 x, j2 y)
{
    j2 r = { x.v * y.v, x.d * y.v + x.v * y.d, x.dd * y.v + 2.0 * x.d * y.d + x.v * y.dd };
    return r;
}
static inline j2 j2_inv(j2 y)
{
    double i = 1.0 / y.v;
    j2 r = { i, -y.d * i * i, (2.0 * y.d * y.d * i - y.dd) * i * i };
    return r;
}
static inline j2 j2_div(j2 x, j2 y) { return j2_mul(x, j2_inv(y)); }
/* θ is a constant in this jet: sin/cos just act on the value */
static inline j2 j2_sin(j2 x) { return j2_const(sin(x.v)); }
static inline j2 j2_cos(j2 x) { return j2_const(cos(x.v)); }

static inline j2 j2_atan(j2 x)
{
    double w = 1.0 / (1.0 + x.v * x.v);
    j2 r = { atan(x.v), w * x.d, w * x.dd - 2.0 * x.v * x.d * x.d * w * w };
    return r;
}

#define NUM j2
#define N_ATAN j2_atan
#define N_VAL(x) ((x).v)
#define FN(name) name##_j2
#define N_CONST j2_const
#define N_ADD j2_add
#define N_SUB j2_sub
#define N_MUL j2_mul
#define N_DIV j2_div
#define N_SCALE j2_scale
#define N_NEG j2_neg
#define N_SIN j2_sin
#define N_COS j2_cos
#include "metrics_tmpl.h"
#undef NUM
#undef FN
#undef N_CONST
#undef N_ADD
#undef N_SUB
#undef N_MUL
#undef N_DIV
#undef N_SCALE
#undef N_NEG
#undef N_SIN
#undef N_COS
#undef N_ATAN
#undef N_VAL

/* ------------------------------------------------------------------------------------
 * metric_jacobian, auto-diff.jl:206-211
 * ---------------------------------------------------------------------------------- */
static void metric_d2(const orc_config* c, double r, double th, d2 g[5])
{
    d2 rr = { r, 1.0, 0.0 }, tt = { th, 0.0, 1.0 };
    switch (c->metric_id) {
    case ORC_METRIC_JOHANNSEN: johannsen_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_MORRIS_THORNE: morris_thorne_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_BUMBLEBEE: bumblebee_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_KERR_NEWMAN: kerr_newman_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_JOHANNSEN_PSALTIS: johannsen_psaltis_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_DILATON_AXION: dilaton_axion_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_SPHERICAL: spherical_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_KERR_DARK_MATTER: kerr_dark_matter_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_KERR_REFRACTIVE: kerr_refractive_components_d2(c->params, rr, tt, g); break;
    case ORC_METRIC_NOZ: noz_components_d2(c->params, rr, tt, g); break;
#ifdef ORC_WITH_TEST_METRIC
    case ORC_METRIC_TEST_BUMP: test_bump_components_d2(c->params, rr, tt, g); break;
#endif
    default: kerr_components_d2(c->params, rr, tt, g);
    }
}

void orc_metric_jacobian(const orc_config* c, double r, double th, double g[5], double dr[5], double dth[5])
{
    d2 gd[5];
    metric_d2(c, r, th, gd);
    for (int i = 0; i < 5; ++i) {
        g[i] = gd[i].v;
        dr[i] = gd[i].a;
        dth[i] = gd[i].b;
    }
}

/* inverse_metric_components, auto-diff.jl:59-76 */
static void inverse_metric_components(const double g[5], double gi[5])
{
    const double g1 = g[0], g2 = g[1], g3 = g[2], g4 = g[3], g5 = g[4];
    const double term = g1 * g2 * g3 * g4 - (g5 * g5) * g2 * g3;
    const double D = 1.0 / term;
    gi[0] = (g2 * g3 * g4) * D;
    gi[1] = (g1 * g3 * g4 - (g5 * g5) * g3) * D;
    gi[2] = (g1 * g2 * g4 - (g5 * g5) * g2) * D;
    gi[3] = (g1 * g2 * g3) * D;
    gi[4] = (-g2 * g3 * g5) * D;
}

/* compute_geodesic_equation, auto-diff.jl:115-141.  The reference builds
 * Γ[i,k,l] = ginv[i,m]*(jac[l][m,k] + jac[k][m,l] - jac[m][k,l]) symbolically with the ½
 * deferred, then returns -½ (Γ^i v)·v.  With ∂_t = ∂_φ = 0 the contraction is: */
static void compute_geodesic_equation(const double gi[5], const double j1[5], const double j2_[5],
                                      const double v[4], double acc[4])
{
    const double vt = v[0], vr = v[1], vh = v[2], vp = v[3];
    double gd[5];
    for (int k = 0; k < 5; ++k) gd[k] = j1[k] * vr + j2_[k] * vh;   /* dg_k/dλ */
    const double Dr = j1[0] * vt * vt + j1[1] * vr * vr + j1[2] * vh * vh + j1[3] * vp * vp
                      + 2.0 * j1[4] * vt * vp;
    const double Dh = j2_[0] * vt * vt + j2_[1] * vr * vr + j2_[2] * vh * vh + j2_[3] * vp * vp
                      + 2.0 * j2_[4] * vt * vp;
    const double St = 2.0 * (gd[0] * vt + gd[4] * vp);
    const double Sr = 2.0 * gd[1] * vr - Dr;
    const double Sh = 2.0 * gd[2] * vh - Dh;
    const double Sp = 2.0 * (gd[4] * vt + gd[3] * vp);
    acc[0] = -0.5 * (gi[0] * St + gi[4] * Sp);
    acc[1] = -0.5 * (gi[1] * Sr);
    acc[2] = -0.5 * (gi[2] * Sh);
    acc[3] = -0.5 * (gi[4] * St + gi[3] * Sp);
}

/* geodesic_equation, auto-diff.jl:213-226 */
void orc_geodesic_equation(const orc_config* c, const double x[4], const double v[4], double acc[4])
{
    double g[5], j1[5], j2_[5], gi[5];
    orc_metric_jacobian(c, x[1], x[2], g, j1, j2_);
    inverse_metric_components(g, gi);
    compute_geodesic_equation(gi, j1, j2_, v, acc);
}

/* electromagnetic_potential of Kerr-Newman on dual numbers, kerr-newman-ad.jl:29-33:
 * A = (r Q / Σ) (1, 0, 0, -a sin²θ); only A_t and A_ϕ are non-zero */
static void kerr_newman_potential_d2(const double* p, d2 r, d2 th, d2* At, d2* Ap)
{
    const double a = p[1], Q = p[2];
    d2 c = d2_cos(th), s = d2_sin(th);
    d2 ac = d2_scale(a, c);
    d2 Sig = d2_add(d2_mul(r, r), d2_mul(ac, ac));
    d2 P = d2_div(d2_scale(Q, r), Sig);
    *At = P;
    *Ap = d2_mul(P, d2_scale(-a, d2_mul(s, s)));
}

/* q/μ · F^μ_κ v^κ with F = g⁻¹ (∂A - ∂Aᵀ), faraday_tensor, tracing/utility.jl:89-99, as added to
 * the geodesic acceleration by geodesic_ode_problem(::KerrNewmanMetric), kerr-newman-ad.jl:66-100 */
static void kerr_newman_lorentz_force(const orc_config* c, const double x[4], const double v[4], double f[4])
{
    d2 rr = { x[1], 1.0, 0.0 }, tt = { x[2], 0.0, 1.0 }, At, Ap;
    kerr_newman_potential_d2(c->params, rr, tt, &At, &Ap);
    double g[5], j1[5], j2_[5], gi[5];
    orc_metric_jacobian(c, x[1], x[2], g, j1, j2_);
    inverse_metric_components(g, gi);
    /* w_σ = (∂_κ A_σ - ∂_σ A_κ) v^κ ; ∂_t = ∂_ϕ = 0, A_r = A_θ = 0 */
    const double wt = At.a * v[1] + At.b * v[2];
    const double wp = Ap.a * v[1] + Ap.b * v[2];
    const double wr = -(At.a * v[0] + Ap.a * v[3]);
    const double wh = -(At.b * v[0] + Ap.b * v[3]);
    const double qm = (fabs(c->mu) < 1.4901161193847656e-08) ? c->q : c->q / c->mu;   /* trace.μ ≈ 0 */
    f[0] = qm * (gi[0] * wt + gi[4] * wp);
    f[1] = qm * (gi[1] * wr);
    f[2] = qm * (gi[2] * wh);
    f[3] = qm * (gi[4] * wt + gi[3] * wp);
}

/* _second_order_ode_f, geodesic-problem.jl:87-92 */
static void rhs(const orc_config* c, const double u[8], double du[8])
{
    du[0] = u[4]; du[1] = u[5]; du[2] = u[6]; du[3] = u[7];
    orc_geodesic_equation(c, u, u + 4, du + 4);
    if (c->metric_id == ORC_METRIC_KERR_NEWMAN && c->q != 0.0) {
        double f[4];
        kerr_newman_lorentz_force(c, u, u + 4, f);
        for (int i = 0; i < 4; ++i) du[4 + i] += f[i];
    }
}

static void metric_components(const orc_config* c, double r, double th, double g[5])
{
    d2 gd[5];
    metric_d2(c, r, th, gd);
    for (int i = 0; i < 5; ++i) g[i] = gd[i].v;
}

/* constrain_time, auto-diff.jl:161-179 ; constrain :172-176 */
static double constrain_time(const double g[5], const double v[4], double mu)
{
    const double disc = -g[0] * g[1] * v[1] * v[1] - g[0] * g[2] * v[2] * v[2] - g[0] * mu * mu
                        - (g[0] * g[3] - g[4] * g[4]) * v[3] * v[3];
    return -(g[4] * v[3] + sqrt(disc)) / g[0];
}

double orc_constrain_time(const orc_config* c, const double x[4], const double v[4])
{
    double g[5];
    metric_components(c, x[1], x[2], g);
    return constrain_time(g, v, c->mu);
}

/* ------------------------------------------------------------------------------------
 * 4x4 helpers; metric(m,x) = _symmetric_matrix(comps), utils.jl:60-67, auto-diff.jl:228-232
 * ---------------------------------------------------------------------------------- */
static void sym_matrix(const double g[5], double G[4][4])
{
    memset(G, 0, 16 * sizeof(double));
    G[0][0] = g[0]; G[1][1] = g[1]; G[2][2] = g[2]; G[3][3] = g[3];
    G[0][3] = g[4]; G[3][0] = g[4];
}
static void sym_matrix_inv(const double g[5], double G[4][4])
{
    /* inv(metric) of the block form; the reference calls StaticArrays.inv on the 4x4 */
    const double D = g[0] * g[3] - g[4] * g[4];
    memset(G, 0, 16 * sizeof(double));
    G[0][0] = g[3] / D; G[3][3] = g[0] / D; G[0][3] = -g[4] / D; G[3][0] = -g[4] / D;
    G[1][1] = 1.0 / g[1]; G[2][2] = 1.0 / g[2];
}
/* dotproduct(g, v1, v2) = _fast_dot(g*v1, v2), orthonormalization.jl:3-16 */
static double mdot(double G[4][4], const double a[4], const double b[4])
{
    double res = 0.0;
    for (int i = 0; i < 4; ++i) {
        double gi = 0.0;
        for (int j = 0; j < 4; ++j) gi += G[i][j] * a[j];
        res = fma(gi, b[i], res);
    }
    return res;
}

/* projectbasis, orthonormalization.jl:29-35 */
static void projectbasis(double G[4][4], double basis[][4], int nb, const double v[4], double s[4])
{
    s[0] = s[1] = s[2] = s[3] = 0.0;
    for (int e = 0; e < nb; ++e) {
        const double w = mdot(G, v, basis[e]) / mdot(G, basis[e], basis[e]); /* mproject :27 */
        for (int i = 0; i < 4; ++i) s[i] += w * basis[e][i];
    }
}

/* gramschmidt, orthonormalization.jl:37-48 */
static void gramschmidt(const double vin[4], double basis[][4], int nb, double G[4][4], double out[4])
{
    const double tol = 4.0 * DBL_EPSILON;
    double v[4], p[4];
    memcpy(v, vin, sizeof v);
    projectbasis(G, basis, nb, v, p);
    int guard = 0;
    while (p[0] + p[1] + p[2] + p[3] > tol && guard++ < 1000) {
        for (int i = 0; i < 4; ++i) v[i] -= p[i];
        projectbasis(G, basis, nb, v, p);
    }
    for (int i = 0; i < 4; ++i) v[i] -= p[i];
    const double n = sqrt(fabs(mdot(G, v, v)));
    for (int i = 0; i < 4; ++i) out[i] = v[i] / n;
}

static void tetrad_permute_state(const int s[4], int o[4]) { o[0] = s[0]; o[1] = s[3]; o[2] = s[1]; o[3] = s[2]; }

/* tetradframe, orthonormalization.jl:75-103.  ret[k][:] are the four vectors, ordered t,r,θ,ϕ */
static void tetradframe(double G[4][4], const double vin[4], double ret[4][4])
{
    double vs[4][4];
    const double n = sqrt(fabs(mdot(G, vin, vin)));
    for (int i = 0; i < 4; ++i) vs[0][i] = vin[i] / n;
    int state[4], perm[4], cnt = 0;
    for (int i = 0; i < 4; ++i) { state[i] = (vs[0][i] != 0.0); cnt += state[i]; }
    if (cnt == 1) { state[0] = 1; state[1] = 0; state[2] = 0; state[3] = 1; }
    /* permutations = searchsortedfirst(state[2:end], 1) */
    int permutations = 4;
    for (int i = 1; i < 4; ++i) if (state[i] >= 1) { permutations = i; break; }
    double sv[4];
    for (int i = 0; i < 4; ++i) sv[i] = (double)state[i];
    gramschmidt(sv, vs, 1, G, vs[1]);
    tetrad_permute_state(state, perm);
    for (int i = 0; i < 4; ++i) { state[i] |= perm[i]; sv[i] = (double)state[i]; }
    gramschmidt(sv, vs, 2, G, vs[2]);
    tetrad_permute_state(state, perm);
    for (int i = 0; i < 4; ++i) { state[i] |= perm[i]; sv[i] = (double)state[i]; }
    gramschmidt(sv, vs, 3, G, vs[3]);
    int order[4] = { 0, 1, 2, 3 };
    for (int k = 2; k <= permutations; ++k) {
        int o2[4] = { order[0], order[3], order[1], order[2] };
        memcpy(order, o2, sizeof o2);
    }
    for (int k = 0; k < 4; ++k) memcpy(ret[k], vs[order[k]], 4 * sizeof(double));
}

/* lnrframe, orthonormalization.jl:106-111 */
void orc_lnrframe(const orc_config* c, const double x[4], double F[16])
{
    double g[5], G[4][4], ret[4][4];
    metric_components(c, x[1], x[2], g);
    sym_matrix(g, G);
    const double om = -G[0][3] / G[3][3];
    const double v[4] = { 1.0, 0.0, 0.0, om };
    tetradframe(G, v, ret);
    /* hcat(vecs...): column k = vector k */
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 4; ++i) F[i * 4 + k] = ret[k][i];
}

/* lnrbasis, orthonormalization.jl:114-122 */
void orc_lnrbasis(const orc_config* c, const double x[4], double Tx[16])
{
    double g[5], G[4][4], Gi[4][4], ret[4][4];
    metric_components(c, x[1], x[2], g);
    sym_matrix(g, G);
    sym_matrix_inv(g, Gi);
    const double om = -G[0][3] / G[3][3];
    const double v[4] = { -om, 0.0, 0.0, 1.0 };
    tetradframe(Gi, v, ret);           /* (vϕ, vr, vθ, vt) */
    const int map[4] = { 3, 1, 2, 0 }; /* rearranged to (vt, vr, vθ, vϕ) */
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 4; ++i) Tx[i * 4 + k] = ret[map[k]][i];
}

/* lnr_momentum_to_global_velocity_transform, tracing/utility.jl:32-40:  p̄ -> ginv*(Tx*p̄) */
void orc_lnr_transform(const orc_config* c, const double x[4], double Mx[16])
{
    double Tx[16], g[5], Gi[4][4];
    orc_lnrbasis(c, x, Tx);
    metric_components(c, x[1], x[2], g);
    sym_matrix_inv(g, Gi);
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            double s = 0.0;
            for (int j = 0; j < 4; ++j) s += Gi[i][j] * Tx[j * 4 + k];
            Mx[i * 4 + k] = s;
        }
}

/* local_momentum, tracing/utility.jl:13-20 */
static void local_momentum(double r_obs, double alpha, double beta, double p[4])
{
    const double b = beta / r_obs, a = alpha / r_obs;
    const double pr = -1.0 / sqrt(1.0 + a * a + b * b);
    p[0] = 1.0; p[1] = pr; p[2] = b * pr; p[3] = a * pr;
}

static void apply_Mx(const double Mx[16], const double p[4], double v[4])
{
    for (int i = 0; i < 4; ++i)
        v[i] = Mx[i * 4 + 0] * p[0] + Mx[i * 4 + 1] * p[1] + Mx[i * 4 + 2] * p[2] + Mx[i * 4 + 3] * p[3];
}

void orc_map_impact_parameters(const orc_config* c, const double x[4], double alpha, double beta, double v[4])
{
    double Mx[16], p[4];
    orc_lnr_transform(c, x, Mx);
    local_momentum(x[1], alpha, beta, p);
    apply_Mx(Mx, p, v);
}

/* range(a, b, n)[k+1], k 0-based.  Julia evaluates this in twice precision; a lerp is within
 * an ulp of it. */
static double range_at(double a, double b, int64_t n, int64_t k)
{
    if (n <= 1) return a;
    const double t = (double)k / (double)(n - 1);
    return (1.0 - t) * a + t * b;
}

/* _render_velocity_function, rendering/rendering.jl:140-163 */
void orc_render_velocities(const orc_config* c, const double x[4], double a0, double a1, double b0,
                           double b1, int64_t W, int64_t H, int64_t i0, int64_t n, double* v_out)
{
    double Mx[16];
    orc_lnr_transform(c, x, Mx);
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = i0 + k;          /* 0-based */
        const int64_t xi = i / H;          /* (i-1) ÷ H + 1, 0-based */
        const int64_t yi = i % H;          /* mod1(i, H), 0-based */
        const double alpha = range_at(a0, a1, W, xi) + 1e-6;
        const double beta = range_at(b0, b1, H, yi) + 1e-6;
        double p[4];
        local_momentum(x[1], alpha, beta, p);
        apply_Mx(Mx, p, v_out + 4 * k);
    }
}

/* ------------------------------------------------------------------------------------
 * Tsit5 [3P] -- SURVEY App. A.1 / A.2
 * ---------------------------------------------------------------------------------- */
static const double TS_C[7] = { 0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0 };
static const double TS_A[7][7] = {
    { 0 },
    { 0.161 },
    { -0.008480655492356989, 0.335480655492357 },
    { 2.8971530571054935, -6.359448489975075, 4.3622954328695815 },
    { 5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525 },
    { 5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383 },
    { 0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774 },
};
static const double TS_BT[7] = { -0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                                 -0.1447110071732629, 0.5823571654525552, -0.45808210592918697,
                                 0.015151515151515152 };
/* dense output: b_1(Θ) = Θ(r11 + Θ(r12 + Θ(r13 + Θ r14))), b_i(Θ) = Θ²(r_i2 + Θ(r_i3 + Θ r_i4)) */
static const double TS_R[7][4] = {
    { 1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216 },
    { 0.0, 0.13169999999999998, -0.2234, 0.1017 },
    { 0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253 },
    { 0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902 },
    { 0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928 },
    { 0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661 },
    { 0.0, 1.5, -4.0, 2.5 },
};

void orc_tsit5_tableau(double c[7], double a[7][7], double btilde[7], double r[7][4])
{
    memcpy(c, TS_C, sizeof TS_C);
    memcpy(a, TS_A, sizeof TS_A);
    memcpy(btilde, TS_BT, sizeof TS_BT);
    memcpy(r, TS_R, sizeof TS_R);
}

static void interp_weights(double th, double b[7])
{
    b[0] = th * (TS_R[0][0] + th * (TS_R[0][1] + th * (TS_R[0][2] + th * TS_R[0][3])));
    for (int i = 1; i < 7; ++i) b[i] = th * th * (TS_R[i][1] + th * (TS_R[i][2] + th * TS_R[i][3]));
}

/* ode_interpolant (Tsit5 free 4th-order interpolant) [3P] */
static void interpolate(const double u[8], const double k[7][8], double dt, double th, double out[8])
{
    double b[7];
    interp_weights(th, b);
    for (int i = 0; i < 8; ++i) {
        double s = 0.0;
        for (int j = 0; j < 7; ++j) s += b[j] * k[j][i];
        out[i] = u[i] + dt * s;
    }
}

/* ODE_DEFAULT_NORM on an 8-vector: sqrt(sum(abs2)/length) [3P] */
#ifdef ORC_TANGENT_BUILD
/* tangent_oracle.cpp only (this translation unit compiled on a value + 2 tangents scalar): with the switch on, the norms
 * are DiffEqBase's ForwardDiff-extension ones [3P]: |u|_D = sqrt(value² + Σ partials²) for a scalar and
 * sqrt(Σ sse / (N (1 + P))) for an array.  Off (default), and in the plain build, values only. */
#define ORC_ABS(x) (g_norm_with_tangents ? T2(sqrt(orct_sse(x))) : fabs(x))
static double rms8(const double x[8])
{
    if (g_norm_with_tangents) {
        T2 s = 0.0;
        for (int i = 0; i < 8; ++i) s += T2(orct_sse(x[i]));
        return sqrt(s / 24.0);
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += x[i] * x[i];
    return sqrt(s / 8.0);
}
#else
#define ORC_ABS(x) fabs(x)
static double rms8(const double x[8])
{
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += x[i] * x[i];
    return sqrt(s / 8.0);
}
#endif

/* ------------------------------------------------------------------------------------
 * callbacks
 * ---------------------------------------------------------------------------------- */
/* distance_to_disc(::ThinDisc), geometry/discs/thin-disc.jl:20-26 ; _gtol_error discs.jl:7 ;
 * _equatorial_project / _spinaxis_project utils.jl:146-152 */
static double disc_condition(const orc_config* c, const double u[8])
{
    double r = u[1], th = u[2];
    /* distance_to_disc(::DatumPlane), datum-plane.jl:6-10: signed height above the plane */
    if (c->disc_id == ORC_DISC_DATUM) return r * cos(th) - c->disc_params[0];
    if (c->disc_id == ORC_DISC_ELLIPTICAL) {
        /* distance_to_disc(::EllipticalDisc), geometry/discs.jl:57-72 (radial test on r = x4[2], as written there);
         * disc_r_in = inner_radius, disc_params = {semi_major, semi_minor} */
        const double a = c->disc_params[0], b = c->disc_params[1];
        if (a < r || r < c->disc_r_in) return 1.0;
        const double y = sqrt((1.0 - (r / a) * (r / a)) * b * b);
        return fabs(r * cos(th)) - y - c->gtol * fabs(r);
    }
    if (c->disc_id == ORC_DISC_PRECESSING_THIN) {
        /* distance_to_disc(::PrecessingDisc), geometry/discs.jl:74-96 around a ThinDisc: the position is
         * rotated by R = Rx(-β) after shifting ϕ by γ, then handed to the inner disc; disc_params = {β, γ} */
        const double be = c->disc_params[0], ga = c->disc_params[1];
        const double ph = u[3] - ga;
        const double v1 = sin(th) * sin(ph), v2 = sin(th) * cos(ph), v3 = cos(th);
        const double cb = cos(-be), sb = sin(-be);
        /* SMatrix{3,3}(1,0,0, 0,cos(-β),-sin(-β), 0,sin(-β),cos(-β)) is column-major */
        const double x1 = v1, x2 = cb * v2 + sb * v3, x3 = -sb * v2 + cb * v3;
        th = atan2(sqrt(x1 * x1 + x2 * x2), x3);
    }
    const double rho = r * fabs(sin(th));
    if (c->disc_id == ORC_DISC_TABULATED || c->disc_id == ORC_DISC_TORUS) {
        /* ThickDisc(f): cross_section(d, ρ) = f(ρ), thick-disc.jl:57-66 (inner/outer radius 0/Inf) */
        double height;
        if (c->disc_id == ORC_DISC_TORUS) {
            /* _thick_disc, test/smoke-tests/rendergeodesics.jl:7-14.  disc_params[3] selects the LEGACY thick-disc
             * semantics the smoke test's recorded fingerprints were computed with (tests/test_thick_disc_independent.py):
             * bit 0 = subtract gtol |r| like the thin disc, bit 1 = cross section at the spherical radius u[2]
             * (the convention thick-disc.jl:16-27's docstring still shows).  0 = the reference as it is today. */
            const int legacy = (int)c->disc_params[3];
            const double ctr = c->disc_params[0], rad = c->disc_params[1];
            const double q = (legacy & 2) ? r : rho;
            if (q < ctr - rad || q > ctr + rad) height = -1.0;
            else { const double xx = (q - ctr) / rad; height = rad * sqrt(1.0 - xx * xx); }
            if (height <= 0.0) return 1.0;
            return r * fabs(cos(th)) - height - ((legacy & 1) ? c->gtol * fabs(r) : 0.0);
        } else {
            const double r0 = c->disc_params[0], r1 = c->disc_params[1];
            const int64_t n = c->disc_table_n;
            if (rho < r0 || rho > r1) height = -1.0;
            else {
                const double uu = (rho - r0) * ((double)(n - 1) / (r1 - r0));
                int64_t k = (int64_t)uu;
                if (k > n - 2) k = n - 2;
                const double w = uu - (double)k;
                height = (1.0 - w) * c->disc_table[k] + w * c->disc_table[k + 1];
                /* distance_to_disc(::WarpedThinDisc), thin-disc.jl:55-66: the table is the signed height h(ρ) */
                if (c->disc_id == ORC_DISC_TABULATED && c->disc_params[3] != 0.0) {
                    if (rho < c->disc_r_in || rho > c->disc_r_out) return 1.0;
                    return fabs(height - r * cos(th)) - c->gtol * fabs(r);
                }
            }
        }
        if (height <= 0.0) return 1.0;
        return r * fabs(cos(th)) - height;
    }
    if (c->disc_id == ORC_DISC_SHAKURA_SUNYAEV) {
        /* cross_section(::ShakuraSunyaev), geometry/discs/shakura-sunyaev.jl:28-33 ;
         * distance_to_disc(::AbstractThickAccretionDisc), geometry/discs/thick-disc.jl:60-66 */
        double height;
        if (rho < c->disc_r_in) height = -0.0;
        else height = 3.0 * c->disc_params[1] * c->disc_params[0] * (1.0 - sqrt(c->disc_r_in / rho));
        if (height <= 0.0) return 1.0;
        /* disc_params[3] bit 0: legacy semantics (- gtol |r|), see ORC_DISC_TORUS above */
        return r * fabs(cos(th)) - height - (((int)c->disc_params[3] & 1) ? c->gtol * fabs(r) : 0.0);
    }
    if (rho < c->disc_r_in || rho > c->disc_r_out) return 1.0;
    return r * fabs(cos(th)) - c->gtol * fabs(r);
}

/* condition k of a CompositeGeometry: the distance_to_disc of its k-th geometry (bootstrap.jl:86-99 maps
 * intersection_callbacks over cg.geometry, all with the same gtol) */
static double component_condition(const orc_config* c, int k, const double u[8])
{
    orc_config one = *c;
    one.disc_id = c->comp[k].disc_id;
    one.disc_r_in = c->comp[k].disc_r_in;
    one.disc_r_out = c->comp[k].disc_r_out;
    memcpy(one.disc_params, c->comp[k].disc_params, sizeof one.disc_params);
    return disc_condition(&one, u);
}

static int sgn(double x) { return (x > 0.0) - (x < 0.0); }

/* ---- MeshAccretionGeometry (geometry/meshes.jl) ------------------------------------------------------------ */
/* to_cartesian, geometry/geometry.jl:1-3,13-16 */
static void to_cartesian3(const double u[8], double q[3])
{
    const double sth = sin(u[2]);
    q[0] = u[1] * sth * cos(u[3]);
    q[1] = u[1] * sth * sin(u[3]);
    q[2] = u[1] * cos(u[2]);
}
static void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
/* jsf_algorithm, geometry/intersections.jl:58-101 (Jiménez, Segura & Feito 2010), ϵ = 1e-8.  Returns whether the segment
 * Q1 -> Q2 meets the triangle; *tpar = w / (s - w) as there (the callers on this path use only the boolean). */
int orc_jsf(const double V1[3], const double V2[3], const double V3[3], const double Q1[3], const double Q2[3], double* tpar)
{
    const double eps = 1e-8;
    double A[3], B[3], C[3], D[3], W1[3], W2[3], s, t, u;
    for (int i = 0; i < 3; ++i) { A[i] = Q1[i] - V3[i]; B[i] = V1[i] - V3[i]; C[i] = V2[i] - V3[i]; }
    cross3(B, C, W1);
    const double w = dot3(A, W1);
    if (tpar) *tpar = 0.0;
    if (w > eps) {
        for (int i = 0; i < 3; ++i) D[i] = Q2[i] - V3[i];
        s = dot3(D, W1);
        if (s > eps) return 0;
        cross3(A, D, W2);
        t = dot3(W2, C);
        if (t < -eps) return 0;
        u = -dot3(W2, B);
        if (u < -eps) return 0;
        if (w < s + t + u) return 0;
    } else if (w < -eps) {
        return 0;
    } else {
        for (int i = 0; i < 3; ++i) D[i] = Q2[i] - V3[i];
        s = dot3(D, W1);
        if (s > eps) {
            return 0;
        } else if (s < -eps) {
            cross3(D, A, W2);
            t = dot3(W2, C);
            if (t > eps) return 0;
            u = -dot3(W2, B);
            if (u > eps) return 0;
            if (-s > t + u) return 0;
        } else {
            return 0;
        }
    }
    if (tpar) *tpar = w / (s - w);
    return 1;
}
/* intersects_geometry (intersections.jl:7-16) = in_nearby_region (meshes.jl:46-51: the step's end strictly inside the bounding
 * box) && has_intersect (meshes.jl:53-64: the triangles whose first vertex is within 3 of the step's end, in mesh order) on
 * cartesian_line_element (geometry.jl:38-40) = (to_cartesian(uprev), to_cartesian(u)) */
static int mesh_intersects(const orc_config* c, const double uprev[8], const double u[8])
{
    const double* tb = c->disc_table;
    double Q1[3], Q2[3];
    to_cartesian3(uprev, Q1);
    to_cartesian3(u, Q2);
    if (!(tb[0] < Q2[0] && Q2[0] < tb[1] && tb[2] < Q2[1] && Q2[1] < tb[3] && tb[4] < Q2[2] && Q2[2] < tb[5])) return 0;
    for (int64_t k = 0; k < c->disc_table_n; ++k) {
        const double* T = tb + 6 + 9 * k;
        const double dx = T[0] - Q2[0], dy = T[1] - Q2[1], dz = T[2] - Q2[2];
        if (dx * dx + dy * dy + dz * dz < 9.0 && orc_jsf(T, T + 3, T + 6, Q1, Q2, NULL)) return 1;
    }
    return 0;
}

/* DiscreteCallbacks, in CallbackSet order: user callbacks (domain_upper_hemisphere,
 * callbacks.jl:31-40) then the chart (charts.jl:9-23); every callback whose condition holds
 * has its affect! applied, so a later one overrides the status of an earlier one [3P]. */
static int discrete_callbacks(const orc_config* c, const double u[8], int* status)
{
    int term = 0;
    if (c->upper_hemisphere) {
        if (u[1] * cos(u[2]) < c->hemi_delta) { *status = ORC_OUT_OF_DOMAIN; term = 1; }
    }
    double rmin = c->r_inner;
    if (c->chart_table_n > 1) {
        /* PoloidalShapeChart: rmin = shapefunc(θ), a linear interpolation over the tabulated
         * horizon shape (charts.jl:31-48, 61-70; DataInterpolations.LinearInterpolation [3P]) */
        const int64_t n = c->chart_table_n;
        const double dth = (c->chart_theta1 - c->chart_theta0) / (double)(n - 1);
        double f = (u[2] - c->chart_theta0) / dth;
        int64_t k = (int64_t)floor(f);
        if (k < 0) k = 0;
        if (k > n - 2) k = n - 2;
        const double w = f - (double)k;
        rmin = c->chart_table[k] + w * (c->chart_table[k + 1] - c->chart_table[k]);
    }
    if (u[1] <= rmin || u[1] > c->r_outer) {
        *status = (u[1] <= rmin) ? ORC_WITHIN_INNER_BOUNDARY : ORC_OUT_OF_DOMAIN;
        term = 1;
    }
    return term;
}

/* ------------------------------------------------------------------------------------
 * One ray: SciMLBase.init/reinit!/solve! with Tsit5, abstol/reltol, PI controller,
 * Hairer initial dt, CallbackSet(ContinuousCallback(disc), DiscreteCallback(chart)).
 * tracing.jl:198-252 + SURVEY App. A [3P]
 * ---------------------------------------------------------------------------------- */
typedef struct {
    double *r, *vt, *vr, *vphi;
    int64_t cap, n;
    double *t;      /* optional: affine time of every saved state */
} save_t;

static void save_state(save_t* s, const double u[8], double t)
{
    if (!s || s->n >= s->cap) return;
    s->r[s->n] = u[1]; s->vt[s->n] = u[4]; s->vr[s->n] = u[5]; s->vphi[s->n] = u[7];
    if (s->t) s->t[s->n] = t;
    s->n++;
}

static void integrate(const orc_config* c, const double u0[8], orc_point* out, orc_raystats* st, save_t* save)
{
    int winding = 0;
    const double abstol = c->abstol, reltol = c->reltol;
    const double t0 = c->lambda0, tend = c->lambda1;
    const double dtmax = fabs(tend - t0);
    const double beta1 = 7.0 / 50.0, beta2 = 2.0 / 25.0, gam = 0.9, qmin = 0.2, qmax = 10.0;
    const double qoldinit = 1e-4;

    double u[8], k[7][8], unew[8], tmp[8];
    memcpy(u, u0, sizeof u);
    int status = ORC_NO_STATUS, flags = 0;
    int n_acc = 0, n_rej = 0, n_rhs = 0, n_cond = 0;
    double t = t0;
    save_state(save, u, t);

    /* ---- initial dt: ode_determine_initdt (App. A.4) ---- */
    double f0[8], sk[8];
    rhs(c, u, f0); n_rhs++;
    for (int i = 0; i < 8; ++i) sk[i] = abstol + ORC_ABS(u[i]) * reltol;
    for (int i = 0; i < 8; ++i) tmp[i] = u[i] / sk[i];
    const double d0 = rms8(tmp);
    for (int i = 0; i < 8; ++i) tmp[i] = f0[i] / sk[i];
    const double d1 = rms8(tmp);
    double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    dt0 = fmin(dt0, dtmax);
    double dt;
    if (dt0 < 10.0 * DBL_EPSILON) {
        dt = 1e-6;
    } else {
        double u1[8], f1[8];
        for (int i = 0; i < 8; ++i) u1[i] = u[i] + dt0 * f0[i];
        rhs(c, u1, f1); n_rhs++;
        for (int i = 0; i < 8; ++i) tmp[i] = (f1[i] - f0[i]) / sk[i];
        const double d2_ = rms8(tmp) / dt0;
        const double dm = fmax(d1, d2_);
        double dt1;
        if (dm <= 1e-15) dt1 = fmax(1e-6, dt0 * 1e-3);
        else dt1 = pow(10.0, -(2.0 + log10(dm)) / 5.0);
        dt = fmin(fmin(100.0 * dt0, dt1), dtmax);
    }

    memcpy(k[0], f0, sizeof f0);        /* fsalfirst */
    double qold = qoldinit;
    int64_t iter = 0;
    int terminated = 0;

    while (t < tend && !terminated) {
        if (++iter > c->maxiters) { flags |= ORC_FLAG_MAXITERS; break; }
        /* fix_dt_at_bounds! + modify_dt_for_tstops! */
        dt = fmin(dt, dtmax);
        const double dtmin = 4.0 * DBL_EPSILON * fmax(fabs(t), 1.0);
        if (!(dt == dt)) { flags |= ORC_FLAG_NAN; break; }
        if (dt < dtmin) { flags |= ORC_FLAG_DTMIN; break; }
        dt = fmin(dt, tend - t);

        /* perform_step!(Tsit5ConstantCache) */
        for (int s = 1; s < 7; ++s) {
            for (int i = 0; i < 8; ++i) {
                double acc = 0.0;
                for (int j = 0; j < s; ++j) acc += TS_A[s][j] * k[j][i];
                tmp[i] = u[i] + dt * acc;
            }
            if (s == 6) memcpy(unew, tmp, sizeof tmp);
            rhs(c, tmp, k[s]); n_rhs++;
        }
        double et[8];
        for (int i = 0; i < 8; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 7; ++j) acc += TS_BT[j] * k[j][i];
            const double ut = dt * acc;
            et[i] = ut / (abstol + fmax(ORC_ABS(u[i]), ORC_ABS(unew[i])) * reltol);
        }
        const double EEst = rms8(et);

        /* stepsize_controller!(PIController) (App. A.3) */
        double q, q11 = 1.0;
        if (EEst == 0.0) {
            q = 1.0 / qmax;
        } else {
            q11 = pow(EEst, beta1);
            q = q11 / pow(qold, beta2);
            q = fmax(1.0 / qmax, fmin(1.0 / qmin, q / gam));
        }

        if (EEst <= 1.0) {
            /* accept: step_accept_controller! (qsteady_min = qsteady_max = 1) */
            n_acc++;
            qold = fmax(EEst, qoldinit);
            const double dtnew = dt / q;
            double tnew = t + dt;
            /* fixed_t_for_floatingpoint_error! */
            if (fabs(tnew - tend) < 100.0 * DBL_EPSILON * fmax(fabs(tnew), fabs(tend))) tnew = tend;
            const double dtpropose = fmin(dtmax, dtnew);

            /* ---- handle_callbacks!: continuous first (App. A.5) ---- */
            int event = 0;
            if (c->disc_id == ORC_DISC_COMPOSITE) {
                /* geometry_collision_callback(::CompositeGeometry), geometry/bootstrap.jl:76-110: a VectorContinuousCallback,
                 * component k = distance_to_disc of the k-th geometry, every affect = terminate_with_status!(IntersectedWithGeometry).
                 * [3P] DiffEqBase determine_event_occurance / find_callback_time for VectorContinuousCallback, as published:
                 * prev_sign_k = sign(c_k(u_prev)); a component has an event at the step's end iff prev_sign_k != 0 and
                 * prev_sign_k c_k(u_new) <= 0; unless every component has one, the samples ts = range(tprev, t, length = 8)
                 * are scanned in order with the same test and the first sample with any event decides (its components are the
                 * candidates, it is the bracket's top); each candidate's root is bracketed, the earliest root is the event. */
                const int K = c->comp_n;
                double cp[ORC_COMP_MAX], cn_[ORC_COMP_MAX];
                int psk[ORC_COMP_MAX], mask_end = 0, n_end = 0;
                for (int q = 0; q < K; ++q) {
                    cp[q] = component_condition(c, q, u); cn_[q] = component_condition(c, q, unew); n_cond += 2;
                    psk[q] = sgn(cp[q]);
                    if (psk[q] != 0 && (double)psk[q] * cn_[q] <= 0.0) { mask_end |= 1 << q; n_end++; }
                }
                int mask = 0;
                double th_top = 0.0;
                if (n_end != K) {
                    for (int j = 1; j <= 6 && !mask; ++j) {
                        const double th = (double)j / 7.0;
                        double uj[8];
                        interpolate(u, k, dt, th, uj);
                        for (int q = 0; q < K; ++q) {
                            const double cj = component_condition(c, q, uj); n_cond++;
                            if (psk[q] != 0 && (double)psk[q] * cj <= 0.0) mask |= 1 << q;
                        }
                        if (mask) th_top = th;
                    }
                }
                if (!mask && mask_end) { mask = mask_end; th_top = 1.0; }
                if (mask) {
                    /* every candidate's root with the bracketing of the scalar callback below (absolute time, left-biased),
                     * so that a composite ends bit for bit where the earlier of its components' own events ends */
                    double best = 2.0;
                    const double top_t0 = (th_top == 1.0) ? tnew : t + th_top * (tnew - t);
                    for (int q = 0; q < K; ++q) {
                        if (!((mask >> q) & 1)) continue;
                        double ev_u[8], ctop, theta_q;
                        if (th_top == 1.0) ctop = component_condition(c, q, unew);
                        else { interpolate(u, k, dt, (top_t0 - t) / dt, ev_u); ctop = component_condition(c, q, ev_u); }
                        n_cond++;
                        if (ctop == 0.0) theta_q = (top_t0 - t) / dt;
                        else {
                            double lo = t, hi = top_t0;
                            for (int it = 0; it < 200; ++it) {
                                const double mid = lo + 0.5 * (hi - lo);
                                if (!(mid > lo && mid < hi)) break;
                                interpolate(u, k, dt, (mid - t) / dt, ev_u);
                                const double cm = component_condition(c, q, ev_u); n_cond++;
                                if (sgn(cm) == psk[q]) lo = mid; else hi = mid;
                            }
                            theta_q = (lo - t) / dt;
                        }
                        if (theta_q < best) best = theta_q;
                    }
                    double ev_u[8];
                    if (best >= 1.0 && th_top == 1.0) memcpy(ev_u, unew, sizeof ev_u);
                    else interpolate(u, k, dt, best, ev_u);
                    memcpy(unew, ev_u, sizeof unew);
                    tnew = t + best * dt;
                    status = ORC_INTERSECTED_WITH_GEOMETRY;
                    terminated = 1;
                    event = 1;
                }
            } else if (c->disc_id != ORC_DISC_NONE && c->disc_id != ORC_DISC_MESH) {
                const double cprev = disc_condition(c, u); n_cond++;
                const double cnext = disc_condition(c, unew); n_cond++;
                const int ps = sgn(cprev);
                double th_top = 0.0;
                if (ps != 0 && ps * sgn(cnext) <= 0) {
                    event = 1; th_top = 1.0;
                } else if (ps != 0) {
                    /* ts = range(tprev, t, length = 8); safety check on the interpolant */
                    for (int j = 1; j <= 7; ++j) {
                        const double th = (double)j / 7.0;
                        double uj[8];
                        if (j == 7) memcpy(uj, unew, sizeof uj);
                        else interpolate(u, k, dt, th, uj);
                        const double cj = disc_condition(c, uj); n_cond++;
                        if ((double)ps * cj < 0.0) { event = 1; th_top = th; break; }
                    }
                }
                if (event) {
                    /* bracketing root find on absolute time between tprev and top_t, biased to
                     * the left (last time whose sign still equals prev_sign) */
                    double top_t = (th_top == 1.0) ? tnew : t + th_top * (tnew - t);
                    double ev_u[8];
                    double ctop;
                    if (th_top == 1.0) ctop = disc_condition(c, unew);
                    else { interpolate(u, k, dt, (top_t - t) / dt, ev_u); ctop = disc_condition(c, ev_u); }
                    n_cond++;
                    double theta_ev;
                    if (ctop == 0.0) {
                        theta_ev = (top_t - t) / dt;
                    } else {
                        double lo = t, hi = top_t;
                        for (int it = 0; it < 200; ++it) {
                            const double mid = lo + 0.5 * (hi - lo);
                            if (!(mid > lo && mid < hi)) break;      /* adjacent floats */
                            interpolate(u, k, dt, (mid - t) / dt, ev_u);
                            const double cm = disc_condition(c, ev_u); n_cond++;
                            if (sgn(cm) == ps) lo = mid; else hi = mid;
                        }
                        theta_ev = (lo - t) / dt;
                        top_t = lo;
                    }
                    /* change_t_via_interpolation!: u <- interpolant, t <- event time */
                    if (theta_ev >= 1.0 && th_top == 1.0 && ctop == 0.0) memcpy(ev_u, unew, sizeof ev_u);
                    else interpolate(u, k, dt, theta_ev, ev_u);
                    memcpy(unew, ev_u, sizeof unew);
                    tnew = t + theta_ev * dt;
                    status = ORC_INTERSECTED_WITH_GEOMETRY;   /* terminate_with_status!, callbacks.jl:1-6 */
                    terminated = 1;
                }
            }
            /* ---- then the discrete callbacks, on the (possibly moved) state: the geometry's first (merge_callbacks puts it
             * ahead of the user's and the chart's, callbacks.jl:19-38 with bootstrap.jl:11-21; meshes.jl:67-78) ---- */
            if (c->disc_id == ORC_DISC_MESH && mesh_intersects(c, u, unew)) { status = ORC_INTERSECTED_WITH_GEOMETRY; terminated = 1; }
            if (discrete_callbacks(c, unew, &status)) terminated = 1;
            /* winding_callback of TraceWindings, photon-rings.jl:1-15 (a DiscreteCallback merged into the set) */
            if (c->count_windings && ((winding % 2 == 0) ? unew[2] > c->winding_plane : unew[2] < c->winding_plane)) winding++;

            /* apply_step! */
            memcpy(u, unew, sizeof u);
            memcpy(k[0], k[6], sizeof k[0]);   /* FSAL */
            t = tnew;
            dt = dtpropose;
            save_state(save, u, t);
            /* check_error!: unstable_check = any(isnan, u) */
            int bad = 0;
            for (int i = 0; i < 8; ++i) if (!(u[i] == u[i])) bad = 1;
            if (bad) { flags |= ORC_FLAG_NAN; break; }
        } else {
            /* step_reject_controller! */
            n_rej++;
            dt = dt / fmin(1.0 / qmin, q11 / gam);
        }
    }

    out->status = status;
    out->flags = flags | ((winding > 0xFFFF ? 0xFFFF : winding) << 16);   /* gp.aux.winding in the upper half */
    out->lambda_min = t0;
    out->lambda_max = t;
    memcpy(out->x_init, u0, 4 * sizeof(double));
    memcpy(out->v_init, u0 + 4, 4 * sizeof(double));
    memcpy(out->x, u, 4 * sizeof(double));
    memcpy(out->v, u + 4, 4 * sizeof(double));
    if (st) { st->accepted = n_acc; st->rejected = n_rej; st->rhs_evals = n_rhs; st->cond_evals = n_cond; }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ensemble_solve_tracing_problem(::EnsembleEndpointThreads), tracing.jl:151-196 */
int orc_trace(const orc_config* c, const double* xs, int64_t x_stride, const double* vs, int64_t N,
              orc_point* out, orc_raystats* stats, int nthreads)
{
    if (!c || !xs || !vs || !out) return -1;
    if (nthreads <= 0) nthreads = orc_max_threads();
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (int64_t i = 0; i < N; ++i) {
        double u0[8];
        const double* x = xs + i * x_stride;
        memcpy(u0, x, 4 * sizeof(double));
        memcpy(u0 + 4, vs + 4 * i, 4 * sizeof(double));
        u0[4] = orc_constrain_time(c, x, vs + 4 * i);     /* constrain_all, constraints.jl:14-15 */
        integrate(c, u0, &out[i], stats ? &stats[i] : NULL, NULL);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Circular orbits, circular-orbits.jl
 * ---------------------------------------------------------------------------------- */
/* _Ω_analytic :11-18 (prograde) */
static double omega_analytic(const double dr[5])
{
    const double D = sqrt(dr[4] * dr[4] - dr[0] * dr[3]);
    return -(dr[4] - D) / dr[3];
}
/* ut_uϕ :26-37 */
static void ut_uphi(double Om, const double gi[5], double* ut, double* up)
{
    const double A = -(Om * gi[0] - gi[4]);
    const double B = (Om * gi[4] - gi[3]);
    const double den = B * B * gi[0] + 2.0 * A * B * gi[4] + A * A * gi[3];
    const double d = -(double)sgn(den) * sqrt(1.0 / fabs(den));
    *ut = B * d;
    *up = A * d;
}
/* fourvelocity(m, r::Number) = fourvelocity(m, SVector(r, π/2)) :114-121 */
void orc_circular_fourvelocity(const orc_config* c, double r, double v[4])
{
    double g[5], dr[5], dth[5], gi[5], ut, up;
    orc_metric_jacobian(c, r, M_PI / 2.0, g, dr, dth);
    inverse_metric_components(g, gi);
    ut_uphi(omega_analytic(dr), gi, &ut, &up);
    v[0] = gi[0] * ut + gi[4] * up;   /* vt :58-59 */
    v[1] = 0.0;
    v[2] = 0.0;
    v[3] = gi[4] * ut + gi[3] * up;   /* vϕ :60-61 */
}

/* CircularOrbits.energy(m, r) = -u_t :46-48, with its r-derivative via the second-order jet */
static void energy_jet(const orc_config* c, double r, double* E, double* dE)
{
    j2 g[5];
    j2 rr = { r, 1.0, 0.0 }, th = { M_PI / 2.0, 0.0, 0.0 };
    switch (c->metric_id) {
    case ORC_METRIC_JOHANNSEN: johannsen_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_MORRIS_THORNE: morris_thorne_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_BUMBLEBEE: bumblebee_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_KERR_NEWMAN: kerr_newman_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_JOHANNSEN_PSALTIS: johannsen_psaltis_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_DILATON_AXION: dilaton_axion_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_SPHERICAL: spherical_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_KERR_DARK_MATTER: kerr_dark_matter_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_KERR_REFRACTIVE: kerr_refractive_components_j2(c->params, rr, th, g); break;
    case ORC_METRIC_NOZ: noz_components_j2(c->params, rr, th, g); break;
#ifdef ORC_WITH_TEST_METRIC
    case ORC_METRIC_TEST_BUMP: test_bump_components_j2(c->params, rr, th, g); break;
#endif
    default: kerr_components_j2(c->params, rr, th, g);
    }
    /* first-order duals in r: metric g = (v,d); its r-derivative ∂g = (d,dd) */
    double gv[5], gd[5], pv[5], pd[5];
    for (int i = 0; i < 5; ++i) { gv[i] = g[i].v; gd[i] = g[i].d; pv[i] = g[i].d; pd[i] = g[i].dd; }
    /* Ω and dΩ/dr */
    const double disc = pv[4] * pv[4] - pv[0] * pv[3];
    const double ddisc = 2.0 * pv[4] * pd[4] - pd[0] * pv[3] - pv[0] * pd[3];
    const double sq = sqrt(disc), dsq = 0.5 * ddisc / sq;
    const double num = -(pv[4] - sq), dnum = -(pd[4] - dsq);
    const double Om = num / pv[3], dOm = (dnum - Om * pd[3]) / pv[3];
    /* inverse metric and derivative (block form; equals inverse_metric_components) */
    const double D = gv[0] * gv[3] - gv[4] * gv[4];
    const double dD = gd[0] * gv[3] + gv[0] * gd[3] - 2.0 * gv[4] * gd[4];
    const double itt = gv[3] / D, ditt = (gd[3] - itt * dD) / D;
    const double ipp = gv[0] / D, dipp = (gd[0] - ipp * dD) / D;
    const double itp = -gv[4] / D, ditp = (-gd[4] - itp * dD) / D;
    const double A = -(Om * itt - itp), dA = -(dOm * itt + Om * ditt - ditp);
    const double B = (Om * itp - ipp), dB = (dOm * itp + Om * ditp - dipp);
    const double den = B * B * itt + 2.0 * A * B * itp + A * A * ipp;
    const double dden = 2.0 * B * dB * itt + B * B * ditt + 2.0 * (dA * B * itp + A * dB * itp + A * B * ditp)
                        + 2.0 * A * dA * ipp + A * A * dipp;
    const double s = (double)sgn(den);
    const double ad = fabs(den);
    const double d = -s / sqrt(ad);
    const double dd = 0.5 * s * (s * dden) / (ad * sqrt(ad));
    const double ut = B * d, dut = dB * d + B * dd;
    *E = -ut;
    *dE = -dut;
}

/* __BoyerLindquistFO.isco, kerr-metric-first-order.jl:297-337 */
static double kerr_isco(double M, double a)
{
    const double x = a / M;
    const double Z1 = 1.0 + cbrt(1.0 - x * x) * (cbrt(1.0 + x) + cbrt(1.0 - x));
    const double Z2 = sqrt(3.0 * x * x + Z1 * Z1);
    const double s = sqrt((3.0 - Z1) * (3.0 + Z1 + 2.0 * Z2));
    return (a > 0.0) ? M * (3.0 + Z2 - s) : M * (3.0 + Z2 + s);
}

/* isco(m::AbstractStaticAxisSymmetric), special-radii.jl:14-60: find_isco_bounds then a
 * bracketing root find of dE/dr on (lower, upper) */
static double generic_isco(const orc_config* c)
{
    const double upper = 100.0, step = 0.005;
    double lower = 0.0;
    int found = 0;
    const int64_t nsteps = (int64_t)floor((upper - 1.0) / step + 1e-9);
    for (int64_t i = 0; i <= nsteps; ++i) {
        const double r = upper - step * (double)i;
        double E, dE;
        energy_jet(c, r, &E, &dE);
        if (fabs(E) > 1.0 || !(E == E)) { lower = r; found = 1; break; }
    }
    if (!found) return NAN;
    double lo = lower, hi = upper, Elo, dlo, Ehi, dhi;
    energy_jet(c, lo, &Elo, &dlo);
    energy_jet(c, hi, &Ehi, &dhi);
    if (!(dlo == dlo)) {
        /* E is complex-valued (NaN) at the bound itself; nudge inside */
        lo += step;
        energy_jet(c, lo, &Elo, &dlo);
    }
    if (sgn(dlo) == sgn(dhi)) return NAN;
    for (int it = 0; it < 200; ++it) {
        const double mid = lo + 0.5 * (hi - lo);
        if (!(mid > lo && mid < hi)) break;
        double Em, dm;
        energy_jet(c, mid, &Em, &dm);
        if (sgn(dm) == sgn(dlo)) lo = mid; else hi = mid;
    }
    return 0.5 * (lo + hi);
}

double orc_isco(const orc_config* c)
{
    if (c->metric_id == ORC_METRIC_KERR) return kerr_isco(c->params[0], c->params[1]);
    return generic_isco(c);
}

/* ------------------------------------------------------------------------------------
 * Redshift, redshift.jl
 * ---------------------------------------------------------------------------------- */
static double kerr_Delta(double M, double r, double a) { return r * r - 2.0 * M * r + a * a; }
/* Lₑ :93 */
static double plunge_Le(double M, double rms, double a)
{
    return sqrt(M) * (rms * rms - 2.0 * a * sqrt(M * rms) + a * a)
           / (pow(rms, 1.5) - 2.0 * M * sqrt(rms) + a * sqrt(M));
}
/* H :106 */
static double plunge_H(double M, double rms, double r, double a)
{
    return (2.0 * M * r - a * plunge_Le(M, rms, a)) / kerr_Delta(M, r, a);
}
static double plunge_gamma(double M, double rms) { return sqrt(1.0 - (2.0 * M) / (3.0 * rms)); }  /* γₑ :123 */
static double plunge_ur(double M, double rms, double r)                                            /* uʳ :140 */
{
    return -sqrt((2.0 * M) / (3.0 * rms)) * pow(rms / r - 1.0, 1.5);
}
static double plunge_uphi(double M, double rms, double r, double a)                                /* uᶲ :153 */
{
    return plunge_gamma(M, rms) / (r * r) * (plunge_Le(M, rms, a) + a * plunge_H(M, rms, r, a));
}
static double plunge_ut(double M, double rms, double r, double a)                                  /* uᵗ :164 */
{
    return plunge_gamma(M, rms) * (1.0 + 2.0 * M * (1.0 + plunge_H(M, rms, r, a)) / r);
}

/* NaNLinearInterpolator, interpolations.jl:7-29 */
static double nan_linear_interp(const double* t, const double* y, int64_t n, double x)
{
    /* idx = clamp(searchsortedlast(t, x), 1, n-1), 1-based */
    int64_t lo = 0, hi = n;       /* count of elements <= x */
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (t[m] <= x) lo = m + 1; else hi = m; }
    int64_t idx = lo;             /* searchsortedlast (1-based) */
    if (idx < 1) idx = 1;
    if (idx > n - 1) idx = n - 1;
    const double x1 = t[idx - 1], x2 = t[idx], y1 = y[idx - 1], y2 = y[idx];
    const double w = (x - x1) / (x2 - x1);
    const double v = (1.0 - w) * y1 + w * y2;
    if (!(v == v)) {
        if (w < 0.5) return (y1 == y1) ? y1 : 0.0;
        return (y2 == y2) ? y2 : 0.0;
    }
    return v;
}

/* _redshift_dotproduct, redshift.jl:204-220 */
static double redshift_dot(const orc_config* c, const orc_point* gp, const double v_disc[4])
{
    double g[5], G[4][4], Gobs[4][4];
    metric_components(c, gp->x[1], gp->x[2], g);
    sym_matrix(g, G);
    metric_components(c, gp->x_init[1], gp->x_init[2], g);
    sym_matrix(g, Gobs);
    const double v_obs[4] = { 1.0, 0.0, 0.0, 0.0 };
    const double E_disc = mdot(G, gp->v, v_disc);
    const double E_obs = mdot(Gobs, gp->v_init, v_obs);
    return E_obs / E_disc;
}

static double redshift_pf(const orc_config* c, const orc_pf* pf, const orc_point* gp)
{
    const double rho = gp->x[1] * fabs(sin(gp->x[2]));      /* _equatorial_project */
    double v_disc[4];
    if (c->metric_id == ORC_METRIC_KERR && pf->n_plunge == 0) {
        /* redshift_function(m::KerrMetric, gp), redshift.jl:192-202 */
        const double M = c->params[0], a = c->params[1];
        const double isco = kerr_isco(M, a);
        if (rho < isco) {
            v_disc[0] = plunge_ut(M, isco, rho, a);
            v_disc[1] = -plunge_ur(M, isco, rho);
            v_disc[2] = 0.0;
            v_disc[3] = plunge_uphi(M, isco, rho, a);
        } else {
            orc_circular_fourvelocity(c, rho, v_disc);
        }
    } else {
        /* interpolate_redshift closure, redshift.jl:246-276 */
        if (rho < pf->r_isco) {
            double rb = rho;                                   /* _enforce_interpolation_bounds */
            if (rb < pf->plunge_r[0]) rb = pf->plunge_r[0];
            if (rb > pf->plunge_r[pf->n_plunge - 1]) rb = pf->plunge_r[pf->n_plunge - 1];
            v_disc[0] = nan_linear_interp(pf->plunge_r, pf->plunge_vt, pf->n_plunge, rb);
            v_disc[1] = -nan_linear_interp(pf->plunge_r, pf->plunge_vr, pf->n_plunge, rb);
            v_disc[2] = 0.0;
            v_disc[3] = nan_linear_interp(pf->plunge_r, pf->plunge_vphi, pf->n_plunge, rb);
        } else {
            orc_circular_fourvelocity(c, rho, v_disc);
        }
    }
    return redshift_dot(c, gp, v_disc);
}

/* apply_to_image!, rendering.jl:103-107 ; pf ∘ FilterPointFunction, point-functions.jl:111-127 */
void orc_apply_pf(const orc_config* c, const orc_pf* pf, const orc_point* pts, int64_t N, double max_time,
                  double* out, int nthreads)
{
    if (nthreads <= 0) nthreads = orc_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < N; ++i) {
        const orc_point* gp = &pts[i];
        int pass = 1;
        if (pf->filter_id == ORC_FILTER_EARLY_TERM) pass = gp->lambda_max < max_time;
        else if (pf->filter_id == ORC_FILTER_INTERSECTED) pass = gp->status == ORC_INTERSECTED_WITH_GEOMETRY;
        if (!pass) { out[i] = pf->fill; continue; }
        switch (pf->pf_id) {
        case ORC_PF_AFFINE_TIME: out[i] = gp->lambda_max; break;
        case ORC_PF_REDSHIFT: out[i] = redshift_pf(c, pf, gp); break;
        case ORC_PF_STATUS: out[i] = (double)gp->status; break;
        case ORC_PF_R: out[i] = gp->x[1] * fabs(sin(gp->x[2])); break;
        default: out[i] = NAN;
        }
    }
}

/* plunging_fourvelocity, circular-orbits.jl:128-146 ; interpolate_plunging_velocities,
 * orbit-solving.jl:137-167 */
int64_t orc_plunging_table(const orc_config* c, double r_isco, double* r, double* vt, double* vr,
                           double* vphi, int64_t cap)
{
    orc_config cc = *c;
    const double reltol = 1e-9, dr = reltol * 10.0;
    cc.mu = 1.0;
    cc.reltol = reltol;
    cc.abstol = 1e-9;
    cc.lambda0 = 0.0;
    cc.lambda1 = 50000.0;
    cc.disc_id = ORC_DISC_NONE;
    cc.upper_hemisphere = 0;
    /* chart_for_metric(m; closest_approach = 1.000001): inner_radius(m) = M + sqrt(M^2 - a^2) */
    const double M = c->params[0], a = c->params[1];
    cc.r_inner = (M + sqrt(M * M - a * a)) * 1.000001;
    cc.r_outer = 12000.0;

    double g[5], dg[5], dth[5], gi[5], ut, up;
    orc_metric_jacobian(&cc, r_isco, M_PI / 2.0, g, dg, dth);
    inverse_metric_components(g, gi);
    ut_uphi(omega_analytic(dg), gi, &ut, &up);
    const double E = -ut, L = up;
    const double vtt = gi[0] * ut + gi[4] * up, vpp = gi[4] * ut + gi[3] * up;
    const double nom = gi[0] * E * E - 2.0 * gi[4] * E * L + gi[3] * L * L + 1.0;
    const double den = -g[1];
    double u0[8] = { 0.0, r_isco - dr, M_PI / 2.0, 0.0, vtt, -sqrt(fabs(nom / den)), 0.0, vpp };
    u0[4] = orc_constrain_time(&cc, u0, u0 + 4);
    save_t sv = { r, vt, vr, vphi, cap, 0, NULL };
    orc_point pt;
    integrate(&cc, u0, &pt, NULL, &sv);
    return sv.n;
}

/* debugging aid for the tests: every accepted step (t, r) of one ray */
int64_t orc_trace_steps(const orc_config* c, const double x[4], const double v[4], orc_point* out,
                        double* t, double* r, int64_t cap)
{
    double u0[8];
    memcpy(u0, x, 4 * sizeof(double));
    memcpy(u0 + 4, v, 4 * sizeof(double));
    u0[4] = orc_constrain_time(c, x, v);
    double* scratch = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
    save_t sv = { r, scratch, scratch + cap, scratch + 2 * cap, cap, 0, t };
    integrate(c, u0, out, NULL, &sv);
    free(scratch);
    return sv.n;
}


// flopcount.cpp -- ORACLE tooling (test/bench infrastructure only).
//
// Compiles the oracle (gradus_oracle.c, unchanged) with `double` replaced by a counting scalar, and
// prints the exact number of floating-point operations the reference formulation of the render
// path performs ("exact count from the CPU restatement instantiated on a
// counting scalar type").  + - * / sqrt and each transcendental count 1, fma counts 2,
// comparisons / abs / negation / min / max count 0.
//
//   make -C oracle count   ->   oracle/flopcount.json
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static long long g_add, g_mul, g_div, g_sqrt, g_trans, g_fma;

struct Counted {
    double v;
    Counted() : v(0.0) {}
    Counted(double x) : v(x) {}
    Counted(long double x) : v((double)x) {}
    Counted(int x) : v((double)x) {}
    Counted(long x) : v((double)x) {}
    Counted(long long x) : v((double)x) {}
    template <class T> explicit operator T() const { return (T)v; }
};
static inline Counted operator+(Counted a, Counted b) { ++g_add; return a.v + b.v; }
static inline Counted operator-(Counted a, Counted b) { ++g_add; return a.v - b.v; }
static inline Counted operator*(Counted a, Counted b) { ++g_mul; return a.v * b.v; }
static inline Counted operator/(Counted a, Counted b) { ++g_div; return a.v / b.v; }
static inline Counted operator-(Counted a) { return -a.v; }
static inline Counted& operator+=(Counted& a, Counted b) { ++g_add; a.v += b.v; return a; }
static inline Counted& operator-=(Counted& a, Counted b) { ++g_add; a.v -= b.v; return a; }
#define CMP(op) static inline bool operator op(Counted a, Counted b) { return a.v op b.v; }
CMP(<) CMP(>) CMP(<=) CMP(>=) CMP(==) CMP(!=)
#undef CMP
static inline Counted sqrt(Counted a) { ++g_sqrt; return std::sqrt(a.v); }
static inline Counted cbrt(Counted a) { ++g_trans; return std::cbrt(a.v); }
static inline Counted sin(Counted a) { ++g_trans; return std::sin(a.v); }
static inline Counted atan(Counted a) { ++g_trans; return std::atan(a.v); }
static inline Counted atan2(Counted a, Counted b) { ++g_trans; return std::atan2(a.v, b.v); }
static inline Counted cos(Counted a) { ++g_trans; return std::cos(a.v); }
static inline Counted pow(Counted a, Counted b) { ++g_trans; return std::pow(a.v, b.v); }
static inline Counted log10(Counted a) { ++g_trans; return std::log10(a.v); }
static inline Counted fabs(Counted a) { return std::fabs(a.v); }
static inline Counted floor(Counted a) { return std::floor(a.v); }
static inline Counted fmax(Counted a, Counted b) { return std::fmax(a.v, b.v); }
static inline Counted fmin(Counted a, Counted b) { return std::fmin(a.v, b.v); }
static inline Counted fma(Counted a, Counted b, Counted c) { ++g_fma; return std::fma(a.v, b.v, c.v); }

#ifdef _OPENMP
#include <omp.h>
#endif
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define double Counted
#include "gradus_oracle.c"
#undef double

static long long total() { return g_add + g_mul + g_div + g_sqrt + g_trans + 2 * g_fma; }
static void reset() { g_add = g_mul = g_div = g_sqrt = g_trans = g_fma = 0; }

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 96;   // S x S sample of the bench image plane
    orc_config c;
    memset((void*)&c, 0, sizeof c);
    c.metric_id = ORC_METRIC_KERR;
    c.disc_id = ORC_DISC_THIN;
    c.params[0] = 1.0; c.params[1] = 0.998;
    const double rp = 1.0 + std::sqrt(1.0 - 0.998 * 0.998);
    c.r_inner = rp * 1.01; c.r_outer = 12000.0;
    c.disc_r_in = 1.2369706551751847; c.disc_r_out = 50.0; c.gtol = 1e-2;
    c.lambda0 = 0.0; c.lambda1 = 2000.0; c.abstol = 1e-9; c.reltol = 1e-9; c.mu = 0.0;
    c.maxiters = 1000000;
    Counted x[4] = { 0.0, 1000.0, 75.0 * M_PI / 180.0, 0.0 };
    const long N = (long)S * S;
    std::vector<Counted> v(4 * N);
    orc_render_velocities(&c, x, -60.0, 60.0, -35.0, 35.0, S, S, 0, N, v.data());
    std::vector<orc_point> pts(N);
    std::vector<orc_raystats> st(N);

    // (1) whole trace
    reset();
    orc_trace(&c, x, 0, v.data(), N, pts.data(), st.data(), 1);
    const long long trace_flops = total();
    const long long a = g_add, m = g_mul, d = g_div, s = g_sqrt, t = g_trans, f = g_fma;
    long long steps = 0, rhs_evals = 0, conds = 0;
    for (long i = 0; i < N; ++i) { steps += st[i].accepted + st[i].rejected; rhs_evals += st[i].rhs_evals; conds += st[i].cond_evals; }

    // (2) one RHS evaluation
    reset();
    Counted acc[4];
    orc_geodesic_equation(&c, pts[0].x_init, pts[0].v_init, acc);
    const long long rhs_flops = total();

    // (3) point function
    reset();
    orc_pf pf;
    memset((void*)&pf, 0, sizeof pf);
    pf.pf_id = ORC_PF_REDSHIFT; pf.filter_id = ORC_FILTER_INTERSECTED; pf.fill = NAN; pf.r_isco = c.disc_r_in;
    std::vector<Counted> out(N);
    orc_apply_pf(&c, &pf, pts.data(), N, 2000.0, out.data(), 1);
    const long long pf_flops = total();

    // model used by bench.py: flops scale with attempted steps (6 RHS + stage sums + error norm +
    // controller + 9 condition evaluations on the dense output per step); the per-ray constant
    // part (initial dt, constraint, point function) is folded into the per-step figure.
    const double ps = ((double)trace_flops + (double)pf_flops) / (double)steps;
    const double fixed_total = 0.0;

    printf("{\n");
    printf("  \"workload\": \"Kerr a=0.998, r_obs=1000, theta=75deg, ThinDisc(isco,50), %dx%d sample of the bench image plane\",\n", S, S);
    printf("  \"rays\": %ld,\n  \"attempted_steps\": %lld,\n  \"rhs_evals\": %lld,\n  \"cond_evals\": %lld,\n", N, steps, rhs_evals, conds);
    printf("  \"trace_flops\": %lld,\n  \"add\": %lld, \"mul\": %lld, \"div\": %lld, \"sqrt\": %lld, \"transcendental\": %lld, \"fma\": %lld,\n", trace_flops, a, m, d, s, t, f);
    printf("  \"flops_per_rhs\": %lld,\n  \"pointfunction_flops_per_ray\": %.1f,\n", rhs_flops, (double)pf_flops / (double)N);
    printf("  \"flops_per_ray_mean\": %.1f,\n", ((double)trace_flops + (double)pf_flops) / (double)N);
    printf("  \"flops_per_step\": %.2f,\n  \"flops_per_ray_fixed\": %.1f\n}\n", ps, fixed_total);
    return 0;
}

// gr_tabmetric.hpp -- GR_METRIC_TABULATED: a user-defined AbstractStaticAxisSymmetric metric on the device.
//
// The reference's plugin contract for a metric is one method, metric_components(m, (r, θ)) -> (g_tt, g_rr, g_θθ, g_ϕϕ, g_tϕ)
// (src/metrics/kerr-metric.jl:62-70; src/Gradus.jl:78-86), ForwardDiff supplies the Jacobian the geodesic equation needs
// (src/tracing/method-implementations/auto-diff.jl:206-211).  A closure cannot cross the C ABI, so the HOST samples it and the
// device evaluates the five components AND their (∂r, ∂θ) derivatives from piecewise polynomials:
//
//   * radial patches: the m_r equal parts of every octave [2^e, 2^(e+1)) of x = r - r0, e = e_min .. e_min + n_oct - 1.  With
//     r0 just inside the horizon the patches shrink geometrically towards it (where g_rr has its pole) and grow geometrically
//     outwards (where the metric flattens), and the patch of a radius comes out of the exponent and mantissa bits of x: no
//     logarithm, no search, no reciprocal (du/dr = 2 m_r 2^-e is an ldexp);
//   * polar patches: n_theta equal parts of [0, π]; θ outside is folded (components are even about both poles, ∂θ is odd);
//   * on a patch, in local coordinates u, v ∈ [-1, 1], every component is ONE polynomial of TOTAL degree kDegree,
//         g_k(u, v) = Σ_{i + j <= p} c_kij u^i v^j,
//     evaluated together with ∂u and ∂v by nested Horner recurrences (68 FMAs per component at p = 7 against 152 for the tensor
//     degree; the terms of total degree > p that a tensor-degree fit would add are below the truncation error anyway).
//     Coefficients are stored in exactly the order the recurrences consume them (rows i = p .. 0, j = p - i .. 0 inside a row).
//
// Table = kHeaderDoubles header doubles + n_patches x kPatchDoubles; patch (ir, it) at index ir * n_theta + it.
// The fit (metric_table.hip, host only): N x N Chebyshev nodes per patch -> Chebyshev coefficients -> truncation to total
// degree p -> monomials.  The dropped coefficients are the error estimate the caller refines the grid against.
#pragma once

#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define GR_TAB_HD __host__ __device__
#else
#define GR_TAB_HD
#endif

namespace gr_tab {

constexpr int kDegree = 7;
constexpr int kCoefs = (kDegree + 1) * (kDegree + 2) / 2;      // 36 per component
constexpr int kComps = 5;
constexpr int kPatchDoubles = 184;                              // 5 x 36 = 180, padded to 23 x 64 bytes
constexpr int kHeaderDoubles = 16;
constexpr int kFitNodes = 12;                                   // Chebyshev nodes per patch and direction sampled by the fit
constexpr double kMagic = 1196576084.0;                         // 'GRMT'

// header slots
enum { H_MAGIC = 0, H_DEGREE, H_R0, H_EMIN, H_NOCT, H_MR, H_NTHETA, H_STRIDE, H_BUILD_ID, H_ERR_VAL, H_ERR_DR, H_ERR_DTH,
       H_RMIN, H_RMAX, H_POLE_FACTOR, H_RES1 };

// offset of row i inside a component's block (rows are stored i = p, p-1, ..., 0; row i has p - i + 1 coefficients)
constexpr int row_offset(int i) { return (kDegree - i) * (kDegree - i + 1) / 2; }

// P, ∂u P, ∂v P of the five components from one patch.  coef(k) returns coefficient k of the patch (k < 180) as a double.
// Every coefficient is used where it arrives and nowhere else (a row's leading coefficient enters as c0 v, which both the row's
// value and its derivative start from): on the device a coefficient is half of a 16-byte register tuple fresh from LDS -- or
// one scalar register pair, of which an instruction takes one -- and a coefficient needed again later would keep its whole tuple
// alive.  The arithmetic is handed over as operations that involve at most one coefficient k:
//     fma(a, b, c) = a b + c      add(a, b) = a + b      fmak(a, b, k) = a b + k      mulk(a, k) = a k      addk(a, k) = a + k.
// 77 operations per component at p = 7.  Host and device evaluate the same operations in the same order.
// op.row_done(k, i, P, Pu, Pv) is called when row i of component k has been folded into the three accumulators.
template <class T, class OPS, class C>
GR_TAB_HD __attribute__((always_inline)) inline void eval_patch(const C& coef, const OPS& op, T u, T v, T P[kComps], T Pu[kComps], T Pv[kComps])
{
    static_assert(kDegree >= 3, "the recurrences below special-case the two highest rows");
#pragma unroll
    for (int k = 0; k < kComps; ++k) {
        const int base = k * kCoefs;
        // rows p and p - 1:  P = c_p0 u + (c_(p-1)1 v + c_(p-1)0);  ∂u P = c_p0 and ∂v P = c_(p-1)1 enter the next row as products
        const T tcp = op.mulk(u, coef(base));                 // c_p0 u
        const double c10 = coef(base + 1);
        const T t10v = op.mulk(v, c10), t10u = op.mulk(u, c10);
        T p_ = op.add(tcp, op.addk(t10v, coef(base + 2)));
        T pu, pv;
#pragma unroll
        for (int i = kDegree - 2; i >= 0; --i) {
            const int n = kDegree - i, off = base + row_offset(i);
            // q(v) = Σ_j c_ij v^j and q'(v) by Horner from the row's leading coefficient
            const T t0 = op.mulk(v, coef(off));                // c0 v
            T q = op.addk(t0, coef(off + 1));                  // c0 v + c1
            T dq = op.add(t0, q);                              // 2 c0 v + c1
            q = op.fmak(q, v, coef(off + 2));
#pragma unroll
            for (int t = 3; t <= n; ++t) {
                dq = op.fma(dq, v, q);
                q = op.fmak(q, v, coef(off + t));
            }
            if (i == kDegree - 2) {
                pu = op.add(tcp, p_);
                pv = op.add(t10u, dq);
            } else {
                pu = op.fma(pu, u, p_);
                pv = op.fma(pv, u, dq);
            }
            p_ = op.fma(p_, u, q);
            op.row_done(k, i, p_, pu, pv);      // (a hook for the device's coefficient stream; nothing on the host)
        }
        P[k] = p_; Pu[k] = pu; Pv[k] = pv;
    }
}

// the operations on plain doubles (host, and the device's per-lane loads from global memory)
struct HostOps {
    static GR_TAB_HD double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static GR_TAB_HD double add(double a, double b) { return a + b; }
    static GR_TAB_HD double fmak(double a, double b, double k) { return __builtin_fma(a, b, k); }
    static GR_TAB_HD double mulk(double a, double k) { return a * k; }
    static GR_TAB_HD double addk(double a, double k) { return a + k; }
    static GR_TAB_HD void row_done(int, int, double&, double&, double&) {}
};

// g_ϕϕ and g_tϕ from their stored forms h = g / sin²θ (gr_metric_grid.pole_factor): g = w h, ∂r g = w ∂r h, ∂θ g = w' h + w ∂θ h
// with w = sin²θ, w' = 2 sinθ cosθ at the ACTUAL θ (w is even about both poles like the stored h, so the fold changes nothing)
template <class T>
GR_TAB_HD __attribute__((always_inline)) inline void pole_factor_apply(T w, T dw, T g[kComps], T dr[kComps], T dth[kComps])
{
#pragma unroll
    for (int k = 3; k < kComps; ++k) {
        dth[k] = dw * g[k] + w * dth[k];
        g[k] = w * g[k];
        dr[k] = w * dr[k];
    }
}

// ---- where (r, θ) lies in the grid ----

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline int tab_ilogb(double x) { return __builtin_amdgcn_frexp_exp(x) - 1; }
__device__ inline double tab_ldexp(double x, int e) { return __builtin_amdgcn_ldexp(x, e); }
__device__ inline double tab_rint(double x) { return __builtin_rint(x); }
__device__ inline double tab_fabs(double x) { return __builtin_fabs(x); }
#else
inline int tab_ilogb(double x) { int e; (void)__builtin_frexp(x, &e); return e - 1; }
inline double tab_ldexp(double x, int e) { return __builtin_ldexp(x, e); }
inline double tab_rint(double x) { return __builtin_rint(x); }
inline double tab_fabs(double x) { return __builtin_fabs(x); }
#endif

// The grid as the kernels carry it: everything locate() needs, already in the form it needs it (the doubles formed once on the
// host: on the device a value derived from an integer by a conversion lives in VECTOR registers for the whole step loop).
struct GridK {
    double r0;
    double xmin;              // 2^e_min
    double mr;                // m_r
    double nth_over_pi;       // n_theta / π
    int32_t e_min, e_max;     // first and last octave
    int32_t m_r, n_theta;
};
GR_TAB_HD __attribute__((always_inline)) inline GridK make_gridk(double r0, int e_min, int n_oct, int m_r, int n_theta)
{
    GridK k;
    k.r0 = r0;
    k.xmin = tab_ldexp(1.0, e_min);
    k.mr = (double)m_r;
    k.nth_over_pi = (double)n_theta * (1.0 / 3.14159265358979323846);
    k.e_min = e_min;
    k.e_max = e_min + n_oct - 1;
    k.m_r = m_r;
    k.n_theta = n_theta;
    return k;
}

// patch index, local coordinates and the chain-rule factors du/dr, dv/dθ (the latter carries the sign of the fold).
// A radius outside the table's octaves takes the nearest patch (below the first octave: its inner edge; beyond the last: the
// polynomial extrapolates -- stage points of a step that overshoots the chart by a hair; the chart callbacks end such a ray at
// the step's end).
GR_TAB_HD __attribute__((always_inline)) inline void locate(const GridK& k, double r, double th, int& patch, double& u, double& v,
                                                              double& su, double& sv)
{
    constexpr double kTwoPi = 6.28318530717958647692, kInvTwoPi = 0.15915494309189533577;
    // θ -> [0, π]: even about both poles
    const double q = tab_rint(th * kInvTwoPi);
    const double w = __builtin_fma(-q, kTwoPi, th);
    const double a = tab_fabs(w);
    const double y = a * k.nth_over_pi;
    int it = (int)y;
    it = it > k.n_theta - 1 ? k.n_theta - 1 : it;
    v = __builtin_fma(2.0, y - (double)it, -1.0);
    const double sv_mag = 2.0 * k.nth_over_pi;
    sv = w < 0.0 ? -sv_mag : sv_mag;
    // r -> octave e of x = r - r0, part j of the octave
    double x = r - k.r0;
    x = x > k.xmin ? x : k.xmin;         // (also catches NaN and r <= r0)
    int e = tab_ilogb(x);
    e = e > k.e_max ? k.e_max : e;
    const double f = tab_ldexp(x, -e);                 // [1, 2) unless clamped above
    const double z = (f - 1.0) * k.mr;
    int j = (int)z;
    j = j > k.m_r - 1 ? k.m_r - 1 : j;
    u = __builtin_fma(2.0, z - (double)j, -1.0);
    su = tab_ldexp(2.0 * k.mr, -e);
    patch = ((e - k.e_min) * k.m_r + j) * k.n_theta + it;
}

}  // namespace gr_tab

// gr_tabmetric.hpp -- GR_METRIC_TABULATED: a user-defined AbstractStaticAxisSymmetric metric on the device.
//
// The reference's plugin contract for a metric is one method, metric_components(m, (r, θ)) -> (g_tt, g_rr, g_θθ, g_ϕϕ, g_tϕ)
// (src/metrics/kerr-metric.jl:62-70; src/Gradus.jl:78-86), ForwardDiff supplies the Jacobian the geodesic equation needs
// (src/tracing/method-implementations/auto-diff.jl:206-211).  A closure cannot cross the C ABI, so the HOST samples it and the
// device evaluates the five components AND their (∂r, ∂θ) derivatives from piecewise polynomials:
//
//   * radial SEGMENTS [r_lo, r_hi): each an octave grid of x = ±(r - anchor) > 0 -- the m_r equal parts of every octave
//     [2^e, 2^(e+1)), e = e_lo .. e_hi, optionally preceded by a CORE of m_r equal parts of [0, 2^e_lo).  The patch of a radius
//     comes out of the exponent and mantissa bits of x: no logarithm, no search, no reciprocal (du/dr = ±2 m_r 2^-e is an ldexp).
//       - segment 0 is anchored just inside the horizon: its patches shrink geometrically towards the pole of g_rr and grow
//         geometrically outwards, where the metric flattens.  A smooth metric has this one segment, and the kernels a fast path;
//       - a metric that is PIECEWISE in r (src/metrics/kerr-dark-matter.jl:12-20, utils.jl:158-168 in kerr-refractive-ad.jl) names
//         its break radii (gr_metric_grid_plan_breaks): a segment starts at every break, anchored there, so no patch straddles
//         one ("hard" segment ends: the rows at such an end are fitted on their part inside the segment only); a break with a
//         SCALE -- a feature that narrow centred there -- gets geometric patches on both sides down to that scale;
//       - r may be negative (a chart through the throat of a wormhole, src/metrics/morris-thorne-ad.jl): x is a distance from an
//         anchor, not the radius;
//   * polar patches: n_theta equal parts of [0, π]; θ outside is folded (components are even about both poles, ∂θ is odd);
//   * on a patch, in local coordinates u, v ∈ [-1, 1], every component is ONE polynomial of TOTAL degree kDegree,
//         g_k(u, v) = Σ_{i + j <= p} c_kij u^i v^j,
//     evaluated together with ∂u and ∂v by nested Horner recurrences ((p + 1)(p + 2) + 5 operations per component; the terms of
//     total degree > p that a tensor-degree fit would add are below the truncation error anyway).
//     Coefficients are stored in exactly the order the recurrences consume them (rows i = p .. 0, j = p - i .. 0 inside a row);
//   * the azimuthal components g_ϕϕ, g_tϕ in one of three forms (H_POLE_FACTOR):
//         1  h = g / sin²θ               both vanish like sin²θ on the axis of a regular metric: a polynomial with an absolute
//                                        error would leave g^ϕϕ with an unbounded relative one for rays that graze the axis;
//         2  g = K(r, θ) + sin²θ h,  K = K_m(r) + K_d(r) cos θ      a metric whose g_ϕϕ, g_tϕ do NOT vanish on the axis (an axion
//                                        charge, src/metrics/dilaton-axion-ad.jl:13-14: W ∝ csc²θ): K_m ± K_d are the limits on the two
//                                        poles, one polynomial in u per radial row each; h is smooth again and g keeps its relative
//                                        accuracy next to the zero it then has a few milliradians off the axis;
//         0  as sampled                  (the reference's MorrisThorneWormhole: g_ϕϕ ∝ sin θ).
//
// Table = kHeaderDoubles header doubles | kMaxSeg segment records | [form 2: n_rows x kAxisDoubles] | n_rows x n_theta patches of
// kPatchDoubles; patch (row, it) at index row * n_theta + it.
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

// Degree 5 against degree 7, the round-5 choice, at EQUAL fit error (profiles/r6_tab_degree_ab.log; C2, 1024² rays): degree 5 on a
// (24, 96) grid 17.8 ms, degree 6 on (12, 48) 18.9, degree 7 on (8, 32) 21.5 -- 210 operations per evaluation instead of 385, and
// the finer grid's extra patch misses cost less than the higher degree's arithmetic.  (Degree 5 on (16, 64): 16.9 ms with the
// image 1e-10 from the fused kernel's instead of 1e-11 -- but jumps of 1e-7 in the derivatives at patch edges, which a tolerance of
// 1e-11 or a difference quotient of traces resolves: not the default.)
#ifndef GR_TAB_DEGREE
#define GR_TAB_DEGREE 5
#endif
constexpr int kDegree = GR_TAB_DEGREE;                          // total degree of a patch polynomial (a build-time choice: 5, 6 or 7)
constexpr int kCoefs = (kDegree + 1) * (kDegree + 2) / 2;      // 36 per component at degree 7, 28 at 6, 21 at 5
constexpr int kComps = 5;
constexpr int kPatchDoubles = (kComps * kCoefs + 7) / 8 * 8;    // padded to whole 64-byte lines: 184 (degree 7), 144, 112
constexpr int kHeaderDoubles = 32;
constexpr int kMaxSeg = 12;                                     // = GR_METRIC_MAX_SEG of the header
constexpr int kSegDoubles = 8;                                  // one segment record, 64 bytes
constexpr int kSegOff = kHeaderDoubles;                         // the records start here ...
constexpr int kBodyOff = kHeaderDoubles + kMaxSeg * kSegDoubles;     // ... and the axis terms / patches here (a multiple of 8 doubles)
constexpr int kAxisPolys = 4;                                   // K_m and K_d of g_ϕϕ, then of g_tϕ
constexpr int kAxisDoubles = kAxisPolys * (kDegree + 1);        // per radial row; leading coefficient first
constexpr int kFitNodes = 12;                                   // Chebyshev nodes per patch and direction sampled by the fit
constexpr double kMagic = 1196576085.0;                         // 'GRMT' + 1: the layout of ABI 8

// header slots
enum { H_MAGIC = 0, H_DEGREE, H_R0, H_EMIN, H_NOCT, H_MR, H_NTHETA, H_STRIDE, H_BUILD_ID, H_ERR_VAL, H_ERR_DR, H_ERR_DTH,
       H_RMIN, H_RMAX, H_POLE_FACTOR, H_NSEG, H_NROWS, H_AXIS_OFF, H_PATCH_OFF, H_RES0 };

// One radial segment as the table stores it (and as the kernels read it with scalar loads): r in [r_lo, r_hi), x = dir (r - anchor).
struct SegRec {
    double r_lo, r_hi;
    double anchor;
    double xmin;              // 2^e_lo
    int32_t e_lo, e_hi;       // octaves
    int32_t first_row, n_rows;
    int32_t dir;              // +1: x = r - anchor, -1: x = anchor - r
    int32_t core;             // 1: m_r equal parts of [0, 2^e_lo) come first
    int32_t hard_lo, hard_hi; // the metric's functions change form at r_lo / r_hi: rows are fitted inside the segment only
};
static_assert(sizeof(SegRec) == 8 * kSegDoubles, "segment record layout");

// offset of row i inside a component's block (rows are stored i = p, p-1, ..., 0; row i has p - i + 1 coefficients)
constexpr int row_offset(int i) { return (kDegree - i) * (kDegree - i + 1) / 2; }

// P, ∂u P, ∂v P of the five components from one patch.  coef(k) returns coefficient k of the patch (k < 5 kCoefs) as a double.
// Every coefficient is used where it arrives and nowhere else (a row's leading coefficient enters as c0 v, which both the row's
// value and its derivative start from): on the device a coefficient is half of a 16-byte register tuple fresh from LDS, and a
// coefficient needed again later would keep its whole tuple alive.  The arithmetic is handed over as operations that involve at
// most one coefficient k (T = the number type of u and v: double, or a value with tangents):
//     fma(a, b, c) = a b + c      add(a, b) = a + b      fmak(a, b, k) = a b + k      mulk(a, k) = a k      addk(a, k) = a + k.
// Host and device evaluate the same operations in the same order.
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

// ... and the SECOND derivatives as well: P, ∂u P, ∂v P, ½ ∂uu P, ∂uv P, ½ ∂vv P of the five components (T: a plain number).  What the
// tangent flavour of the kernels needs: the tangents of g and of ∂g with respect to (r, θ) are contractions of these with the
// tangents of (u, v) -- 58 operations per component at degree 5 and 14 more to contract, where the same recurrences run on
// numbers that carry two tangents (gr_tangent.hpp) take 42 x 5.  Coefficients are consumed in the order of eval_patch; a
// coefficient that starts a row is read twice inside its row (as c0 v and as the seed of a derivative).
// op.lift(k) = the coefficient as a T; op.row_done2(k, i, P, Pu, Pv, Puu, Puv, Pvv) is the coefficient stream's hook.
template <class T, class OPS, class C>
GR_TAB_HD __attribute__((always_inline)) inline void eval_patch2(const C& coef, const OPS& op, T u, T v, T P[kComps], T Pu[kComps], T Pv[kComps],
                                                                   T Puu_half[kComps], T Puv[kComps], T Pvv_half[kComps])
{
    static_assert(kDegree >= 3, "the recurrences below special-case the three highest rows");
#pragma unroll
    for (int k = 0; k < kComps; ++k) {
        const int base = k * kCoefs;
        T p_ = T(0.0), pu = T(0.0), puu = T(0.0), pv = T(0.0), puv = T(0.0), pvv = T(0.0);
#pragma unroll
        for (int i = kDegree; i >= 0; --i) {
            const int d = kDegree - i, off = base + row_offset(i);      // the row's degree in v
            // q(v), q'(v), ½ q''(v) of the row by Horner from its leading coefficient
            T q = T(0.0), dq = T(0.0), ddq = T(0.0);
            if (d == 0) {
                q = op.lift(coef(off));
            } else if (d == 1) {
                const double c0 = coef(off);
                q = op.addk(op.mulk(v, c0), coef(off + 1));
                dq = op.lift(c0);
            } else {
                const double c0 = coef(off);
                const T t0 = op.mulk(v, c0);                       // c0 v
                q = op.addk(t0, coef(off + 1));                    // q_1 = c0 v + c1
                dq = op.add(t0, q);                                // q'_2 = 2 c0 v + c1
                q = op.fmak(q, v, coef(off + 2));                  // q_2
                if (d == 2) {
                    ddq = op.lift(c0);                             // ½ q''_2 = c0
                } else {
                    ddq = op.add(t0, dq);                          // ½ q''_3 = 3 c0 v + c1
                    dq = op.fma(dq, v, q);                         // q'_3
                    q = op.fmak(q, v, coef(off + 3));              // q_3
#pragma unroll
                    for (int t = 4; t <= d; ++t) {
                        ddq = op.fma(ddq, v, dq);
                        dq = op.fma(dq, v, q);
                        q = op.fmak(q, v, coef(off + t));
                    }
                }
            }
            // the recurrences in u: (½ Puu, Pu, P) and (Puv, Pv) and ½ Pvv, the leading rows without the zeros they would multiply
            if (d == 0) {
                p_ = q;
            } else if (d == 1) {
                pu = p_;
                p_ = op.fma(p_, u, q);
                pv = dq;
            } else if (d == 2) {
                puu = pu;
                pu = op.fma(pu, u, p_);
                p_ = op.fma(p_, u, q);
                puv = pv;
                pv = op.fma(pv, u, dq);
                pvv = ddq;
            } else {
                puu = op.fma(puu, u, pu);
                pu = op.fma(pu, u, p_);
                p_ = op.fma(p_, u, q);
                puv = op.fma(puv, u, pv);
                pv = op.fma(pv, u, dq);
                pvv = op.fma(pvv, u, ddq);
            }
            if (d >= 2) op.row_done2(k, i, p_, pu, pv, puu, puv, pvv);
        }
        P[k] = p_; Pu[k] = pu; Pv[k] = pv; Puu_half[k] = puu; Puv[k] = puv; Pvv_half[k] = pvv;
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
    static GR_TAB_HD double lift(double k) { return k; }
    static GR_TAB_HD void row_done2(int, int, double&, double&, double&, double&, double&, double&) {}
};

// K(u) and dK/du of one axis polynomial (kDegree + 1 coefficients, leading one first); T as in eval_patch
template <class T, class C>
GR_TAB_HD __attribute__((always_inline)) inline void eval_axis_poly(const C& coef, int first, T u, T& K, T& Ku)
{
    K = T(coef(first));
    Ku = T(0.0);
#pragma unroll
    for (int t = 1; t <= kDegree; ++t) {
        Ku = Ku * u + K;
        K = K * u + coef(first + t);
    }
}

// g_ϕϕ and g_tϕ from their stored forms h (forms 1 and 2 above): g = w h [+ K], with w = sin²θ, w' = 2 sinθ cosθ at the ACTUAL θ
// (w is even about both poles like the stored h, so the fold changes nothing).  Form 2 adds K = K_m + K_d c per component:
// axis[4 j + 0..3] = K_m, ∂r K_m, K_d, ∂r K_d of component 3 + j.
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
template <class T>
GR_TAB_HD __attribute__((always_inline)) inline void axis_terms_apply(const T axis[8], T s, T c, T g[kComps], T dr[kComps], T dth[kComps])
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        g[3 + j] = g[3 + j] + (axis[4 * j + 0] + axis[4 * j + 2] * c);
        dr[3 + j] = dr[3 + j] + (axis[4 * j + 1] + axis[4 * j + 3] * c);
        dth[3 + j] = dth[3 + j] - axis[4 * j + 2] * s;
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
// r0 .. e_max describe segment 0 (the fast path of a one-segment table); n_seg > 1 sends locate() to the records.
struct GridK {
    double r0;
    double xmin;              // 2^e_min
    double mr;                // m_r
    double nth_over_pi;       // n_theta / π
    int32_t e_min, e_max;     // first and last octave
    int32_t m_r, n_theta;
    int32_t n_seg;
};
GR_TAB_HD __attribute__((always_inline)) inline GridK make_gridk(double r0, int e_min, int n_oct, int m_r, int n_theta, int n_seg = 1)
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
    k.n_seg = n_seg;
    return k;
}

// polar patch, local coordinate and dv/dθ (which carries the sign of the fold)
GR_TAB_HD __attribute__((always_inline)) inline void locate_theta(const GridK& k, double th, int& it, double& v, double& sv)
{
    constexpr double kTwoPi = 6.28318530717958647692, kInvTwoPi = 0.15915494309189533577;
    // θ -> [0, π]: even about both poles
    const double q = tab_rint(th * kInvTwoPi);
    const double w = __builtin_fma(-q, kTwoPi, th);
    const double a = tab_fabs(w);
    const double y = a * k.nth_over_pi;
    it = (int)y;
    it = it > k.n_theta - 1 ? k.n_theta - 1 : it;
    v = __builtin_fma(2.0, y - (double)it, -1.0);
    const double sv_mag = 2.0 * k.nth_over_pi;
    sv = w < 0.0 ? -sv_mag : sv_mag;
}

// Radial row, local coordinate and du/dr inside ONE segment without a core, given by plain values (the kernels' fast path: segment 0
// from kernel arguments).  A radius outside the octaves takes the nearest row (below the first octave: its inner edge; beyond the
// last: the polynomial extrapolates -- stage points of a step that overshoots the chart by a hair; the chart callbacks end such a
// ray at the step's end); +inf (a trial step that overflowed) and NaN stay finite: x is clamped two octaves beyond the last.
GR_TAB_HD __attribute__((always_inline)) inline void locate_row0(double r0, double xmin, int e_lo, int e_hi, double mr, int m_r, double r,
                                                                   int& row, double& u, double& su)
{
    double x = r - r0;
    x = x > xmin ? x : xmin;         // (also catches NaN and r <= r0)
    const double xmax = tab_ldexp(1.0, e_hi + 2);
    x = x < xmax ? x : xmax;
    int e = tab_ilogb(x);
    e = e > e_hi ? e_hi : e;
    e = e < e_lo ? e_lo : e;
    const double f = tab_ldexp(x, -e);                 // [1, 2) unless clamped above
    const double z = (f - 1.0) * mr;
    int j = (int)z;
    j = j > m_r - 1 ? m_r - 1 : j;
    u = __builtin_fma(2.0, z - (double)j, -1.0);
    su = tab_ldexp(2.0 * mr, -e);
    row = (e - e_lo) * m_r + j;
}

// ... inside any segment (a record's fields): direction and core as well
GR_TAB_HD __attribute__((always_inline)) inline void locate_row(double anchor, double xmin, int e_lo, int e_hi, int first_row, int dir, int core,
                                                                  double mr, int m_r, double r, int& row, double& u, double& su)
{
    double x = dir >= 0 ? r - anchor : anchor - r;
    const double xmax = tab_ldexp(1.0, e_hi + 2);
    x = x < xmax ? x : xmax;
    x = x > 0.0 ? x : 0.0;                     // (also catches NaN)
    const bool lin = core != 0 && x < xmin;
    x = (x > xmin || lin) ? x : xmin;          // no core: the first octave's inner edge
    int e = tab_ilogb(x);
    e = e > e_hi ? e_hi : e;
    e = e < e_lo ? e_lo : e;
    double f = tab_ldexp(x, -e);               // [1, 2) unless clamped above
    int oct = e - e_lo + (core != 0 ? 1 : 0);
    if (lin) {                                 // the core: [0, 2^e_lo) in m_r parts, each as wide as one of the first octave's
        f = __builtin_fma(x, tab_ldexp(1.0, -e_lo), 1.0);
        e = e_lo;
        oct = 0;
    }
    const double z = (f - 1.0) * mr;
    int j = (int)z;
    j = j > m_r - 1 ? m_r - 1 : j;
    j = j < 0 ? 0 : j;
    u = __builtin_fma(2.0, z - (double)j, -1.0);
    const double sm = tab_ldexp(2.0 * mr, -e);
    su = dir >= 0 ? sm : -sm;
    row = first_row + oct * m_r + j;
}

// patch index, local coordinates and the chain-rule factors du/dr, dv/dθ of a ONE-SEGMENT table (the fast path of the kernels)
GR_TAB_HD __attribute__((always_inline)) inline void locate(const GridK& k, double r, double th, int& row, int& patch, double& u, double& v,
                                                              double& su, double& sv)
{
    int it;
    locate_theta(k, th, it, v, sv);
    locate_row0(k.r0, k.xmin, k.e_min, k.e_max, k.mr, k.m_r, r, row, u, su);
    patch = row * k.n_theta + it;
}

// which segment: the number of records whose lower end lies at or below r (NaN: segment 0)
template <class SegPtr>
GR_TAB_HD __attribute__((always_inline)) inline int segment_of(int n_seg, SegPtr segs, double r)
{
    int s = 0;
    for (int q = 1; q < n_seg; ++q) s += r >= segs[q].r_lo ? 1 : 0;
    return s;
}

// ... of any table, per point (host code, cold device code; the step loop of the kernels reads the record with scalar loads instead:
// gr_device.hpp, TabulatedMetricT::locate_any)
template <class SegPtr>
GR_TAB_HD __attribute__((always_inline)) inline void locate_segments(const GridK& k, SegPtr segs, double r, double th, int& row, int& patch,
                                                                       double& u, double& v, double& su, double& sv)
{
    int it;
    locate_theta(k, th, it, v, sv);
    const int s = segment_of(k.n_seg, segs, r);
    locate_row(segs[s].anchor, segs[s].xmin, segs[s].e_lo, segs[s].e_hi, segs[s].first_row, segs[s].dir, segs[s].core, k.mr, k.m_r, r, row, u, su);
    patch = row * k.n_theta + it;
}

// The grid of a table in the form the kernels' TabulatedMetric::load reads it from gr_config.params (the host unit's
// stage_metric_table, and tests/host_harness.cpp): doubles as doubles, integers as bit fields of doubles -- a double -> int conversion
// on the device would be a vector instruction whose (uniform) result then sits in vector registers for the whole step loop.
inline void stage_params(const double* t, double params[8])
{
    const GridK gk = make_gridk(t[H_R0], (int)t[H_EMIN], (int)t[H_NOCT], (int)t[H_MR], (int)t[H_NTHETA], (int)t[H_NSEG]);
    params[0] = gk.r0;
    params[1] = gk.xmin;
    params[2] = gk.mr;
    params[3] = gk.nth_over_pi;
    params[4] = 0.0;
    const unsigned long long b5 = (unsigned long long)(long long)t[H_PATCH_OFF];      // where the patches start, in doubles
    const unsigned long long b6 = (unsigned long long)(uint32_t)gk.e_min | ((unsigned long long)(uint32_t)gk.e_max << 32);
    const unsigned long long b7 = (unsigned long long)gk.m_r | ((unsigned long long)gk.n_theta << 16) | ((unsigned long long)((int)t[H_POLE_FACTOR] & 3) << 32)
                                  | ((unsigned long long)(gk.n_seg & 0xff) << 40);
    __builtin_memcpy(&params[5], &b5, 8);
    __builtin_memcpy(&params[6], &b6, 8);
    __builtin_memcpy(&params[7], &b7, 8);
}

}  // namespace gr_tab

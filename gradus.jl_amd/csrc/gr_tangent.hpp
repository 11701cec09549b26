// gr_tangent.hpp -- forward-mode scalar with GR_TAN_W tangent directions, for the third build of the integrator
// (GR_REAL_IS_TAN2, namespace grt): the SAME source as the fp64 kernels with `real` = a number that carries derivatives
// with respect to the impact parameters, i.e. "dual-number state through the integrator", which is how the reference
// computes the Jacobian ∂(rₑ, g)/∂(α, β) of its Cunningham transfer functions (ForwardDiff.jacobian around tracegeodesics,
// src/tracing/precision-solvers.jl:401-451).  A Runge-Kutta step commutes with differentiation with respect to the initial
// state, so this integrates the tangent equations with the very steps the value takes.  Every branch looks at VALUES only
// (the comparisons and the conversions to float / int below); the step-size controller sees values AND tangents by default
// (DiffEqBase's norm on Dual state, gr_device.hpp Ray::step; gr_ctx_set "tangent_norm" 0 = values only), and the event
// time's dependence on (α, β) is added by implicit differentiation where the event is resolved (Ray::resolve_event).
//
// GR_TAN_W = 2 (the shipped kernels, the host harness): one number carries ∂/∂α and ∂/∂β.
// GR_TAN_W = 1 (build option, measured in round 4: kernels_tu.hip): one number carries ONE direction and a ray is traced by a
//   PAIR of neighbouring lanes -- the even lane carries ∂/∂α, the odd one ∂/∂β, both compute the same value part with the same
//   instructions (hence the same bits and the same branches; the Dual norm's sums cross the pair with one DPP exchange each).
//   A lane's state is 2/3 as wide and the kernel fits 256 registers, i.e. two waves per SIMD.
#pragma once

#include <cmath>

#ifndef GR_TAN_W
#define GR_TAN_W 2
#endif

#ifdef GR_HOST_HARNESS
#define GR_TAN_FN inline
#else
#define GR_TAN_FN __host__ __device__ __forceinline__
#endif

// (plain members, not an array: with `double d[W]` the optimiser left part of the integrator's state in private memory --
// 956 bytes of scratch on the Kerr kernel where the member form needs 76)
struct gr_tan2 {
    double v;
    double a;            // ∂/∂α (GR_TAN_W = 1: this lane's direction)
#if GR_TAN_W == 2
    double b;            // ∂/∂β
    constexpr gr_tan2() : v(0.0), a(0.0), b(0.0) {}
    constexpr gr_tan2(double x) : v(x), a(0.0), b(0.0) {}          // implicit: literals and plain doubles are constants
#else
    constexpr gr_tan2() : v(0.0), a(0.0) {}
    constexpr gr_tan2(double x) : v(x), a(0.0) {}
#endif
    // values only, and only when asked for
    constexpr explicit operator double() const { return v; }
    constexpr explicit operator float() const { return (float)v; }
    constexpr explicit operator int() const { return (int)v; }
    constexpr explicit operator long long() const { return (long long)v; }
    constexpr explicit operator long() const { return (long)v; }
};
// GR_TAN_B(code): code that exists only when there is a second tangent member
#if GR_TAN_W == 2
#define GR_TAN_B(...) __VA_ARGS__
#else
#define GR_TAN_B(...)
#endif

// a value with the unit tangent in direction `slot` of the FULL Jacobian (0 = ∂/∂α, 1 = ∂/∂β): with GR_TAN_W = 1 the lane's
// own direction `mine` decides whether its single member is that unit or zero
GR_TAN_FN gr_tan2 gr_t_seed(double x, int slot, int mine)
{
    gr_tan2 r(x);
#if GR_TAN_W == 1
    r.a = slot == mine ? 1.0 : 0.0;
#else
    (void)mine;
    r.a = slot == 0 ? 1.0 : 0.0;
    r.b = slot == 1 ? 1.0 : 0.0;
#endif
    return r;
}
// a value whose FIRST tangent member holds `t` (the other zero): d/dλ along the ray through the same arithmetic
GR_TAN_FN gr_tan2 gr_t_along(double x, double t)
{
    gr_tan2 r(x);
    r.a = t;
    return r;
}

// Reciprocal and square root of a plain double for the value parts below.  On the device: hardware seed + two Newton steps
// (<= 1 ulp, as rcp_full / sqrt_fast of gr_device.hpp) instead of the IEEE division / library sqrt sequences; on the host:
// the exact operations.
GR_TAN_FN double gr_d_rcp(double x)
{
#if defined(GR_HOST_HARNESS) || !defined(__HIP_DEVICE_COMPILE__)
    return 1.0 / x;
#else
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#endif
}
GR_TAN_FN double gr_d_sqrt(double x)
{
#if defined(GR_HOST_HARNESS) || !defined(__HIP_DEVICE_COMPILE__)
    return ::sqrt(x);
#else
    if (!(x > 0.0)) return (x == 0.0) ? 0.0 : ::sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, hh = 0.5 * y;
    double r = __builtin_fma(-hh, g, 0.5);
    g = __builtin_fma(g, r, g);
    hh = __builtin_fma(hh, r, hh);
    r = __builtin_fma(-hh, g, 0.5);
    g = __builtin_fma(g, r, g);
    hh = __builtin_fma(hh, r, hh);
    return __builtin_fma(__builtin_fma(-g, g, x), hh, g);
#endif
}

GR_TAN_FN gr_tan2 operator+(gr_tan2 x, gr_tan2 y) { x.v += y.v; x.a += y.a; GR_TAN_B(x.b += y.b;) return x; }
GR_TAN_FN gr_tan2 operator+(gr_tan2 x, double y) { x.v += y; return x; }
GR_TAN_FN gr_tan2 operator+(double y, gr_tan2 x) { x.v += y; return x; }
GR_TAN_FN gr_tan2 operator-(gr_tan2 x, gr_tan2 y) { x.v -= y.v; x.a -= y.a; GR_TAN_B(x.b -= y.b;) return x; }
GR_TAN_FN gr_tan2 operator-(gr_tan2 x, double y) { x.v -= y; return x; }
GR_TAN_FN gr_tan2 operator-(double y, gr_tan2 x) { x.v = y - x.v; x.a = -x.a; GR_TAN_B(x.b = -x.b;) return x; }
GR_TAN_FN gr_tan2 operator-(gr_tan2 x) { x.v = -x.v; x.a = -x.a; GR_TAN_B(x.b = -x.b;) return x; }
GR_TAN_FN gr_tan2 operator*(gr_tan2 x, gr_tan2 y)
{
    gr_tan2 r(x.v * y.v);
    r.a = x.a * y.v + x.v * y.a;
    GR_TAN_B(r.b = x.b * y.v + x.v * y.b;)
    return r;
}
GR_TAN_FN gr_tan2 operator*(gr_tan2 x, double y) { x.v *= y; x.a *= y; GR_TAN_B(x.b *= y;) return x; }
GR_TAN_FN gr_tan2 operator*(double y, gr_tan2 x) { x.v *= y; x.a *= y; GR_TAN_B(x.b *= y;) return x; }
// a b + c with one rounding per part (GR_FMA of the tangent build): value fma(a, b, c), tangent fma(a', b, fma(a, b', c'))
GR_TAN_FN gr_tan2 gr_t_fma(gr_tan2 a, gr_tan2 b, gr_tan2 c)
{
    gr_tan2 r(__builtin_fma(a.v, b.v, c.v));
    r.a = __builtin_fma(a.a, b.v, __builtin_fma(a.v, b.a, c.a));
    GR_TAN_B(r.b = __builtin_fma(a.b, b.v, __builtin_fma(a.v, b.b, c.b));)
    return r;
}
GR_TAN_FN gr_tan2 gr_t_fma(gr_tan2 a, double b, gr_tan2 c)
{
    gr_tan2 r(__builtin_fma(a.v, b, c.v));
    r.a = __builtin_fma(a.a, b, c.a);
    GR_TAN_B(r.b = __builtin_fma(a.b, b, c.b);)
    return r;
}
GR_TAN_FN gr_tan2 gr_t_fma(double a, gr_tan2 b, gr_tan2 c) { return gr_t_fma(b, a, c); }
GR_TAN_FN gr_tan2 gr_t_fma(gr_tan2 a, gr_tan2 b, double c)
{
    gr_tan2 r(__builtin_fma(a.v, b.v, c));
    r.a = __builtin_fma(a.a, b.v, a.v * b.a);
    GR_TAN_B(r.b = __builtin_fma(a.b, b.v, a.v * b.b);)
    return r;
}
GR_TAN_FN gr_tan2 gr_t_fma(gr_tan2 a, double b, double c)
{
    gr_tan2 r(__builtin_fma(a.v, b, c));
    r.a = a.a * b;
    GR_TAN_B(r.b = a.b * b;)
    return r;
}
GR_TAN_FN gr_tan2 gr_t_fma(double a, gr_tan2 b, double c) { return gr_t_fma(b, a, c); }
GR_TAN_FN gr_tan2 gr_t_fma(double a, double b, gr_tan2 c) { c.v = __builtin_fma(a, b, c.v); return c; }
GR_TAN_FN double gr_t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
GR_TAN_FN gr_tan2 operator/(gr_tan2 x, gr_tan2 y)
{
    const double i = gr_d_rcp(y.v), q = x.v * i;
    gr_tan2 r(q);
    r.a = (x.a - q * y.a) * i;
    GR_TAN_B(r.b = (x.b - q * y.b) * i;)
    return r;
}
GR_TAN_FN gr_tan2 operator/(gr_tan2 x, double y) { x.v /= y; x.a /= y; GR_TAN_B(x.b /= y;) return x; }
GR_TAN_FN gr_tan2 operator/(double x, gr_tan2 y)
{
    const double i = gr_d_rcp(y.v), q = x * i, m = -q * i;
    gr_tan2 r(q);
    r.a = m * y.a;
    GR_TAN_B(r.b = m * y.b;)
    return r;
}
GR_TAN_FN gr_tan2& operator+=(gr_tan2& x, gr_tan2 y) { x = x + y; return x; }
GR_TAN_FN gr_tan2& operator-=(gr_tan2& x, gr_tan2 y) { x = x - y; return x; }
GR_TAN_FN gr_tan2& operator*=(gr_tan2& x, gr_tan2 y) { x = x * y; return x; }
GR_TAN_FN gr_tan2& operator+=(gr_tan2& x, double y) { x.v += y; return x; }
GR_TAN_FN gr_tan2& operator-=(gr_tan2& x, double y) { x.v -= y; return x; }
GR_TAN_FN gr_tan2& operator*=(gr_tan2& x, double y) { x = x * y; return x; }

#define GR_TAN_CMP(op)                                                            \
    GR_TAN_FN constexpr bool operator op(gr_tan2 x, gr_tan2 y) { return x.v op y.v; }       \
    GR_TAN_FN constexpr bool operator op(gr_tan2 x, double y) { return x.v op y; }          \
    GR_TAN_FN constexpr bool operator op(double x, gr_tan2 y) { return x op y.v; }
GR_TAN_CMP(<) GR_TAN_CMP(>) GR_TAN_CMP(<=) GR_TAN_CMP(>=) GR_TAN_CMP(==) GR_TAN_CMP(!=)
#undef GR_TAN_CMP

// elementary functions (plain-double overloads beside them so that one macro serves both)
GR_TAN_FN double gr_t_abs(double x) { return x < 0.0 ? -x : x; }
GR_TAN_FN gr_tan2 gr_t_abs(gr_tan2 x) { return x.v < 0.0 ? -x : x; }
GR_TAN_FN double gr_t_max(double x, double y) { return x > y ? x : y; }
GR_TAN_FN gr_tan2 gr_t_max(gr_tan2 x, gr_tan2 y) { return x.v > y.v ? x : y; }
GR_TAN_FN double gr_t_min(double x, double y) { return x < y ? x : y; }
GR_TAN_FN gr_tan2 gr_t_min(gr_tan2 x, gr_tan2 y) { return x.v < y.v ? x : y; }
GR_TAN_FN double gr_t_rint(double x) { return ::rint(x); }
GR_TAN_FN double gr_t_floor(double x) { return ::floor(x); }
GR_TAN_FN double gr_t_sqrt(double x) { return gr_d_sqrt(x); }
GR_TAN_FN gr_tan2 gr_t_rint(gr_tan2 x) { return gr_tan2(::rint(x.v)); }        // piecewise constant
GR_TAN_FN gr_tan2 gr_t_floor(gr_tan2 x) { return gr_tan2(::floor(x.v)); }
GR_TAN_FN gr_tan2 gr_t_sqrt(gr_tan2 x)
{
    const double s = gr_d_sqrt(x.v), h = 0.5 * gr_d_rcp(s);
    x.v = s; x.a *= h; GR_TAN_B(x.b *= h;)
    return x;
}
GR_TAN_FN gr_tan2 gr_t_rcp(gr_tan2 x)
{
    const double i = gr_d_rcp(x.v), m = -i * i;
    x.v = i; x.a *= m; GR_TAN_B(x.b *= m;)
    return x;
}
GR_TAN_FN gr_tan2 gr_t_rsq(gr_tan2 x)
{
    const double i = gr_d_rcp(gr_d_sqrt(x.v)), m = -0.5 * i * gr_d_rcp(x.v);
    x.v = i; x.a *= m; GR_TAN_B(x.b *= m;)
    return x;
}
GR_TAN_FN gr_tan2 gr_t_pow(gr_tan2 x, gr_tan2 y)          // y is a constant exponent wherever the integrator calls this
{
    const double p = ::pow(x.v, y.v), dd = y.v * p / x.v;
    x.v = p; x.a *= dd; GR_TAN_B(x.b *= dd;)
    return x;
}
GR_TAN_FN gr_tan2 gr_t_atan(gr_tan2 x)
{
    const double w = 1.0 / (1.0 + x.v * x.v);
    x.v = ::atan(x.v); x.a *= w; GR_TAN_B(x.b *= w;)
    return x;
}
// (sin, cos) of a tangent number from the (sin, cos) of its value: the tangents are cos θ θ' and -sin θ θ'
GR_TAN_FN void gr_t_sincos_lift(gr_tan2 th, double s, double c, gr_tan2& s_out, gr_tan2& c_out)
{
    s_out = gr_tan2(s);
    c_out = gr_tan2(c);
    s_out.a = c * th.a; c_out.a = -s * th.a;
    GR_TAN_B(s_out.b = c * th.b; c_out.b = -s * th.b;)
}
// Σ over ALL directions of the Jacobian of x'², formed the same way in both lanes of a pair (GR_TAN_W = 1: the partner lane
// holds the other direction; t + partner's t is the same sum in both, so they keep taking the same decisions)
GR_TAN_FN double gr_t_tan_sq(gr_tan2 x)
{
#if GR_TAN_W == 1
    const double t = x.a * x.a;
#if defined(GR_HOST_HARNESS) || !defined(__HIP_DEVICE_COMPILE__)
    return t;
#else
    return t + __shfl_xor(t, 1, 64);
#endif
#else
    return x.a * x.a + x.b * x.b;
#endif
}

// Pin a number's tangent parts to this point of the program.  Branches look at values only, so nothing stops the optimiser
// from sinking a stage's TANGENT arithmetic past the (value-dependent) branches of the later stages -- it did: the tangents of
// all six right-hand sides of a step ended up in one 3000-instruction block behind the last stage, with every value
// intermediate they need kept alive until then (900 bytes of scratch per lane, 90 ms instead of 50 for 1024² rays).  An empty
// volatile asm that "modifies" the tangent registers keeps each stage's tangents inside its stage.
GR_TAN_FN void gr_t_pin(gr_tan2& x)
{
#if !defined(GR_HOST_HARNESS) && defined(__HIP_DEVICE_COMPILE__)
#if GR_TAN_W == 2
    asm volatile("" : "+v"(x.a), "+v"(x.b));
#else
    asm volatile("" : "+v"(x.a));
#endif
#else
    (void)x;
#endif
}

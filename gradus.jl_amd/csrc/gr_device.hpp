// gr_device.hpp -- device-side physics and the per-lane Tsit5 ray integrator (gfx950).
//
// One null geodesic per work-item, the whole ODE state in VGPRs.  What this replaces in the
// reference (Gradus.jl, citations into /root/reference):
//   metric_components + ForwardDiff Jacobian   src/metrics/kerr-metric.jl:11-28,
//                                              src/metrics/johannsen-ad.jl:12-34,
//                                              src/tracing/method-implementations/auto-diff.jl:206-211
//   inverse_metric_components                  auto-diff.jl:59-76
//   compute_geodesic_equation                  auto-diff.jl:115-141
//   constrain_time                             auto-diff.jl:161-179
//   Tsit5 / PI controller / initial dt / callbacks  (OrdinaryDiffEq, DiffEqBase)
//   chart callback                             src/tracing/charts.jl:9-23
//   ThinDisc distance_to_disc                  src/geometry/discs/thin-disc.jl:20-26
//   unpack_solution -> GeodesicPoint           src/solution-processing.jl:86-112
//   PointFunctions / redshift                  src/const-point-functions.jl:26-79, src/redshift.jl:93-220
//
// Design notes (MI355X): fp64 VALU-bound.  Everything is fully unrolled with compile-time
// tableau constants so no array is runtime-indexed (no scratch); divisions go through one
// shared v_rcp_f64 + Newton per RHS; sin/cos use a branch-free Cody-Waite + minimax kernel
// instead of the library's Payne-Hanek-capable sincos; the error norm is kept squared so the
// step controller needs one log2 and one exp2 per step and no sqrt.
#pragma once

#include <stdint.h>

#include "../../include/gradus_mi355x.h"

// Scalar type of the integrator.  The fp32 build (kernels_tu.hip with -DGR_TU_F32) defines GR_REAL_IS_FLOAT
// and GR_NS = gr32 and is compiled with -Xclang -cl-single-precision-constant so that every
// floating literal below is a float there.  All I/O (gr_point, images, tables) stays double.
#ifndef GR_NS
#define GR_NS gr
#endif
#if defined(GR_REAL_IS_TAN2)
// third build: real = a forward-mode number with two tangent directions (gr_tangent.hpp)
#include "gr_tangent.hpp"
typedef gr_tan2 gr_real_t;
#define GR_FMA(a, b, c) gr_t_fma((a), (b), (c))
#define GR_FABS gr_t_abs
#define GR_FMAX gr_t_max
#define GR_FMIN gr_t_min
#define GR_RINT gr_t_rint
#define GR_FLOOR gr_t_floor
#define GR_SQRT gr_t_sqrt
#define GR_POW gr_t_pow
#define GR_ATAN gr_t_atan
#define GR_EPS 2.220446049250313e-16
#elif defined(GR_REAL_IS_FLOAT)
typedef float gr_real_t;
#define GR_FMA __builtin_fmaf
#define GR_FABS __builtin_fabsf
#define GR_FMAX __builtin_fmaxf
#define GR_FMIN __builtin_fminf
#define GR_RINT __builtin_rintf
#define GR_FLOOR __builtin_floorf
#define GR_SQRT __builtin_sqrtf
#define GR_POW ::powf
#define GR_ATAN ::atanf
#define GR_EPS 1.1920929e-07f
#else
typedef double gr_real_t;
#define GR_FMA __builtin_fma
#define GR_FABS __builtin_fabs
#define GR_FMAX __builtin_fmax
#define GR_FMIN __builtin_fmin
#define GR_RINT __builtin_rint
#define GR_FLOOR __builtin_floor
#define GR_SQRT __builtin_sqrt
#define GR_POW ::pow
#define GR_ATAN ::atan
#define GR_EPS 2.220446049250313e-16
#endif

#ifndef GR_GENERIC_MIN_WAVES
#define GR_GENERIC_MIN_WAVES 2
#endif
// Stages of the Tsit5 step whose accelerations the one-ray-per-lane kernels of the non-Kerr metrics park in LDS (ParkA below)
// to run three waves per SIMD instead of two.  MEASURED in round 4 and left OFF: with -DGR_PARK_DEFAULT=5 every such kernel fits
// 142-168 registers without scratch (205 -> 157 for Johannsen) and 2.4 waves are resident per SIMD instead of 1.7 -- and the
// issue rate does not move (0.850 against 0.847 per 4 clocks, FP64 pipe 0.77 against 0.75 busy; profiles/r4e_c4_*): Johannsen
// 1024² 8.44 against 8.07 ms, 2048² 27.8 against 28.5; Bumblebee 2048² 22.9 against 22.4; dilaton-axion 38.5 against 39.2.  The
// slots two waves leave empty are not waiting for a third wave: 89 % of these kernels' instructions are FP64 (Kerr: 85 %), which
// the pipe takes at 0.77 per 4 clocks at best.  The fp32 kernels have registers to spare and never park.
#ifndef GR_PARK_DEFAULT
#define GR_PARK_DEFAULT 0
#endif
// MeshAccretionGeometry (GR_DISC_MESH) exists in the fp64 kernels only (and in the host harness); the fp32 and tangent flavours
// do not instantiate it and the host unit refuses the combination
#if defined(GR_REAL_IS_TAN2) || defined(GR_REAL_IS_FLOAT)
#define GR_HAS_MESH 0
#else
#define GR_HAS_MESH 1
#endif
// tests due in one wave after a step up to which the wave takes them one by one, all lanes on one test (above: each lane its own)
#ifndef GR_MESH_WAVE_MAX
#define GR_MESH_WAVE_MAX 16
#endif

#ifdef GR_HOST_HARNESS
// tests/host_harness.cpp compiles this header with g++ to trace single rays on the CPU next to
// the oracle.  Test infrastructure only: the shipped library never defines GR_HOST_HARNESS.
#include <cmath>
#define GR_DEV inline
#ifdef GR_REAL_IS_TAN2
#define GR_RCP_SEED(x) gr_t_rcp(x)
#define GR_RSQ_SEED(x) gr_t_rsq(x)
#else
#define GR_RCP_SEED(x) (1.0 / (x))
#define GR_RSQ_SEED(x) (1.0 / std::sqrt(x))
#endif
#define GR_LOG2F(x) std::log2((float)(x))
#define GR_EXP2F(x) std::exp2((float)(x))
#define GR_RCPF(x) (1.0f / (x))
#else
#include <hip/hip_runtime.h>
#define GR_DEV __device__ __forceinline__
#if defined(GR_REAL_IS_TAN2)
#define GR_RCP_SEED(x) gr_t_rcp(x)
#define GR_RSQ_SEED(x) gr_t_rsq(x)
#elif defined(GR_REAL_IS_FLOAT)
#define GR_RCP_SEED(x) __builtin_amdgcn_rcpf(x)
#define GR_RSQ_SEED(x) __builtin_amdgcn_rsqf(x)
#else
#define GR_RCP_SEED(x) __builtin_amdgcn_rcp(x)
#define GR_RSQ_SEED(x) __builtin_amdgcn_rsq(x)
#endif
#define GR_LOG2F(x) __builtin_amdgcn_logf(x)     // v_log_f32
#define GR_EXP2F(x) __builtin_amdgcn_exp2f(x)    // v_exp_f32
#define GR_RCPF(x) __builtin_amdgcn_rcpf(x)      // v_rcp_f32
#endif

#include "gr_tabmetric.hpp"

namespace GR_NS {

typedef gr_real_t real;
#ifdef GR_REAL_IS_TAN2
typedef double creal;     // tableau and other compile-time tables: plain numbers (constants have no tangent)
#else
typedef real creal;
#endif
// The step size and everything formed from it alone: a PLAIN number in every build.  The reference's dt has the type of its
// time span (Float64) also when the state carries dual numbers, and a tangent build that lets h be a (value, 0, 0) triple pays
// one dead FMA per tangent member in every weighted sum of a step.
typedef creal hreal;

// PACKED single precision (fp32 build only).  The 157 TFLOP/s FP32 vector peak of CDNA4 is v_pk_fma_f32's: a scalar v_fma_f32
// issues at the FP64 rate (one per 4 clocks and SIMD, profiles/r5_valu_calib.json).  The weighted sums of a Tsit5 step -- stage
// arguments, new state, error estimate -- are axpys over the FOUR components of one ray with a common coefficient: components
// (0, 1) and (2, 3) ride in the two halves of one packed instruction (the coefficient broadcast from one scalar register,
// op_sel_hi), no second ray, no masks.  -DGR_PK_F32=0 builds the scalar sums (the A/B of profiles/r6_c5f32_packed_ab.log).
#ifndef GR_PK_F32
#ifdef GR_REAL_IS_FLOAT
#define GR_PK_F32 1
#else
#define GR_PK_F32 0
#endif
#endif
#if GR_PK_F32
typedef float gr_f2 __attribute__((ext_vector_type(2)));
#define GR_PK2(arr, i) (gr_f2{ (arr)[(i)], (arr)[(i) + 1] })
#define GR_PKFMA(c, a, b) __builtin_elementwise_fma((gr_f2)(c), (a), (b))
#endif

// ---------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------
GR_DEV real rcp_full(real x)
{
#if defined(GR_REAL_IS_TAN2) && !defined(GR_TAN_OLD_RCP)
    return gr_t_rcp(x);      // value: seed + two Newton steps on the plain double; tangents: -x' / x² (no Newton steps on tangents)
#else
    // v_rcp_f64 seed (4.6e-8 relative, measured) + two Newton steps: <= 1 ulp for normal, finite x
    real r = GR_RCP_SEED(x);
    real e = GR_FMA(-x, r, 1.0);
    r = GR_FMA(r, e, r);
    e = GR_FMA(-x, r, 1.0);
    r = GR_FMA(r, e, r);
    return r;
#endif
}
GR_DEV real rcp_raw(real x) { return GR_RCP_SEED(x); }
GR_DEV real rcp_fast(real x);
// The reciprocal inside the fused right-hand sides: 1 = seed + one Newton step (2e-15 relative, the rounding level of the ~85
// operations it feeds), 2 = two steps (<= 1 ulp).
#ifndef GR_RHS_RCP_STEPS
#define GR_RHS_RCP_STEPS 1
#endif
GR_DEV real rcp_rhs(real x)
{
#if GR_RHS_RCP_STEPS == 1
    return rcp_fast(x);
#else
    return rcp_full(x);
#endif
}
GR_DEV real rcp_fast(real x)
{
#if defined(GR_REAL_IS_TAN2) && !defined(GR_TAN_OLD_RCP)
    return gr_t_rcp(x);
#else
    // one Newton step: ~2e-15 relative
    real r = GR_RCP_SEED(x);
    real e = GR_FMA(-x, r, 1.0);
    return GR_FMA(r, e, r);
#endif
}
GR_DEV real sqrt_fast(real x)
{
#if defined(GR_REAL_IS_TAN2) && !defined(GR_TAN_OLD_RCP)
    return gr_t_sqrt(x);
#else
    // x > 0, normal range.  rsq seed + two coupled Newton steps (Goldschmidt): <= 1 ulp
    if (!(x > 0.0)) return (x == 0.0) ? 0.0 : GR_SQRT(x);
    real y = GR_RSQ_SEED(x);
    real g = x * y, hh = 0.5 * y;
    real r = GR_FMA(-hh, g, 0.5);
    g = GR_FMA(g, r, g);
    hh = GR_FMA(hh, r, hh);
    r = GR_FMA(-hh, g, 0.5);
    g = GR_FMA(g, r, g);
    hh = GR_FMA(hh, r, hh);
    const real d = GR_FMA(-g, g, x);
    return GR_FMA(d, hh, g);
#endif
}
GR_DEV int sgn(real x) { return (x > 0.0) - (x < 0.0); }
// A value that is the same in every lane, forced into scalar registers.  Products of metric parameters (a², 2M, -3 α13, ...)
// are formed by the vector ALU (the scalar unit has no FP64), so the compiler keeps them in VGPRs for the whole step loop;
// at the 168-register budget it spilled exactly those to scratch and reloaded them in every stage.  Two v_readfirstlane
// once per kernel put them where uniform values belong.
#ifndef GR_UNIFORM_TO_SGPR
#define GR_UNIFORM_TO_SGPR 1
#endif
GR_DEV real uni(real x)
{
#if defined(GR_HOST_HARNESS) || defined(GR_REAL_IS_TAN2) || !GR_UNIFORM_TO_SGPR
    return x;
#elif defined(GR_REAL_IS_FLOAT)
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
#else
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
#endif
}
// max(|a|, |b|) as ONE instruction.  fmax(fabs(a), fabs(b)) compiles to three under IEEE mode (each fabs is made canonical by
// its own v_max x, |a|, |a| before the real maximum): 15 of the step's FP64 instructions were such canonicalisations (error
// norm scales, the event pre-filter's max |A^θ|).  The source modifiers of v_max_f64 take the absolute values for free and the
// instruction's NaN behaviour is fmax's (the other operand is returned).
#if defined(GR_HOST_HARNESS) || defined(GR_REAL_IS_TAN2)
GR_DEV real absmax(real a, real b) { return GR_FMAX(GR_FABS(a), GR_FABS(b)); }
#elif defined(GR_REAL_IS_FLOAT)
GR_DEV real absmax(real a, real b)
{
    real r;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
GR_DEV real absmax(real a, real b)
{
    real r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif
// a b + K with the constant K held in scalar registers.  The compiler selects the two-address v_fmac_f64 for an FMA whose
// addend is a constant and has to build that constant in the destination VECTOR registers first (two v_mov_b32 per use);
// the three-address form reads it from an SGPR pair filled by the scalar unit, which has issue slots to spare here.
#if defined(GR_HOST_HARNESS) || defined(GR_REAL_IS_TAN2) || defined(GR_REAL_IS_FLOAT)
template <class T>
GR_DEV T fma_sk(T a, T b, creal k) { return GR_FMA(a, b, (T)k); }
#else
GR_DEV real fma_sk(real a, real b, creal k)
{
    real r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#endif
GR_DEV void gr_atomic_add(double* p, double v)
{
#ifdef GR_HOST_HARNESS
    *p += v;
#else
    atomicAdd(p, v);
#endif
}
GR_DEV float fast_log2f(float x) { return GR_LOG2F(x); }
GR_DEV float fast_exp2f(float x) { return GR_EXP2F(x); }

// The full evaluation behind a rotation that is out of range is taken in 1-2 % of the wave-steps.  On plain doubles (the value
// part of the tangent build) it is cheap enough for the optimiser to if-convert -- compute both, select -- which fused the six
// stages of a step into one 3000-instruction block, evaluated the full sin/cos at EVERY stage and spilled 900 bytes per lane
// to scratch (90 ms instead of 50 for 1024² rays).  An empty volatile asm cannot be speculated: the branch stays a branch.
#if defined(GR_REAL_IS_TAN2) && !defined(GR_HOST_HARNESS)
#define GR_NO_SPECULATION() asm volatile("")
#else
#define GR_NO_SPECULATION()
#endif

// sin and cos of x for moderate |x| (|x| < ~1e5): two-term Cody-Waite reduction by pi/2 with
// exact-product FMAs, then the fdlibm minimax kernels on [-pi/4, pi/4].  < 1 ulp each.
template <class T>
GR_DEV void sincos_fast_impl(T x, T& s_out, T& c_out)
{
    const T TWO_OVER_PI = 6.36619772367581382433e-01;
    const T PIO2_HI = 1.57079632679489655800e+00;
    const T PIO2_LO = 6.12323399573676603587e-17;
    const T kf = GR_RINT(x * TWO_OVER_PI);
    T y = GR_FMA(-kf, PIO2_HI, x);
    y = GR_FMA(-kf, PIO2_LO, y);
    const int q = (int)kf;
    const T z = y * y;
    // __kernel_sin
    const T S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    T ps = GR_FMA(z, S6, S5);
    ps = GR_FMA(z, ps, S4);
    ps = GR_FMA(z, ps, S3);
    ps = GR_FMA(z, ps, S2);
    ps = GR_FMA(z, ps, S1);
    const T sn = GR_FMA(y * z, ps, y);
    // __kernel_cos
    const T C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    T pc = GR_FMA(z, C6, C5);
    pc = GR_FMA(z, pc, C4);
    pc = GR_FMA(z, pc, C3);
    pc = GR_FMA(z, pc, C2);
    pc = GR_FMA(z, pc, C1);
    const T hz = 0.5 * z;
    const T w = 1.0 - hz;
    const T cs = w + (((1.0 - w) - hz) + z * (z * pc));
    // quadrant
    const T s0 = (q & 1) ? cs : sn;
    const T c0 = (q & 1) ? sn : cs;
    s_out = (q & 2) ? -s0 : s0;
    c_out = ((q + 1) & 2) ? -c0 : c0;
}

#if defined(GR_REAL_IS_TAN2) && !defined(GR_TAN_OLD_SINCOS)
// the polynomials run on the VALUE; the tangents are cos θ θ' and -sin θ θ'
GR_DEV void sincos_fast(real x, real& s_out, real& c_out)
{
    double sv, cv;
    sincos_fast_impl<double>(x.v, sv, cv);
    gr_t_sincos_lift(x, sv, cv, s_out, c_out);
}
#else
GR_DEV void sincos_fast(real x, real& s_out, real& c_out) { sincos_fast_impl<real>(x, s_out, c_out); }
#endif

// sin and cos of θ0 + δ from (sin θ0, cos θ0) by rotation, for the Runge-Kutta stage points of
// one step.  |δ| <= 1/32 covers every stage of 99 % of the steps of a WAVE at tolerance 1e-9 (the largest |δ| over
// the 64 lanes is below 2^-5 in 99.05 % and below 2^-4 in all of the wave-steps of the bench image: tests/host_harness.cpp
// hh_wave_stats) and needs two three-term polynomials (truncation 2.5e-18 / 2.3e-17, below half an ulp) and four
// FMAs: no range reduction, no quadrant logic.  Larger δ takes the full evaluation.
constexpr creal SINCOS_ROT_MAX = 0.03125;
// The two polynomials each open with an FMA that has TWO constant operands, and a VALU instruction of this ISA reads at
// most one scalar / literal operand: the other constant has to sit in a vector register.  Left to itself the compiler
// rebuilds both (v_mov_b32 pairs + a copy) at each of the six stage points of a step -- 36 of the step's ~130 non-FP64
// vector instructions.  RotK keeps the two addends in registers for the life of the ray (4 VGPRs), made opaque once in
// Ray::init so that they are not rematerialised.
#ifndef GR_ROT_MODE
#define GR_ROT_MODE 3
#endif
struct RotK {
    real s2, c2;      // 1/120, 1/24 (GR_ROT_MODE 1 only)
    GR_DEV void load()
    {
        s2 = 8.3333333333333333e-03;
        c2 = 4.1666666666666664e-02;
#if GR_ROT_MODE == 1 && !defined(GR_HOST_HARNESS) && !defined(GR_REAL_IS_TAN2)
        asm volatile("" : "+v"(s2), "+v"(c2));
#endif
    }
};
template <class T>
GR_DEV void sincos_rot_impl(const RotK& k, T th0, T s0, T c0, T th, T& s_out, T& c_out)
{
    const T d = th - th0;
    if (GR_FABS(d) <= SINCOS_ROT_MAX) {
        const T z = d * d;
#if GR_ROT_MODE == 2
        // every FMA with ONE non-inline constant (the second operand is 1.0 or -0.5, which the ISA encodes inline):
        //   sin δ = δ + δ z S1 (1 + z (S2/S1) (1 + z S3/S2)),   cos δ - 1 = z (-1/2 + z C2 (1 + z C3/C2))
        // three multiplications more than Horner's form, six register moves fewer per stage point
        const T u1 = GR_FMA(z, -2.3809523809523808e-02, 1.0);          // S3/S2 = -1/42
        const T u3 = GR_FMA(z * u1, -5.0e-02, 1.0);                    // S2/S1 = -1/20
        const T sd = GR_FMA((d * z) * u3, -1.6666666666666666e-01, d); // sin δ
        const T t1 = GR_FMA(z, -3.3333333333333333e-02, 1.0);          // C3/C2 = -1/30
        const T pc = GR_FMA(z * t1, 4.1666666666666664e-02, -0.5);
#elif GR_ROT_MODE == 3
        // the two-constant FMA opened as a product and a sum (one scalar operand each)
        T ps, pc;
        {
#pragma clang fp contract(off)
            ps = z * -1.9841269841269841e-04;
            ps = ps + 8.3333333333333333e-03;
            pc = z * -1.3888888888888889e-03;
            pc = pc + 4.1666666666666664e-02;
        }
        ps = fma_sk(z, ps, -1.6666666666666666e-01);
        const T sd = GR_FMA(d * z, ps, d);                    // sin δ
        pc = GR_FMA(z, pc, -0.5);
#else
        T ps = GR_FMA(z, -1.9841269841269841e-04, (T)k.s2);
        ps = GR_FMA(z, ps, -1.6666666666666666e-01);
        const T sd = GR_FMA(d * z, ps, d);                    // sin δ
        T pc = GR_FMA(z, -1.3888888888888889e-03, (T)k.c2);
        pc = GR_FMA(z, pc, -0.5);
#endif
        const T cm1 = z * pc;                                        // cos δ - 1
        s_out = GR_FMA(c0, sd, GR_FMA(s0, cm1, s0));
        c_out = GR_FMA(-s0, sd, GR_FMA(c0, cm1, c0));
    } else {
        GR_NO_SPECULATION();
        sincos_fast_impl<T>(th, s_out, c_out);
    }
}

#if defined(GR_REAL_IS_TAN2) && !defined(GR_TAN_OLD_SINCOS)
// rotation on the VALUES (the tangents of sin θ0, cos θ0 are not needed: d sin θ = cos θ θ', d cos θ = -sin θ θ')
GR_DEV void sincos_rot(const RotK& k, real th0, real s0, real c0, real th, real& s_out, real& c_out)
{
    double sv, cv;
    sincos_rot_impl<double>(k, th0.v, s0.v, c0.v, th.v, sv, cv);
    gr_t_sincos_lift(th, sv, cv, s_out, c_out);
}
#else
GR_DEV void sincos_rot(const RotK& k, real th0, real s0, real c0, real th, real& s_out, real& c_out)
{
    sincos_rot_impl<real>(k, th0, s0, c0, th, s_out, c_out);
}
#endif

// The same rotation for the five INTERIOR stage points of a step, with two-term polynomials: sin δ = δ (1 + z (S1 + z S2)),
// cos δ - 1 = z (-1/2 + z C2).  Truncation δ⁶/5040 resp. δ⁶/720 of the leading term: 1.8e-13 / 1.3e-12 at the bound |δ| = 1/32,
// 2.8e-15 / 2e-14 at 2⁻⁶, below an ulp from 2⁻⁷ down -- and the wave's largest |δ| is below 2⁻⁶ in 84 % and below 2⁻⁷ in 60 % of
// the wave-steps of the bench image (hh_wave_stats).  An interior stage's sin θ, cos θ enter ONE right-hand side and are
// formed anew from the step's base at the next stage: the error does not accumulate (the base itself, rotated once per
// accepted step, keeps the three-term form and its resynchronisation), and 1e-12 of an acceleration for one stage of the
// rare large steps is three orders below the integration tolerance the kernels run at.  Three instructions per stage fewer.
#ifndef GR_ROT_STAGE_TERMS
#define GR_ROT_STAGE_TERMS 2
#endif
template <class T>
GR_DEV void sincos_rot_stage_impl(const RotK& k, T th0, T s0, T c0, T th, T& s_out, T& c_out)
{
#if GR_ROT_STAGE_TERMS == 2 && GR_ROT_MODE == 3
    const T d = th - th0;
    if (GR_FABS(d) <= SINCOS_ROT_MAX) {
        const T z = d * d;
        T ps;
        {
#pragma clang fp contract(off)
            ps = z * 8.3333333333333333e-03;
            ps = ps + -1.6666666666666666e-01;
        }
        const T sd = GR_FMA(d * z, ps, d);                         // sin δ
        const T pc = GR_FMA(z, 4.1666666666666664e-02, -0.5);        // one scalar constant, one inline constant
        const T cm1 = z * pc;                                        // cos δ - 1
        s_out = GR_FMA(c0, sd, GR_FMA(s0, cm1, s0));
        c_out = GR_FMA(-s0, sd, GR_FMA(c0, cm1, c0));
    } else {
        GR_NO_SPECULATION();
        sincos_fast_impl<T>(th, s_out, c_out);
    }
#else
    sincos_rot_impl<T>(k, th0, s0, c0, th, s_out, c_out);
#endif
}

#if defined(GR_REAL_IS_TAN2) && !defined(GR_TAN_OLD_SINCOS)
GR_DEV void sincos_rot_stage(const RotK& k, real th0, real s0, real c0, real th, real& s_out, real& c_out)
{
    double sv, cv;
    sincos_rot_stage_impl<double>(k, th0.v, s0.v, c0.v, th.v, sv, cv);
    gr_t_sincos_lift(th, sv, cv, s_out, c_out);
}
#else
GR_DEV void sincos_rot_stage(const RotK& k, real th0, real s0, real c0, real th, real& s_out, real& c_out)
{
    sincos_rot_stage_impl<real>(k, th0, s0, c0, th, s_out, c_out);
}
#endif

// ---------------------------------------------------------------------------------------
// Forward-mode dual number with two partials, for metrics without hand-written derivatives
// (the reference differentiates every metric this way, auto-diff.jl:206-211).
// ---------------------------------------------------------------------------------------
// Typed since round 4: every metric of the catalogue is built from functions of r alone and of θ alone (Δ(r), sin²θ, ...) that
// meet in a few products (Σ, A); a sub-expression that depends on one coordinate carries ONE partial, the other is absent from
// the type instead of being a zero that is multiplied and added through (x·0 cannot be folded under IEEE rules).  The values
// are those of the untyped form (a term a·0 + b is b); DualR = ∂_r only, DualT = ∂_θ only, Dual2 = both.
template <bool HA, bool HB>
struct DualP {
    real v, a, b;          // a = ∂_r (read only where HA), b = ∂_θ (only where HB)
};
typedef DualP<true, true> Dual2;
typedef DualP<true, false> DualR;
typedef DualP<false, true> DualT;
template <bool A1, bool B1, bool A2, bool B2>
GR_DEV DualP<A1 || A2, B1 || B2> operator+(DualP<A1, B1> x, DualP<A2, B2> y)
{
    DualP<A1 || A2, B1 || B2> z{ x.v + y.v, 0.0, 0.0 };
    if constexpr (A1 && A2) z.a = x.a + y.a; else if constexpr (A1) z.a = x.a; else if constexpr (A2) z.a = y.a;
    if constexpr (B1 && B2) z.b = x.b + y.b; else if constexpr (B1) z.b = x.b; else if constexpr (B2) z.b = y.b;
    return z;
}
template <bool A1, bool B1, bool A2, bool B2>
GR_DEV DualP<A1 || A2, B1 || B2> operator-(DualP<A1, B1> x, DualP<A2, B2> y)
{
    DualP<A1 || A2, B1 || B2> z{ x.v - y.v, 0.0, 0.0 };
    if constexpr (A1 && A2) z.a = x.a - y.a; else if constexpr (A1) z.a = x.a; else if constexpr (A2) z.a = -y.a;
    if constexpr (B1 && B2) z.b = x.b - y.b; else if constexpr (B1) z.b = x.b; else if constexpr (B2) z.b = -y.b;
    return z;
}
template <bool A1, bool B1, bool A2, bool B2>
GR_DEV DualP<A1 || A2, B1 || B2> operator*(DualP<A1, B1> x, DualP<A2, B2> y)
{
    DualP<A1 || A2, B1 || B2> z{ x.v * y.v, 0.0, 0.0 };
    if constexpr (A1 && A2) z.a = GR_FMA(x.a, y.v, x.v * y.a); else if constexpr (A1) z.a = x.a * y.v; else if constexpr (A2) z.a = x.v * y.a;
    if constexpr (B1 && B2) z.b = GR_FMA(x.b, y.v, x.v * y.b); else if constexpr (B1) z.b = x.b * y.v; else if constexpr (B2) z.b = x.v * y.b;
    return z;
}
template <bool A, bool B> GR_DEV DualP<A, B> operator-(DualP<A, B> x) { return { -x.v, -x.a, -x.b }; }
template <bool A, bool B> GR_DEV DualP<A, B> operator+(DualP<A, B> x, real y) { return { x.v + y, x.a, x.b }; }
template <bool A, bool B> GR_DEV DualP<A, B> operator+(real y, DualP<A, B> x) { return { x.v + y, x.a, x.b }; }
template <bool A, bool B> GR_DEV DualP<A, B> operator-(DualP<A, B> x, real y) { return { x.v - y, x.a, x.b }; }
template <bool A, bool B> GR_DEV DualP<A, B> operator-(real y, DualP<A, B> x) { return { y - x.v, -x.a, -x.b }; }
template <bool A, bool B> GR_DEV DualP<A, B> operator*(real s, DualP<A, B> x) { return { s * x.v, A ? s * x.a : x.a, B ? s * x.b : x.b }; }
template <bool A, bool B> GR_DEV DualP<A, B> operator*(DualP<A, B> x, real s) { return s * x; }
template <bool A, bool B>
GR_DEV DualP<A, B> dinv(DualP<A, B> y)
{
    const real i = rcp_full(y.v);
    const real m = -i * i;
    return { i, A ? m * y.a : y.a, B ? m * y.b : y.b };
}
// what a metric's component function stores: a typed dual widens to the full one (absent partial = 0), a plain value stays
template <bool A, bool B> GR_DEV void dput(Dual2& d, DualP<A, B> x) { d.v = x.v; d.a = A ? x.a : (real)0.0; d.b = B ? x.b : (real)0.0; }
GR_DEV void dput(real& d, real x) { d = x; }

// ---------------------------------------------------------------------------------------
// Metrics.  eval() returns, for the block form (tt, rr, θθ, ϕϕ, tϕ):
//   g[5], gr[5] = ∂_r g, gt[5] = ∂_θ g, gi[5] = inverse components (tt, rr, θθ, ϕϕ, tϕ)
// given r and (sinθ, cosθ).
// ---------------------------------------------------------------------------------------
GR_DEV void inverse_generic(const real g[5], real gi[5])
{
    // inverse_metric_components, auto-diff.jl:59-76, with a single reciprocal
    const real D = GR_FMA(g[0], g[3], -g[4] * g[4]);
    const real rt = g[1] * g[2];
    const real P = rcp_full(D * rt);
    const real iD = P * rt;
    gi[0] = g[3] * iD;
    gi[1] = P * D * g[2];
    gi[2] = P * D * g[1];
    gi[3] = g[0] * iD;
    gi[4] = -g[4] * iD;
}

// Kerr (kerr-metric.jl:11-28) and, with CHARGED, Kerr-Newman (kerr-newman-ad.jl:6-27): the same
// closed forms with Δ -> Δ + Q² and 2Mr -> 2Mr - Q²; g_tt g_ϕϕ - g_tϕ² = -Δ sin²θ holds for both.
template <bool CHARGED>
struct KerrFamily {
    static constexpr bool kHasForce = CHARGED;
    static constexpr bool kFusedRhs = true;                    // rhs() below replaces eval() + the generic contraction
    static constexpr int kMinWavesPerSimd = CHARGED ? 2 : 1;   // persistent kernel: Kerr fits 2 waves/SIMD on its own
    // one-ray-per-lane kernel: capped at 168 VGPRs = 3 waves/SIMD.  The few spilled values (36-68 B of scratch) live in
    // the rarely executed event-sampling blocks, none on the step's main path; the third wave hides the dependent
    // FP64 chains that two waves leave exposed: 22.95 -> 21.97 ms on the 2048² image (profiles/r2_ab_variants.txt)
    // Kerr-Newman (and every metric below) since round 4: three waves as well, made to fit by parking the stage accelerations
    // in LDS (ParkA: 197 -> 163 registers, no scratch)
    static constexpr int kLaneWavesPerSimd = (CHARGED && GR_PARK_DEFAULT == 0) ? 2 : 3;
    static constexpr int kParkStages = CHARGED ? GR_PARK_DEFAULT : 0;
    static constexpr bool kColdRare = !CHARGED;                // at the register cap: the event sampling parks in LDS (gr_kernels.hpp)
    real M, a;
    real Q, Q2, qm;      // CHARGED only: charge, its square, test-particle q (or q/μ)
    real ka2, ktM, kta2; // a², 2M, 2a²: uniform, formed once (rhs)
    GR_DEV void load(const gr_config& c)
    {
        M = c.params[0]; a = c.params[1];
        ka2 = uni(a * a); ktM = uni(2.0 * M); kta2 = uni(2.0 * (a * a));
        Q = Q2 = qm = 0.0;
        if (CHARGED) {
            Q = c.params[2];
            Q2 = Q * Q;
            const double amu = c.mu < 0.0 ? -c.mu : c.mu;
            qm = (real)(amu < 1.4901161193847656e-08 ? c.q : c.q / c.mu);
        }
    }

    // values only (constraint, redshift)
    GR_DEV void comps(real r, real s, real c, real g[5]) const
    {
        const real r2 = r * r, a2 = a * a, s2 = s * s;
        const real Sig = GR_FMA(a2, c * c, r2);
        real Del = GR_FMA(-2.0 * M, r, r2) + a2;
        if (CHARGED) Del += Q2;
        const real iSig = rcp_full(Sig);
        const real w = CHARGED ? (2.0 * M * r - Q2) * iSig : 2.0 * M * r * iSig;
        g[0] = w - 1.0;
        g[1] = Sig * rcp_full(Del);
        g[2] = Sig;
        g[4] = -a * s2 * w;
        g[3] = s2 * (r2 + a2 - a * g[4]);
    }

    // hand-differentiated; one reciprocal for everything
    GR_DEV void eval(real r, real s, real c, real g[5], real gr[5], real gt[5], real gi[5]) const
    {
        const real r2 = r * r, a2 = a * a, s2 = s * s, sc = s * c;
        const real Sig = GR_FMA(a2, c * c, r2);
        real Del = GR_FMA(-2.0 * M, r, r2) + a2;
        if (CHARGED) Del += Q2;
        const real P = rcp_full(Sig * Del * s2);
        const real Ds2 = Del * s2;
        const real iSig = P * Ds2;           // 1/Σ
        const real iDel = P * Sig * s2;      // 1/Δ
        const real iDs = P * Sig;            // 1/(Δ sin²θ)
        const real tM = 2.0 * M;
        const real iSig2 = iSig * iSig;
        real w = tM * r * iSig;              // (2Mr - Q²)/Σ
        real w_r = tM * (Sig - 2.0 * r2) * iSig2;
        if (CHARGED) {
            w -= Q2 * iSig;
            w_r += 2.0 * r * Q2 * iSig2;
        }
        const real Sig_t = -2.0 * a2 * sc;
        const real w_t = -w * Sig_t * iSig;
        const real as2 = a * s2;
        const real tr = 2.0 * r;

        g[0] = w - 1.0;
        g[1] = Sig * iDel;
        g[2] = Sig;
        g[4] = -as2 * w;
        const real B = r2 + a2 - a * g[4];
        g[3] = s2 * B;

        gr[0] = w_r;
        gr[1] = (tr - g[1] * (tr - tM)) * iDel;
        gr[2] = tr;
        gr[4] = -as2 * w_r;
        gr[3] = s2 * (tr - a * gr[4]);

        gt[0] = w_t;
        gt[1] = Sig_t * iDel;
        gt[2] = Sig_t;
        gt[4] = -a * (2.0 * sc * w + s2 * w_t);
        gt[3] = 2.0 * sc * B - as2 * gt[4];

        gi[0] = -B * iDel;
        gi[1] = Del * iSig;
        gi[2] = iSig;
        gi[3] = -g[0] * iDs;
        gi[4] = g[4] * iDs;
    }

    // The whole right-hand side a^μ = -Γ^μ_{κλ} v^κ v^λ in one pass (metric_jacobian + inverse_metric_components +
    // compute_geodesic_equation of auto-diff.jl:59-141,206-226 for this metric), without forming the ten derivative
    // components.  With w = (2Mr - Q²)/Σ every component is g_tt = w - 1, g_tϕ = -a s² w, g_ϕϕ = s²(r² + a² + a² s² w),
    // g_rr = Σ/Δ, g_θθ = Σ, so along the ray (ẋ = d/dλ, s² = sin²θ, (s²)˙ = 2 sc v^θ):
    //   ġ_tt = ẇ,  ġ_tϕ = -a((s²)˙ w + s² ẇ),  ġ_ϕϕ = (s²)˙ B + s²(2r v^r - a ġ_tϕ),   B = r² + a² - a g_tϕ
    //   D_r  = ∂_r g_μν v^μ v^ν = w_r U² + 2r(v_θ² + s² v_ϕ²) + ∂_r g_rr v_r²,          U = v^t - a s² v^ϕ
    //   D_θ  = w_θ U² + 2sc [B v_ϕ² + a w v^ϕ (a s² v^ϕ - 2 v^t) - a²(v_θ² + v_r²/Δ)]
    // and g^tt = -B/Δ, g^tϕ = g_tϕ/(Δ s²), g^ϕϕ = (1 - w)/(Δ s²) (g_tt g_ϕϕ - g_tϕ² = -Δ s²).  The derivative of g_rr
    // never appears on its own either: g^rr ½∂_r g_rr = r/Σ - (r - M)/Δ.  84 FP64 instructions against 95 for eval() + the
    // generic contraction as the compiler leaves it (scripts/kernel_probe.sh); same numbers up to rounding
    // (tests/test_kernel_logic_host.py::test_fused_kerr_rhs_equals_generic_contraction).
    GR_DEV void rhs(real r, real s, real c, real vt, real vr, real vh, real vp,
                    real& at, real& ar, real& ah, real& ap) const
    {
        const real a2 = ka2, tM = ktM;
        const real s2 = s * s, sc = s * c;
        const real ra2 = GR_FMA(r, r, a2);                          // r² + a²: shared by Σ, Δ and B
        const real Sig = GR_FMA(-a2, s2, ra2);                      // r² + a² cos²θ = (r² + a²) - a² sin²θ (no cancellation: Σ >= r²)
        real Del = GR_FMA(-tM, r, ra2);
        if (CHARGED) Del += Q2;
        const real Ds2 = Del * s2;
        const real P = rcp_rhs(Sig * Ds2);
        const real iSig = P * Ds2;           // 1/Σ
        const real iDs = P * Sig;            // 1/(Δ sin²θ)
        const real iDel = iDs * s2;          // 1/Δ
        const real tr = 2.0 * r;
        const real n = CHARGED ? GR_FMA(tM, r, -Q2) : tM * r;       // 2Mr - Q²
        const real w = n * iSig;
        const real wiS = w * iSig;                                  // w/Σ
        const real hw_r = GR_FMA(-r, wiS, M * iSig);                // ½ ∂_r w = (MΣ - r n)/Σ²
        const real mSig_t = kta2 * sc;                              // -∂_θ Σ
        const real w_t = wiS * mSig_t;                              // ∂_θ w = -w ∂_θΣ/Σ
        const real q = a * s2;
        const real U = GR_FMA(-q, vp, vt);                          // v^t - a s² v^ϕ
        const real U2 = U * U;
        const real vr2 = vr * vr, vh2 = vh * vh, vp2 = vp * vp, vrvh = vr * vh;
        // t-ϕ block:  T_t = ġ_tt v^t + ġ_tϕ v^ϕ = ẇ U - (a (s²)˙ w) v^ϕ ,  T_ϕ = ġ_tϕ U + ((s²)˙ B + 2 r s² v^r) v^ϕ
        const real wd = GR_FMA(2.0 * hw_r, vr, w_t * vh);           // ẇ
        const real s2d = (2.0 * sc) * vh;                           // (s²)˙
        const real aw = a * w;
        const real z1 = aw * s2d;
        const real gtp = -(q * w);                                  // g_tϕ
        const real B = GR_FMA(-a, gtp, ra2);
        const real gtpd = -GR_FMA(q, wd, z1);                       // ġ_tϕ
        const real Tt = GR_FMA(wd, U, -(z1 * vp));
        const real Tp = GR_FMA(gtpd, U, GR_FMA(s2d, B, (tr * s2) * vr) * vp);
        // the inverse t-ϕ block is (-B s², g_tϕ; g_tϕ, 1 - w) / (Δ s²): the common factor is applied once per component
#ifdef GR_REAL_IS_FLOAT
        {
            // single precision keeps the inverse components formed first: with the common factor outside, 2.6 x as many rays of
            // the C5 sweep run into dt < dtmin at tolerance 1e-4 (0.75 % against 0.29 %; profiles/r3n_fp32_flag_ab.txt) -- the
            // rounding noise of a float error estimate near the horizon is that sensitive to where the large factor 1/(Δ s²) enters
            const real gitp = gtp * iDs, gipp = ((real)1.0 - w) * iDs, BiD = B * iDel;
            at = GR_FMA(BiD, Tt, -(gitp * Tp));
            ap = -GR_FMA(gitp, Tt, gipp * Tp);
        }
#else
        at = iDs * GR_FMA(B * s2, Tt, -(gtp * Tp));                 // -(g^tt T_t + g^tϕ T_ϕ)
        ap = -(iDs * GR_FMA(gtp, Tt, GR_FMA(-w, Tp, Tp)));          // -(g^tϕ T_t + g^ϕϕ T_ϕ)
#endif
        // r equation: -g^rr (ġ_rr v^r - ½ D_r) with g^rr ½∂_r g_rr = r/Σ - (r - M)/Δ and ∂_θ g_rr = ∂_θΣ/Δ
        // = (1/Σ)(Δ in - ∂_θΣ v^r v^θ - r v_r²) + (r - M) v_r²/Δ: 1/Σ applied once, v_r²/Δ shared with the θ equation
        const real vr2iD = vr2 * iDel;
        const real in = GR_FMA(hw_r, U2, r * GR_FMA(s2, vp2, vh2));
        ar = GR_FMA(iSig, GR_FMA(Del, in, GR_FMA(mSig_t, vrvh, -(r * vr2))), (r - M) * vr2iD);
        // θ equation: -(1/Σ)(2 r v^r v^θ - sc X),  X = a²(v_θ² - v_r²/Δ + w U²/Σ) + B v_ϕ² + a w v^ϕ (a s² v^ϕ - 2 v^t)
        const real W1n = U + vt;                                    // -(a s² v^ϕ - 2 v^t) = U + v^t
        real X = GR_FMA(-(aw * vp), W1n, B * vp2);
        X = GR_FMA(a2, GR_FMA(wiS, U2, vh2 - vr2iD), X);
        ah = iSig * GR_FMA(sc, X, -(tr * vrvh));
        if (CHARGED) {
            real gi[5] = { -(B * iDel), Del * iSig, iSig, ((real)1.0 - w) * iDs, gtp * iDs };      // g^tt, g^rr, g^θθ, g^ϕϕ, g^tϕ
            add_force(r, s, c, gi, vt, vr, vh, vp, at, ar, ah, ap);
        }
    }

    // q F^μ_κ v^κ, F = g⁻¹(∂A - ∂Aᵀ) (tracing/utility.jl:89-99), A = (rQ/Σ)(1, 0, 0, -a sin²θ)
    // (kerr-newman-ad.jl:29-33), added to the acceleration as in kerr-newman-ad.jl:66-100.
    // Hand-differentiated: with p = rQ/Σ, p_r = Q(Σ - 2r²)/Σ², p_θ = 2a² rQ sinθ cosθ/Σ².
    GR_DEV void add_force(real r, real s, real c, const real gi[5], real vt, real vr, real vh, real vp,
                          real& at, real& ar, real& ah, real& ap) const
    {
        if (!CHARGED || qm == 0.0) return;
        const real r2 = r * r, s2 = s * s, sc = s * c;
        const real Sig = GR_FMA(a * a, c * c, r2);
        const real iSig = rcp_full(Sig);
        const real p = r * Q * iSig;
        const real p_r = Q * (Sig - 2.0 * r2) * iSig * iSig;
        const real p_t = 2.0 * a * a * sc * p * iSig;
        const real Ap_r = -a * s2 * p_r;
        const real Ap_t = -a * (2.0 * sc * p + s2 * p_t);
        const real wt = p_r * vr + p_t * vh;
        const real wp = Ap_r * vr + Ap_t * vh;
        const real wr = -(p_r * vt + Ap_r * vp);
        const real wh = -(p_t * vt + Ap_t * vp);
        at += qm * (gi[0] * wt + gi[4] * wp);
        ar += qm * (gi[1] * wr);
        ah += qm * (gi[2] * wh);
        ap += qm * (gi[4] * wt + gi[3] * wp);
    }
};
typedef KerrFamily<false> KerrMetric;
typedef KerrFamily<true> KerrNewmanMetric;

// Johannsen metric with hand-written derivatives (johannsen-ad.jl:12-34): about half the flops of
// the dual-number evaluation.  With N = (r²+a²)A1 - a²A2 sin²θ:
//   g_tt = -Σ T/N², g_ϕϕ = Σ sin²θ P/N², g_tϕ = -a Σ sin²θ Q/N², g_rr = Σ/(Δ A5), g_θθ = Σ
//   T = Δ - a²A2² sin²θ, P = (r²+a²)²A1² - a²Δ sin²θ, Q = (r²+a²)A1A2 - Δ
// and for f = ΣXW (W = 1/N²): ∂f = (∂Σ X + Σ ∂X) W - 2 f ∂N/N.
struct JohannsenMetric {
    static constexpr bool kHasForce = false;
    static constexpr bool kFusedRhs = true;                    // rhs() below replaces eval() + the generic contraction
    static constexpr int kMinWavesPerSimd = 2;
    // one-ray-per-lane kernel: 205 registers and two waves per SIMD left 17 % of the issue slots empty (profiles/r3zz_c4: issue
    // 0.83); with A[1..5] parked in LDS (ParkA, 10 KB per wave) the kernel needs 157 and runs three
    static constexpr int kLaneWavesPerSimd = GR_PARK_DEFAULT > 0 ? 3 : 2;      // (-DGR_PARK_DEFAULT=0: the round-3 shape, for A/B)
    static constexpr int kParkStages = GR_PARK_DEFAULT;
    real M, a, a13, a22, a52, e3;
    real ka2, ktM, keM3;                             // a², 2M, ϵ3 M³: uniform, formed once (rhs)
    real kA1, kA2, kA5, kA1r, kA2r, kA5r;            // α13 M³, α22 M², α52 M² and -3, -2, -2 times them: A_i = 1 + kA_i / r^n
    GR_DEV void load(const gr_config& c)
    {
        M = c.params[0]; a = c.params[1]; a13 = c.params[2]; a22 = c.params[3]; a52 = c.params[4]; e3 = c.params[5];
        ka2 = uni(a * a); ktM = uni(2.0 * M); keM3 = uni(e3 * M * M * M);
        kA1 = uni(a13 * (M * M * M)); kA2 = uni(a22 * (M * M)); kA5 = uni(a52 * (M * M));
        kA1r = uni(-3.0 * (a13 * (M * M * M))); kA2r = uni(-2.0 * (a22 * (M * M))); kA5r = uni(-2.0 * (a52 * (M * M)));
    }

    GR_DEV void comps(real r, real s, real c, real g[5]) const
    {
        real gr[5], gt[5], gi[5];
        eval(r, s, c, g, gr, gt, gi);
    }

    GR_DEV void eval(real r, real s, real c, real g[5], real gr[5], real gt[5], real gi[5]) const
    {
        const real a2 = a * a, s2 = s * s, sc2 = 2.0 * s * c, r2 = r * r;
        const real ir = rcp_full(r);
        const real Mr = M * ir, Mr2 = Mr * Mr, Mr3 = Mr2 * Mr;
        const real A1 = GR_FMA(a13, Mr3, 1.0), A2 = GR_FMA(a22, Mr2, 1.0), A5 = GR_FMA(a52, Mr2, 1.0);
        const real A1r = -3.0 * a13 * Mr3 * ir, A2r = -2.0 * a22 * Mr2 * ir, A5r = -2.0 * a52 * Mr2 * ir;
        const real eM3 = e3 * M * M * M;
        const real Sig = GR_FMA(a2, c * c, r2) + eM3 * ir;
        const real Sig_r = 2.0 * r - eM3 * ir * ir;
        const real Sig_t = -a2 * sc2;
        const real Del = GR_FMA(-2.0 * M, r, r2) + a2;
        const real Del_r = 2.0 * r - 2.0 * M;
        const real P1 = r2 + a2, P1r = 2.0 * r;
        const real a2s2 = a2 * s2;
        const real N = P1 * A1 - a2s2 * A2;
        const real N_r = P1r * A1 + P1 * A1r - a2s2 * A2r;
        const real N_t = -a2 * A2 * sc2;
        const real DA5 = Del * A5;
        // 1/N, 1/(ΔA5), 1/Σ from one reciprocal
        const real NS = N * Sig;
        const real Pinv = rcp_full(NS * DA5);
        const real iN = Pinv * Sig * DA5;
        const real iDA5 = Pinv * NS;
        const real iSig = Pinv * N * DA5;
        const real W = iN * iN;
        const real A2sq = A2 * A2;
        const real T = Del - a2s2 * A2sq;
        const real T_r = Del_r - 2.0 * a2s2 * A2 * A2r;
        const real T_t = -a2 * A2sq * sc2;
        const real P1A1 = P1 * A1;
        const real P = P1A1 * P1A1 - a2s2 * Del;
        const real P_r = 2.0 * P1A1 * (P1r * A1 + P1 * A1r) - a2s2 * Del_r;
        const real P_t = -a2 * Del * sc2;
        const real Q = P1A1 * A2 - Del;
        const real Q_r = (P1r * A1 + P1 * A1r) * A2 + P1A1 * A2r - Del_r;
        const real SW = Sig * W;
        const real tNr = 2.0 * N_r * iN, tNt = 2.0 * N_t * iN;

        g[0] = -SW * T;
        g[1] = Sig * iDA5;
        g[2] = Sig;
        g[3] = SW * s2 * P;
        g[4] = -a * SW * s2 * Q;

        gr[0] = -(Sig_r * T + Sig * T_r) * W - g[0] * tNr;
        gr[1] = (Sig_r - g[1] * (Del_r * A5 + Del * A5r)) * iDA5;
        gr[2] = Sig_r;
        gr[3] = s2 * (Sig_r * P + Sig * P_r) * W - g[3] * tNr;
        gr[4] = -a * s2 * (Sig_r * Q + Sig * Q_r) * W - g[4] * tNr;

        gt[0] = -(Sig_t * T + Sig * T_t) * W - g[0] * tNt;
        gt[1] = Sig_t * iDA5;
        gt[2] = Sig_t;
        gt[3] = ((Sig_t * s2 + Sig * sc2) * P + Sig * s2 * P_t) * W - g[3] * tNt;
        gt[4] = -a * (Sig_t * s2 + Sig * sc2) * Q * W - g[4] * tNt;

        // inverse: 1/g_rr = ΔA5/Σ, 1/g_θθ = 1/Σ, and the t-ϕ block
        const real D2 = GR_FMA(g[0], g[3], -g[4] * g[4]);
        const real iD2 = rcp_full(D2);
        gi[0] = g[3] * iD2;
        gi[1] = DA5 * iSig;
        gi[2] = iSig;
        gi[3] = g[0] * iD2;
        gi[4] = -g[4] * iD2;
    }

    // The whole right-hand side in one pass, as KerrFamily::rhs.  With K = (r²+a²)A1, N = K - a²s²A2, F = Σ/N² (Σ incl.
    // the ϵ3 term) the t-ϕ block is g_tt = -F T, g_tϕ = -a s² F Q, g_ϕϕ = s² F P with T = Δ - a²s²A2², Q = K A2 - Δ,
    // P = K² - a²s²Δ, and T P + a²s²Q² = Δ N², so det = -Σ² s² Δ/N² and the inverse block needs 1/(ΣΔ) only:
    //   g^tt = -P/(ΣΔ), g^tϕ = -aQ/(ΣΔ), g^ϕϕ = T/(ΣΔs²)         (eval() spends a second reciprocal on it)
    // Every θ-derivative carries the factor S2 = 2 sinθ cosθ: ∂_θ(T, s²Q, s²P, N, Σ) = S2 (-a²A2², Q, P - a²s²Δ, -a²A2, -a²).
    // With ℓ_t = -(T v^t + a s²Q v^ϕ), ℓ_ϕ = s²(P v^ϕ - a Q v^t), Φ = ℓ_t v^t + ℓ_ϕ v^ϕ and κ = ∂ ln F = ∂Σ/Σ - 2∂N/N:
    //   T_t = F(κ̇ ℓ_t + ℓ̇_t), T_ϕ = F(κ̇ ℓ_ϕ + ℓ̇_ϕ)  (dots: along (v^r, v^θ)),   D_x = F(κ_x Φ + Φ_x) + ∂_x g_rr v_r² + ∂_xΣ v_θ².
    // One reciprocal for 1/N, 1/(ΔA5), 1/Σ, 1/(Σs²) besides 1/r.  Equal to eval() + the generic contraction to rounding
    // (tests/test_kernel_logic_host.py::test_fused_rhs_equals_generic_contraction).
    GR_DEV void rhs(real r, real s, real c, real vt, real vr, real vh, real vp,
                    real& at, real& ar, real& ah, real& ap) const
    {
        const real a2 = ka2, tM = ktM, eM3 = keM3;
        // A_i = 1 + α (M/r)^n with the powers of M folded into uniform factors: nine instructions for the three functions
        // and their r-derivatives (twelve when (M/r)^n is formed first)
        const real ir = rcp_rhs(r);
        const real ir2 = ir * ir, ir3 = ir2 * ir, ir4 = ir2 * ir2;
        const real A1 = GR_FMA(kA1, ir3, 1.0), A2 = GR_FMA(kA2, ir2, 1.0), A5 = GR_FMA(kA5, ir2, 1.0);
        const real A1r = kA1r * ir4, A2r = kA2r * ir3, A5r = kA5r * ir3;
        const real s2 = s * s, sc = s * c, S2 = 2.0 * sc, tr = 2.0 * r;
        const real eir = eM3 * ir;
        const real rho2 = GR_FMA(r, r, a2);                         // r² + a²: shared by Σ, Δ and K
        const real a2s2 = a2 * s2;
        const real Sig = (rho2 - a2s2) + eir;                       // r² + a² cos²θ + ϵ3 M³/r
        const real Sig_r = GR_FMA(-eir, ir, tr);
        const real Del = GR_FMA(-tM, r, rho2);
        const real Del_r = tr - tM;
        const real K = rho2 * A1;
        const real K_r = GR_FMA(rho2, A1r, tr * A1);
        const real N = GR_FMA(-a2s2, A2, K);
        const real N_r = GR_FMA(-a2s2, A2r, K_r);
        const real DA5 = Del * A5;
        const real DA5_r = GR_FMA(Del, A5r, Del_r * A5);
        // reciprocals
        const real e1 = N * DA5, e2 = Sig * s2;
        const real R = rcp_rhs(e1 * e2);
        const real ie1 = R * e2, ie2 = R * e1;                      // 1/(N ΔA5), 1/(Σ s²)
        const real iN = ie1 * DA5, iDA5 = ie1 * N, iSig = ie2 * s2;
        const real iDel = A5 * iDA5;
        const real c2 = (iN * iN) * iDel;                           // F/(ΣΔ) = 1/(N²Δ)
        const real is2 = ie2 * Sig;                                 // 1/s²
        const real F = Sig * (iN * iN);
        // the three functions of the t-ϕ block and their r-derivatives
        const real aA2 = a2s2 * A2;
        const real T = GR_FMA(-aA2, A2, Del);
        const real T_r = GR_FMA(-2.0 * aA2, A2r, Del_r);
        const real Q = GR_FMA(K, A2, -Del);
        const real Q_r = GR_FMA(K_r, A2, GR_FMA(K, A2r, -Del_r));
        const real a2s2D = a2s2 * Del;
        const real P = GR_FMA(K, K, -a2s2D);
        const real P_r = GR_FMA(2.0 * K, K_r, -(a2s2 * Del_r));
        const real Pth = P - a2s2D;                                 // ∂_θ(s²P)/S2
        const real a2A22 = a2 * (A2 * A2);                          // -∂_θT/S2
        const real aQ = a * Q;
        // linear and quadratic forms
        const real lt = -GR_FMA(T, vt, (aQ * s2) * vp);
        const real lp = s2 * GR_FMA(P, vp, -(aQ * vt));
        const real Phi = GR_FMA(lt, vt, lp * vp);
        const real iN2 = iN + iN;
        const real kr = GR_FMA(-N_r, iN2, Sig_r * iSig);
        const real kth = a2 * GR_FMA(A2, iN2, -iSig);               // κ_θ/S2
        const real vt2 = vt * vt, vp2 = vp * vp, tvtp = (vt + vt) * vp, vr2 = vr * vr, vh2 = vh * vh, vrvh = vr * vh;
        const real aQr_s2 = (a * Q_r) * s2;
        const real Phi_r = GR_FMA(s2 * P_r, vp2, -GR_FMA(T_r, vt2, aQr_s2 * tvtp));
        const real phith = GR_FMA(Pth, vp2, GR_FMA(a2A22, vt2, -(aQ * tvtp)));
        // r equation
        const real grr = Sig * iDA5;
        const real grr_r = GR_FMA(-grr, DA5_r, Sig_r) * iDA5;
        const real a2S2 = a2 * S2;
        const real FDr = F * GR_FMA(kr, Phi, Phi_r);
        real br = GR_FMA(grr_r, vr2, -GR_FMA(Sig_r, vh2, FDr));     // ∂_r g_rr v_r² - Σ_r v_θ² - F(κ_rΦ + Φ_r)
        br = GR_FMA(0.5, br, -((a2S2 * iDA5) * vrvh));
        ar = -((DA5 * iSig) * br);
        // θ equation
        const real FDh = GR_FMA(F, GR_FMA(kth, Phi, phith), a2 * GR_FMA(-iDA5, vr2, vh2));
        ah = -(iSig * GR_FMA(-sc, FDh, Sig_r * vrvh));
        // t and ϕ equations
        const real S2vh = S2 * vh;
        const real Td = GR_FMA(T_r, vr, -(a2A22 * S2vh));
        const real Qsd = GR_FMA(s2 * Q_r, vr, Q * S2vh);
        const real Psd = GR_FMA(s2 * P_r, vr, Pth * S2vh);
        const real kd = GR_FMA(kr, vr, kth * S2vh);
        const real aQsd = a * Qsd;
        const real ltd = -GR_FMA(Td, vt, aQsd * vp);
        const real lpd = GR_FMA(Psd, vp, -(aQsd * vt));
        const real taut = GR_FMA(kd, lt, ltd);
        const real taup = GR_FMA(kd, lp, lpd);
        at = c2 * GR_FMA(P, taut, aQ * taup);
        ap = c2 * GR_FMA(aQ, taut, -((T * is2) * taup));             // g^ϕϕ = T/(ΣΔs²)
    }
};

// Every other AbstractStaticAxisSymmetric metric: components written once over a number type and
// differentiated with forward-mode duals, as the reference does for all of its metrics.  The functor is a template on
// the metric id, so each metric gets its own kernels holding only its own component function (one translation unit
// per metric, kernels_tu.hip).  Round 1 had ONE functor with a wave-uniform switch: every one of the six RHS sites of a
// step then carried all nine metric bodies -- a 9 600-instruction step loop (57 KB of code against a 64 KB
// instruction cache), 180 B of scratch per lane and 1.8x the time of Kerr.  ID < 0 keeps the run-time switch
// (tests/host_harness.cpp traces every metric through one instantiation).
#ifndef GR_JP_LANE_WAVES
#define GR_JP_LANE_WAVES (GR_PARK_DEFAULT > 0 ? 3 : 2)
#endif
#ifndef GR_GENERIC_LANE_WAVES
#define GR_GENERIC_LANE_WAVES (GR_PARK_DEFAULT > 0 ? 3 : 2)
#endif
#ifndef GR_FUSED23_LANE_WAVES
#define GR_FUSED23_LANE_WAVES 3
#endif
// The fused dilaton-axion form shares one reciprocal of Σh Δh s² K² between 1/Σh, 1/Δh and the inverse t-ϕ block: in single precision
// that costs rays near the horizon (the fp32 soak flags 1089 rays instead of 875 with it, profiles/r4z_soak32_2600_fp32_fused_dilaton_axion.txt) -- the fp32
// kernels keep the dual-number form for this metric, as they keep the inverse components formed first for Kerr -- and for NoZ, whose
// fused form is built the same way (one reciprocal of D Δ Se s²) and was not tried in single precision
#ifdef GR_REAL_IS_FLOAT
#define GR_DA_FUSED false
#else
#define GR_DA_FUSED true
#endif
template <int ID>
struct GenericMetricT {
    static constexpr int kMinWavesPerSimd = GR_GENERIC_MIN_WAVES;
    // one-ray-per-lane kernel: three waves per SIMD with the stage accelerations parked in LDS (181-228 registers without,
    // 142-168 with, no scratch: scripts/kernel_probe.sh "GenericMetricT<id>")
    // Bumblebee and Morris-Thorne with their fused right-hand sides need 167 registers (Bumblebee: with the event sampling's
    // registers parked in LDS like Kerr's, kColdRare): three waves per SIMD (GR_FUSED23_LANE_WAVES=2: the two-wave shape, A/B)
    // (Kerr-dark-matter and Kerr-refractive, fused later in round 4, need 193: at three waves they spill 80-96 bytes and run 6.3 /
    // 8.6 ms against 5.6 / 7.8 at 1024², 19.9 / 26.4 against 20.2 / 26.2 at 2048² -- two waves)
    static constexpr bool kSlimFused = (ID == GR_METRIC_BUMBLEBEE || ID == GR_METRIC_MORRIS_THORNE);
    static constexpr int kLaneWavesPerSimd = kSlimFused ? GR_FUSED23_LANE_WAVES
                                             : (ID == GR_METRIC_JOHANNSEN_PSALTIS) ? GR_JP_LANE_WAVES : GR_GENERIC_LANE_WAVES;
    static constexpr bool kColdRare = (ID == GR_METRIC_BUMBLEBEE) && kLaneWavesPerSimd >= 3;
    static constexpr int kParkStages = (!kSlimFused && kLaneWavesPerSimd >= 3) ? GR_PARK_DEFAULT : 0;
    static constexpr bool kHasForce = false;
    // rhs() below: hand-derived for Johannsen-Psaltis (round 3), for Bumblebee, Morris-Thorne, Kerr-dark-matter and Kerr-refractive
    // (round 4; the last two as Kerr plus the terms of their r-dependent parameter), flat space, dilaton-axion and NoZ: every metric
    // of the catalogue.  eval() + the generic contraction on typed duals remains as the definition the fused forms are tested
    // against (tests/test_kernel_logic_host.py), as the fp32 kernels' form for dilaton-axion and NoZ, and under GR_NO_FUSED_RHS
    static constexpr bool kFusedRhs = (ID == GR_METRIC_JOHANNSEN_PSALTIS || ID == GR_METRIC_BUMBLEBEE || ID == GR_METRIC_MORRIS_THORNE
                                       || ID == GR_METRIC_KERR_DARK_MATTER || ID == GR_METRIC_KERR_REFRACTIVE || ID == GR_METRIC_SPHERICAL
                                       || (ID == GR_METRIC_DILATON_AXION && GR_DA_FUSED) || (ID == GR_METRIC_NOZ && GR_DA_FUSED));
    int32_t id;
    real P[6];
    real ka2, ktM, keps;      // Johannsen-Psaltis rhs(): a², 2M, ϵ3 M³ -- uniform, formed once
    real kik;                 // Bumblebee rhs(): 1/(1 + l)
    real kin0;                // Kerr-refractive rhs(): 1/n
    real kd_bab, kd_N0, kd_c1, kd_c2, kd_tk2a;      // dilaton-axion rhs(): uniform combinations of (M, a, β, b), see there
    real kn_eMa;              // NoZ rhs(): ϵ M a
    GR_DEV void load(const gr_config& c)
    {
        id = ID >= 0 ? ID : c.metric_id;
#pragma unroll
        for (int i = 0; i < 6; ++i) P[i] = c.params[i];
        ka2 = uni(P[1] * P[1]); ktM = uni(2.0 * P[0]); keps = uni(P[2] * P[0] * P[0] * P[0]);
        kik = (ID == GR_METRIC_BUMBLEBEE) ? uni(rcp_full(1.0 + P[2])) : (real)0.0;
        kin0 = (ID == GR_METRIC_KERR_REFRACTIVE) ? uni(rcp_full(P[2])) : (real)0.0;
        kn_eMa = (ID == GR_METRIC_NOZ) ? uni(P[2] * P[0] * P[1]) : (real)0.0;
        kd_bab = kd_N0 = kd_c1 = kd_c2 = kd_tk2a = 0.0;
        if constexpr (ID == GR_METRIC_DILATON_AXION) {
            const real M = P[0], a = P[1], be = P[2], b = P[3];
            const bool z = (be == 0.0);
            const real bb = z ? 0.0 : be * rcp_full(b), ba = z ? 0.0 : be * rcp_full(a), bab = z ? 0.0 : be * rcp_full(a * b);
            const real k1 = M * (M + 2.0 * b) * bb * bb, k2 = M * M * bb;
            kd_bab = uni(bab);
            kd_N0 = uni(ba * ba - bab * bab);
            kd_c1 = uni(a * a - be * be - k1);
            kd_c2 = uni(a * a - be * be + k2 * bb);
            kd_tk2a = uni(2.0 * k2 * a);
        }
    }
    static GR_DEV real inv_(real x) { return rcp_full(x); }
    template <bool A, bool B> static GR_DEV DualP<A, B> inv_(DualP<A, B> x) { return dinv(x); }
    static GR_DEV real val_(real x) { return x; }
    template <bool A, bool B> static GR_DEV real val_(DualP<A, B> x) { return x.v; }
    static GR_DEV real atan_(real x) { return GR_ATAN(x); }
    template <bool A, bool B> static GR_DEV DualP<A, B> atan_(DualP<A, B> x)
    {
        const real w = rcp_full(GR_FMA(x.v, x.v, 1.0));
        return { GR_ATAN(x.v), A ? w * x.a : x.a, B ? w * x.b : x.b };
    }

    // __JohannsenAD.metric_components, johannsen-ad.jl:12-34 ; P = M, a, α13, α22, α52, ϵ3
    template <class TR, class TT, class TG>
    GR_DEV void johannsen(TR r, TT s, TT c, TG g[5]) const
    {
        const real M = P[0], a = P[1], a13 = P[2], a22 = P[3], a52 = P[4], e3 = P[5];
        const real a2 = a * a;
        auto Mr = M * inv_(r);
        auto Mr2 = Mr * Mr;
        auto A1 = 1.0 + a13 * (Mr2 * Mr);
        auto A2 = 1.0 + a22 * Mr2;
        auto A5 = 1.0 + a52 * Mr2;
        auto r2 = r * r;
        auto Sig = r2 + a2 * (c * c) + (e3 * M * M * M) * inv_(r);
        auto Del = r2 - (2.0 * M) * r + a2;
        auto r2a2 = r2 + a2;
        auto s2 = s * s;
        auto dn = r2a2 * A1 - a2 * (A2 * s2);
        auto idenom = inv_(dn * dn);
        auto tt = -(Sig * (Del - a2 * (A2 * A2 * s2)));
        auto pp = (Sig * s2) * ((r2a2 * r2a2) * (A1 * A1) - a2 * (Del * s2));
        auto tp = -(a * ((Sig * s2) * (r2a2 * A1 * A2 - Del)));
        dput(g[0], tt * idenom);
        dput(g[1], Sig * inv_(Del * A5));
        dput(g[2], Sig);
        dput(g[3], pp * idenom);
        dput(g[4], tp * idenom);
    }
    // __MorrisThorneAD.metric_components, morris-thorne-ad.jl:4-15 ; P = b.  (ϕϕ carries sinθ to
    // the FIRST power in the reference; reproduced as is.)
    template <class TR, class TT, class TG>
    GR_DEV void morris_thorne(TR l, TT s, TT c, TG g[5]) const
    {
        const real b2 = P[0] * P[0];
        auto w = l * l + b2;
        dput(g[0], (l - l) - 1.0);
        dput(g[1], (l - l) + 1.0);
        dput(g[2], w);
        dput(g[3], w * s);
        dput(g[4], l - l);
        (void)c;
    }
    // __BumblebeeAD.metric_components, bumblebee-ad.jl:6-21 ; P = M, a, l
    template <class TR, class TT, class TG>
    GR_DEV void bumblebee(TR r, TT s, TT c, TG g[5]) const
    {
        const real M = P[0], a = P[1], lsb = P[2];
        auto s2 = s * s;
        auto r2 = r * r;
        auto ir = inv_(r);
        auto Del = (r2 - (2.0 * M) * r) * (1.0 / (lsb + 1.0));
        dput(g[0], -(1.0 - (2.0 * M) * ir));
        dput(g[1], r2 * inv_(Del));
        dput(g[2], r2);
        dput(g[3], r2 * s2);
        dput(g[4], -((2.0 * M * a) * (s2 * ir)));
        (void)c;
    }
    // __JohannsenPsaltisAD.metric_components, johannsen-psaltis-ad.jl:4-27 ; P = M, a, ϵ3
    template <class TR, class TT, class TG>
    GR_DEV void johannsen_psaltis(TR r, TT s, TT c, TG g[5]) const
    {
        const real M = P[0], a = P[1], e3 = P[2];
        const real a2 = a * a;
        auto r2 = r * r;
        auto Sig = r2 + a2 * (c * c);
        auto iSig = inv_(Sig);
        auto h = (e3 * M * M * M) * (r * (iSig * iSig));
        auto s2 = s * s;
        auto Del = r2 - (2.0 * M) * r + a2;
        auto tMr = (2.0 * M) * r;
        auto hp1 = 1.0 + h;
        dput(g[0], -(hp1 * (1.0 - tMr * iSig)));
        dput(g[1], (Sig * hp1) * inv_(Del + a2 * (s2 * h)));
        dput(g[2], Sig);
        auto term1 = s2 * (r2 + a2 + (a2 * (tMr * s2)) * iSig);
        auto term2 = (h * a2) * ((Sig + tMr) * ((s2 * s2) * iSig));
        dput(g[3], term1 + term2);
        dput(g[4], -((a * tMr) * ((s2 * hp1) * iSig)));
    }

    // __DilatonAxionAD.metric_components, dilaton-axion-ad.jl:8-46 ; P = M, a, β, b
    template <class TR, class TT, class TG>
    GR_DEV void dilaton_axion(TR r, TT s, TT c, TG g[5]) const
    {
        const real M = P[0], a = P[1], be = P[2], b = P[3];
        const real R = M, a2 = a * a;
        const bool z = (be == 0.0);
        const real bb = z ? 0.0 : be * rcp_full(b), ba = z ? 0.0 : be * rcp_full(a), bab = z ? 0.0 : be * rcp_full(a * b);
        auto s2 = s * s;
        auto r2 = r * r;
        auto Sig = r2 + a2 * (c * c);
        auto Del = r2 + a2 - (2.0 * R) * r;
        auto bt = (2.0 * b) * r + be * be;
        auto Delh = Del - bt - (R * (R + 2.0 * b) * bb * bb);
        auto Sigh = Sig - bt + (R * R * bb) * (bb - (2.0 * a) * c);
        auto del = r2 - (2.0 * b) * r + a2;
        auto W = 1.0 + (bab * (2.0 * c - bab) + ba * ba) * inv_(s2);
        auto Was = W * (a * s);
        auto A = del * del - Delh * (Was * Was);
        auto iSigh = inv_(Sigh);
        dput(g[0], -((Delh - a2 * s2) * iSigh));
        dput(g[1], Sigh * inv_(Delh));
        dput(g[2], Sigh);
        dput(g[3], (A * s2) * iSigh);
        dput(g[4], -((a * (del - Delh * W)) * (s2 * iSigh)));
    }

    // SphericalMetric (flat space in spherical coordinates), minkowski.jl:4-13
    template <class TR, class TT, class TG>
    GR_DEV void spherical(TR r, TT s, TT c, TG g[5]) const
    {
        auto r2 = r * r;
        dput(g[0], (r - r) - 1.0);
        dput(g[1], (r - r) + 1.0);
        dput(g[2], r2);
        dput(g[3], r2 * (s * s));
        dput(g[4], r - r);
        (void)c;
    }
    // __KerrDarkMatter.metric_components, kerr-dark-matter.jl:6-49 ; P = M_bh, a, M_dm, Δr, rₛ: Kerr with
    // the mass M_bh + M_dm G((r - rₛ)/Δr), G(x) = (3 - 2x) x², switched on between rₛ and rₛ + Δr
    template <class TR, class TT, class TG>
    GR_DEV void kerr_dark_matter(TR r, TT s, TT c, TG g[5]) const
    {
        const real Mbh = P[0], a = P[1], Mdm = P[2], dR = P[3], rs = P[4];
        const real a2 = a * a;
        const real rv = val_(r);
        auto M = (r - r) + Mbh;
        if (rv >= rs + dR) {
            M = M + Mdm;
        } else if (rv >= rs) {
            auto dr = (r - rs) * rcp_full(dR);
            M = M + Mdm * ((3.0 - 2.0 * dr) * (dr * dr));
        }
        auto R = 2.0 * M;
        auto s2 = s * s;
        auto r2 = r * r;
        auto Sig = r2 + a2 * (1.0 - s2);
        auto iSig = inv_(Sig);
        auto Rr = R * r;
        dput(g[0], -(1.0 - Rr * iSig));
        dput(g[1], Sig * inv_(r2 + a2 - Rr));
        dput(g[2], Sig);
        dput(g[3], s2 * (r2 + a2 + (a2 * (s2 * Rr)) * iSig));
        dput(g[4], -((a * (Rr * s2)) * iSig));
        (void)c;
    }
    // __KerrRefractiveAD.metric_components, kerr-refractive-ad.jl:8-33 ; P = M, a, n, corona_radius: Kerr
    // with tt / n², tϕ / n inside the corona; the boundary is the smooth step of utils.jl:158-168
    // (δx = 2.5, atan(1e4 t)/π), whose gradient the rays must see
    template <class TR, class TT, class TG>
    GR_DEV void kerr_refractive(TR r, TT s, TT c, TG g[5]) const
    {
        const real M = P[0], a = P[1], n0 = P[2], rc = P[3];
        const real a2 = a * a, R = 2.0 * M;
        auto r2 = r * r;
        auto Sig = r2 + a2 * (c * c);
        auto iSig = inv_(Sig);
        auto s2 = s * s;
        auto Rr = R * r;
        const real rv = val_(r);
        auto t = (r - r) + ((rv <= rc - 1.25) ? 1.0 : 0.0);
        if (rv > rc - 1.25 && rv <= rc + 1.25)
            t = 0.5 - 0.3183098861837907 * atan_(1e4 * ((r - rc) * 0.4));
        auto n = t + n0 * (1.0 - t);
        auto in = inv_(n);
        dput(g[0], -(1.0 - Rr * iSig) * (in * in));
        dput(g[1], Sig * inv_(r2 - Rr + a2));
        dput(g[2], Sig);
        dput(g[3], s2 * (r2 + a2 + (a2 * (s2 * Rr)) * iSig));
        dput(g[4], -((a * (Rr * s2)) * iSig) * in);
    }
    // __NoZMetric.metric_components, noz-metric.jl:7-47 ; P = M, a, ϵ (y = cosθ; the θθ component carries
    // the dy² = sin²θ dθ² factor as written there)
    template <class TR, class TT, class TG>
    GR_DEV void noz(TR r, TT s, TT y, TG g[5]) const
    {
        const real M = P[0], a = P[1], e = P[2];
        const real a2 = a * a;
        auto s2 = s * s;
        auto y2 = y * y;
        auto eps = (e * M * a) * y;
        auto r2 = r * r;
        auto a2y2 = a2 * y2;
        auto S = r2 + a2y2;
        auto tMr = (2.0 * M) * r;
        auto iD = inv_(S * S + (r2 - tMr + a2y2) * eps);
        auto Se = S + eps;
        auto omy2 = 1.0 - y2;
        auto big = r2 * r2 + (a2 * a2) * y2 + r2 * (a2 + a2y2 + eps) + a2 * eps + tMr * (a2 - a2y2 - eps);
        dput(g[0], (tMr * S) * iD - 1.0);
        dput(g[1], Se * inv_(r2 - tMr + a2));
        dput(g[2], (Se * inv_(omy2)) * s2);
        dput(g[3], ((omy2 * Se) * big) * iD);
        dput(g[4], -(((a * tMr) * (omy2 * Se)) * iD));
    }

    template <class TR, class TT, class TG>
    GR_DEV void components(TR r, TT s, TT c, TG g[5]) const
    {
        if constexpr (ID == GR_METRIC_SPHERICAL) spherical(r, s, c, g);
        else if constexpr (ID == GR_METRIC_KERR_DARK_MATTER) kerr_dark_matter(r, s, c, g);
        else if constexpr (ID == GR_METRIC_KERR_REFRACTIVE) kerr_refractive(r, s, c, g);
        else if constexpr (ID == GR_METRIC_NOZ) noz(r, s, c, g);
        else if constexpr (ID == GR_METRIC_DILATON_AXION) dilaton_axion(r, s, c, g);
        else if constexpr (ID == GR_METRIC_MORRIS_THORNE) morris_thorne(r, s, c, g);
        else if constexpr (ID == GR_METRIC_BUMBLEBEE) bumblebee(r, s, c, g);
        else if constexpr (ID == GR_METRIC_JOHANNSEN_PSALTIS) johannsen_psaltis(r, s, c, g);
        else if constexpr (ID == GR_METRIC_JOHANNSEN) johannsen(r, s, c, g);
        else {
            switch (id) {
            case GR_METRIC_SPHERICAL: spherical(r, s, c, g); break;
            case GR_METRIC_KERR_DARK_MATTER: kerr_dark_matter(r, s, c, g); break;
            case GR_METRIC_KERR_REFRACTIVE: kerr_refractive(r, s, c, g); break;
            case GR_METRIC_NOZ: noz(r, s, c, g); break;
            case GR_METRIC_DILATON_AXION: dilaton_axion(r, s, c, g); break;
            case GR_METRIC_MORRIS_THORNE: morris_thorne(r, s, c, g); break;
            case GR_METRIC_BUMBLEBEE: bumblebee(r, s, c, g); break;
            case GR_METRIC_JOHANNSEN_PSALTIS: johannsen_psaltis(r, s, c, g); break;
            default: johannsen(r, s, c, g); break;
            }
        }
    }

    GR_DEV void comps(real r, real s, real c, real g[5]) const { components(r, s, c, g); }

    // Johannsen-Psaltis: the whole right-hand side in one pass instead of dual numbers + the generic contraction (the
    // form KerrFamily::rhs and JohannsenMetric::rhs take).  With w = 2Mr/Σ, h = ϵ3 M³ r/Σ², H = 1 + h, η = hΣ:
    //   g_tt = H(w - 1), g_tϕ = -H a s² w, g_ϕϕ = H g_ϕϕ^Kerr - η s², g_rr = ΣH/Δ̃, g_θθ = Σ,   Δ̃ = Δ + a² s² h,
    // the t-ϕ block has determinant -H s² Δ̃ (the Kerr identity (w-1)g_ϕϕ^K - a²s⁴w² = -s²Δ and Σ - 2Mr = Δ - a²s²), so
    //   a^t = [(ρ² + a²s²w - η/H) T_t + a w T_ϕ]/Δ̃,  a^ϕ = [(w - 1) T_ϕ + a s² w T_t]/(s² Δ̃),         ρ² = r² + a².
    // With U = v^t - a s² v^ϕ, ℓ = wU - v^t, ℓ_ϕ^K = s²(ρ² v^ϕ - a w U), Φ^K = wU² - (v^t)² + s²ρ²(v^ϕ)² (dots: along
    // (v^r, v^θ) at fixed velocity; S2 = ∂_θ s² = 2 sinθ cosθ):
    //   T_t = ḣ ℓ + H(ẇ U - a w (s²)˙ v^ϕ),   T_ϕ = ḣ ℓ_ϕ^K + H ℓ̇_ϕ^K - (η s²)˙ v^ϕ,
    //   Φ_x = h_x Φ^K + H Φ^K_x - (η s²)_x (v^ϕ)²,   L_x = ∂_x ln g_rr = Σ_x/Σ + h_x/H - Δ̃_x/Δ̃,
    //   a^r = (Δ̃/(ΣH)) ½(Φ_r + 2r v_θ²) - ½ L_r v_r² - L_θ v_r v_θ,   a^θ = [½Φ_θ + ½ g_rr L_θ v_r² + ½ a² S2 v_θ² - 2r v_r v_θ]/Σ.
    // Two reciprocals (1/Σ; 1/(H Δ̃ s²) shared by 1/H, 1/Δ̃, 1/(s²Δ̃)).  Equal to eval() + the generic contraction to
    // rounding (tests/test_kernel_logic_host.py::test_fused_johannsen_psaltis_rhs_equals_generic_contraction).
    //
    // Morris-Thorne (morris-thorne-ad.jl:4-15; coordinate l, w = l² + b²; g_ϕϕ = w sinθ to the FIRST power as in the
    // reference): g_tt = -1, g_rr = 1, g_θθ = w, g_ϕϕ = w s, no t-ϕ term, so
    //   a^t = 0,  a^l = l (v_θ² + s v_ϕ²),  a^θ = -2 l v_l v_θ / w + ½ c v_ϕ²,  a^ϕ = -(2 l v_l / w + (c/s) v_θ) v_ϕ.
    // One reciprocal, 1/(w s), shared by 1/w and 1/s.
    //
    // Bumblebee (bumblebee-ad.jl:6-21; k = 1 + l, u = 2M/r, K = 2Ma): g_tt = u - 1, g_rr = k r/(r - 2M), g_θθ = r²,
    // g_ϕϕ = r² s², g_tϕ = -K s²/r.  The t-ϕ block has determinant s² D', D' = (u - 1) r² - K² s²/r², so
    //   a^t = -(r² T_t + (K/r) T_ϕ)/D',   a^ϕ = -((K/r) T_t + ((u - 1)/s²) T_ϕ)/D',
    //   T_t = ġ_tt v^t + ġ_tϕ v^ϕ,  T_ϕ = ġ_tϕ v^t + ġ_ϕϕ v^ϕ,  ġ_tt = -(u/r) v_r,  ġ_tϕ = (K s²/r²) v_r - (K S2/r) v_θ,
    //   ġ_ϕϕ = 2 r s² v_r + r² S2 v_θ   (S2 = 2 sinθ cosθ),
    //   a^r = ½ [2M v_r²/(r (r - 2M)) + ((r - 2M)/(k r)) (g_tt,r v_t² + 2 r v_θ² + 2 r s² v_ϕ² + 2 g_tϕ,r v_t v_ϕ)],
    //   a^θ = -2 v_r v_θ / r + ½ (r² S2 v_ϕ² - 2 (K S2 / r) v_t v_ϕ)/r².
    // Two reciprocals: 1/(r (r - 2M)) shared by 1/r and 1/(r - 2M); 1/(D' s²) shared by 1/D' and 1/(s² D').
    // All three equal eval() + the generic contraction to rounding (tests/test_kernel_logic_host.py).
    // The fused Kerr right-hand side (KerrFamily<false>::rhs, which see) with the mass M given per lane and, for the
    // enclosed-mass metric, the mass gradient dM = M'(r) (its terms: at the Kerr-dark-matter branch of rhs() below).  Leaves
    // the pieces its callers build their own terms from.
    struct KerrMid { real iSig, iDel, iDs, Del, w, gtp, B; };
    GR_DEV void kerr_core(real M, real dM, real r, real s, real c, real vt, real vr, real vh, real vp,
                          real& at, real& ar, real& ah, real& ap, KerrMid& k) const
    {
        const real a = P[1];
        const real a2 = ka2, tM = 2.0 * M;
        const real s2 = s * s, sc = s * c;
        const real ra2 = GR_FMA(r, r, a2);
        const real Sig = GR_FMA(-a2, s2, ra2);
        const real Del = GR_FMA(-tM, r, ra2);
        const real Ds2 = Del * s2;
        const real Pr = rcp_rhs(Sig * Ds2);
        const real iSig = Pr * Ds2, iDs = Pr * Sig, iDel = iDs * s2;
        const real tr = 2.0 * r;
        const real w = (tM * r) * iSig;
        const real wiS = w * iSig;
        const real hw_r = GR_FMA(-r, wiS, M * iSig);
        const real mSig_t = (2.0 * a2) * sc;
        const real w_t = wiS * mSig_t;
        const real q = a * s2;
        const real U = GR_FMA(-q, vp, vt);
        const real U2 = U * U;
        const real vr2 = vr * vr, vh2 = vh * vh, vp2 = vp * vp, vrvh = vr * vh;
        const real wd = GR_FMA(2.0 * hw_r, vr, w_t * vh);
        const real s2d = (2.0 * sc) * vh;
        const real aw = a * w;
        const real z1 = aw * s2d;
        const real gtp = -(q * w);
        const real B = GR_FMA(-a, gtp, ra2);
        const real gtpd = -GR_FMA(q, wd, z1);
        real Tt = GR_FMA(wd, U, -(z1 * vp));
        real Tp = GR_FMA(gtpd, U, GR_FMA(s2d, B, (tr * s2) * vr) * vp);
        const real vr2iD = vr2 * iDel;
        const real in = GR_FMA(hw_r, U2, r * GR_FMA(s2, vp2, vh2));
        ar = GR_FMA(iSig, GR_FMA(Del, in, GR_FMA(mSig_t, vrvh, -(r * vr2))), (r - M) * vr2iD);
        if (dM != 0.0) {
            const real dTt = (dM * (tr * iSig)) * (U * vr);
            Tt += dTt;
            Tp = GR_FMA(-q, dTt, Tp);
            ar = GR_FMA(-(dM * r), GR_FMA(-(Del * iSig), iSig * U2, vr2iD), ar);
        }
        at = iDs * GR_FMA(B * s2, Tt, -(gtp * Tp));
        ap = -(iDs * GR_FMA(gtp, Tt, GR_FMA(-w, Tp, Tp)));
        const real W1n = U + vt;
        real X = GR_FMA(-(aw * vp), W1n, B * vp2);
        X = GR_FMA(a2, GR_FMA(wiS, U2, vh2 - vr2iD), X);
        ah = iSig * GR_FMA(sc, X, -(tr * vrvh));
        k.iSig = iSig; k.iDel = iDel; k.iDs = iDs; k.Del = Del; k.w = w; k.gtp = gtp; k.B = B;
    }

    GR_DEV void rhs(real r, real s, real c, real vt, real vr, real vh, real vp,
                    real& at, real& ar, real& ah, real& ap) const
    {
        if constexpr (ID == GR_METRIC_MORRIS_THORNE) {
            const real l = r;
            const real w = GR_FMA(l, l, P[0] * P[0]);
            const real R = rcp_rhs(w * s);                   // 1/(w s)
            const real iw = R * s, is = R * w;
            const real vp2 = vp * vp;
            const real tl = 2.0 * l;
            at = 0.0;
            ar = l * GR_FMA(s, vp2, vh * vh);
            ah = GR_FMA(-(tl * iw), vr * vh, (0.5 * c) * vp2);
            ap = -(GR_FMA(tl * iw, vr, (c * is) * vh) * vp);
            return;
        } else if constexpr (ID == GR_METRIC_SPHERICAL) {
            // flat space in spherical coordinates (minkowski.jl:4-13): g = diag(-1, 1, r², r² s²), so
            //   a^t = 0,  a^r = r (v_θ² + s² v_ϕ²),  a^θ = -2 v_r v_θ / r + s c v_ϕ²,  a^ϕ = -2 (v_r / r + (c/s) v_θ) v_ϕ.
            // One reciprocal, 1/(r s), shared by 1/r and 1/s.
            const real R = rcp_rhs(r * s);
            const real ir = R * s, is = R * r;
            const real vp2 = vp * vp;
            at = 0.0;
            ar = r * GR_FMA(s * s, vp2, vh * vh);
            ah = GR_FMA(-(2.0 * ir), vr * vh, (s * c) * vp2);
            ap = -(2.0 * GR_FMA(ir, vr, (c * is) * vh) * vp);
            return;
        } else if constexpr (ID == GR_METRIC_DILATON_AXION) {
            // Dilaton-axion (dilaton-axion-ad.jl:8-46).  With Δh(r), del(r) = r² - 2br + a², W(θ) = 1 + (β/(ab) (2c - β/(ab)) + (β/a)²)/s²
            // and Σh(r, θ) the components are g_tt = -P/Σh, g_tϕ = -Q/Σh, g_ϕϕ = Φ/Σh, g_rr = Σh/Δh, g_θθ = Σh with
            //   P = Δh - a² s²,   Q = a s² (del - Δh W),   Φ = s² (del² - Δh a² W² s²),
            // and the t-ϕ block has the determinant -Δh s² K²/Σh², K = del - a² s² W (a perfect square, as for Kerr where W = 1).  So with
            // J = 1/(Δh s² K²), H_t = -(Ṗ v^t + Q̇ v^ϕ), H_ϕ = Φ̇ v^ϕ - Q̇ v^t (dots: along (v^r, v^θ)) and p_t, p_ϕ dropping out
            // (g^tμ p_μ = v^t):
            //   a^t = J (Φ H_t + Q H_ϕ) + (Σ̇h/Σh) v^t,     a^ϕ = J (Q H_t - P H_ϕ) + (Σ̇h/Σh) v^ϕ,
            //   a^r = -(1/Σh)[½Σh_r v_r² + Σh_θ v_r v_θ - ½ Σh Δh_r v_r²/Δh] + ½ (Δh/Σh²)(B_r - Σh_r B/Σh) + ½ Σh_r Δh v_θ²/Σh,
            //   a^θ = -(1/Σh)[Σh_r v_r v_θ + ½ Σh_θ (v_θ² - v_r²/Δh) - ½ (B_θ - Σh_θ B/Σh)/Σh],
            // B = -P v_t² - 2Q v_t v_ϕ + Φ v_ϕ² and B_r, B_θ the same form on the partials.  Two reciprocals (1/s² for W; 1/(Σh Δh s² K²)
            // shared by 1/Σh, 1/Δh and J) where the dual-number form takes four.  Equal to eval() + the generic contraction to rounding
            // (tests/test_kernel_logic_host.py).
            const real a = P[1], b = P[3];
            const real a2 = ka2;
            const real s2 = s * s, S2 = 2.0 * (s * c);
            const real is2 = rcp_rhs(s2);
            const real Wm1 = GR_FMA(2.0 * kd_bab, c, kd_N0) * is2;          // W - 1
            const real W = 1.0 + Wm1;
            const real W_h = -(GR_FMA(2.0 * kd_bab, s, Wm1 * S2) * is2);    // ∂_θ W
            const real r2 = r * r;
            const real rb = GR_FMA(-2.0 * b, r, r2);                        // r² - 2br
            const real del = rb + a2;
            const real Dh = GR_FMA(-ktM, r, rb) + kd_c1;                    // Δh
            const real Sh = GR_FMA(-a2, s2, rb) + GR_FMA(-kd_tk2a, c, kd_c2);   // Σh
            const real rp = 2.0 * (r - b);                                  // del' = ∂_r Σh
            const real Dh_r = rp - ktM;
            const real Sh_h = GR_FMA(kd_tk2a, s, -(a2 * S2));               // ∂_θ Σh
            const real aW = a * W;
            const real G = (aW * aW) * s2;                                  // a² W² s²
            const real E = GR_FMA(-Dh, W, del);                             // del - Δh W
            const real K = GR_FMA(-(a2 * s2), W, del);
            const real Pq = GR_FMA(-a2, s2, Dh);
            const real as2 = a * s2;
            const real Q = as2 * E;
            const real F = GR_FMA(-Dh, G, del * del);
            const real Phi = s2 * F;
            const real P_h = -(a2 * S2);
            const real Q_r = as2 * GR_FMA(-Dh_r, W, rp);
            const real Q_h = a * GR_FMA(S2, E, -(s2 * (Dh * W_h)));
            const real Phi_r = s2 * GR_FMA(2.0 * del, rp, -(Dh_r * G));
            const real G_h = a2 * (W * GR_FMA(2.0 * W_h, s2, W * S2));      // ∂_θ (a² W² s²)
            const real Phi_h = GR_FMA(S2, F, -(s2 * (Dh * G_h)));
            const real K2 = K * K;
            const real DsK = (Dh * s2) * K2;
            const real R = rcp_rhs(Sh * DsK);
            const real iSh = R * DsK, J = R * Sh, iDh = J * (s2 * K2);
            const real Pd = GR_FMA(Dh_r, vr, P_h * vh), Qd = GR_FMA(Q_r, vr, Q_h * vh), Phd = GR_FMA(Phi_r, vr, Phi_h * vh);
            const real Shd = GR_FMA(rp, vr, Sh_h * vh);
            const real Ht = -GR_FMA(Pd, vt, Qd * vp);
            const real Hp = GR_FMA(Phd, vp, -(Qd * vt));
            const real sl = Shd * iSh;
            at = GR_FMA(J, GR_FMA(Phi, Ht, Q * Hp), sl * vt);
            ap = GR_FMA(J, GR_FMA(Q, Ht, -(Pq * Hp)), sl * vp);
            const real vt2 = vt * vt, vtp2 = 2.0 * (vt * vp), vp2 = vp * vp, vr2 = vr * vr, vh2 = vh * vh, vrvh = vr * vh;
            const real B0 = GR_FMA(Phi, vp2, -GR_FMA(Pq, vt2, Q * vtp2));
            const real Br = GR_FMA(Phi_r, vp2, -GR_FMA(Dh_r, vt2, Q_r * vtp2));
            const real Bh = GR_FMA(Phi_h, vp2, -GR_FMA(P_h, vt2, Q_h * vtp2));
            const real B0i = B0 * iSh;
            const real vr2iD = vr2 * iDh;
            ar = iSh * (GR_FMA(0.5 * (Dh * iSh), GR_FMA(-rp, B0i, Br), (0.5 * rp) * (Dh * vh2))
                        - GR_FMA(0.5 * rp, vr2, GR_FMA(Sh_h, vrvh, -((0.5 * Sh) * (Dh_r * vr2iD)))));
            ah = -(iSh * (GR_FMA(rp, vrvh, (0.5 * Sh_h) * (vh2 - vr2iD)) - (0.5 * iSh) * GR_FMA(-Sh_h, B0i, Bh)));
            return;
        } else if constexpr (ID == GR_METRIC_NOZ && GR_DA_FUSED) {
            // NoZ (noz-metric.jl:7-47; y = cosθ, η = ϵ M a y, S = r² + a² y², Se = S + η, D = S² + (S - 2Mr) η): g_tt = h_tt/D,
            // g_tϕ = h_tϕ/D, g_ϕϕ = h_ϕϕ/D with h_tt = 2Mr S - D, h_tϕ = -a 2Mr s² Se, h_ϕϕ = s² Se big; g_rr = Se/Δ, g_θθ = Se.  The t-ϕ
            // block has the determinant -s² Δ Se²/D (Δ = r² - 2Mr + a², Kerr's), so g^tt = -big/(Δ Se), g^tϕ = -a 2Mr/(Δ Se),
            // g^ϕϕ = -h_tt/(s² Δ Se²), and with H_t = ḣ_tt v^t + ḣ_tϕ v^ϕ, H_ϕ = ḣ_tϕ v^t + ḣ_ϕϕ v^ϕ (p_t, p_ϕ drop out: g^tμ p_μ = v^t):
            //   a^t = (big H_t + a 2Mr H_ϕ)/(D Δ Se) + (Ḋ/D) v^t,     a^ϕ = (a 2Mr H_t + h_tt H_ϕ/(s² Se))/(D Δ Se) + (Ḋ/D) v^ϕ,
            //   a^r = -(Δ/Se)[½ 2r v_r²/Δ + Se_θ v_r v_θ/Δ - ½ Se Δ_r v_r²/Δ² - ½ (B_r - D_r B/D)/D - ½ 2r v_θ²],
            //   a^θ = -(1/Se)[2r v_r v_θ + ½ Se_θ (v_θ² - v_r²/Δ) - ½ (B_θ - D_θ B/D)/D],      B = h_tt v_t² + 2 h_tϕ v_t v_ϕ + h_ϕϕ v_ϕ².
            // One reciprocal (of D Δ Se s²) where the dual-number form takes four.  No reflection symmetry: the θ partials are
            // -s times the y partials, odd and even powers of y both present.
            const real a = P[1];
            const real a2 = ka2, tM = ktM, kE = kn_eMa;
            const real y = c, s2 = s * s;
            const real eps = kE * y;
            const real r2 = r * r;
            const real a2y = a2 * y;
            const real a2y2 = a2y * y;
            const real S = r2 + a2y2;
            const real tMr = tM * r;
            const real Se = S + eps;
            const real SmT = S - tMr;
            const real D = GR_FMA(S, S, SmT * eps);
            const real Del = GR_FMA(-tM, r, r2) + a2;
            const real tr = 2.0 * r;
            const real S_y = 2.0 * a2y;
            const real Se_y = S_y + kE;
            const real D_r = GR_FMA(2.0 * tr, S, (tr - tM) * eps);
            const real D_y = GR_FMA(S_y, 2.0 * S + eps, SmT * kE);
            const real q1 = a2 + a2y2 + eps, q2 = a2 - a2y2 - eps;
            const real a4 = a2 * a2;
            const real big = GR_FMA(r2, r2 + q1, GR_FMA(a4, y * y, GR_FMA(a2, eps, tMr * q2)));
            const real big_r = GR_FMA(tr, 2.0 * r2 + q1, tM * q2);
            const real big_y = GR_FMA(2.0 * a4, y, GR_FMA(r2 - tMr, Se_y, a2 * kE));
            const real htt = GR_FMA(tMr, S, -D);
            const real htt_r = GR_FMA(tM, S, GR_FMA(tMr, tr, -D_r));
            const real htt_y = GR_FMA(tMr, S_y, -D_y);
            const real atM = a * tMr;
            const real htp = -(atM * (s2 * Se));
            const real htp_r = -((a * s2) * GR_FMA(tM, Se, tMr * tr));
            const real htp_y = -(atM * GR_FMA(-2.0 * y, Se, s2 * Se_y));
            const real Sb = Se * big;
            const real hpp = s2 * Sb;
            const real hpp_r = s2 * GR_FMA(tr, big, Se * big_r);
            const real hpp_y = GR_FMA(-2.0 * y, Sb, s2 * GR_FMA(Se_y, big, Se * big_y));
            const real DlS = Del * Se;
            const real R = rcp_rhs((D * DlS) * s2);
            const real J1 = R * s2;                      // 1/(D Δ Se)
            const real iD = R * (DlS * s2);
            const real is2Se = R * (D * Del);            // 1/(s² Se)
            const real iSe = is2Se * s2;
            const real iDel = R * ((D * Se) * s2);
            const real yd = -(s * vh);                   // ẏ
            const real Dd = GR_FMA(D_r, vr, D_y * yd);
            const real httd = GR_FMA(htt_r, vr, htt_y * yd), htpd = GR_FMA(htp_r, vr, htp_y * yd), hppd = GR_FMA(hpp_r, vr, hpp_y * yd);
            const real Ht = GR_FMA(httd, vt, htpd * vp), Hp = GR_FMA(htpd, vt, hppd * vp);
            const real dl = Dd * iD;
            at = GR_FMA(J1, GR_FMA(big, Ht, atM * Hp), dl * vt);
            ap = GR_FMA(J1, GR_FMA(atM, Ht, (htt * is2Se) * Hp), dl * vp);
            const real vt2 = vt * vt, vtp2 = 2.0 * (vt * vp), vp2 = vp * vp, vr2 = vr * vr, vh2 = vh * vh, vrvh = vr * vh;
            const real B0 = GR_FMA(htt, vt2, GR_FMA(htp, vtp2, hpp * vp2));
            const real Br = GR_FMA(htt_r, vt2, GR_FMA(htp_r, vtp2, hpp_r * vp2));
            const real By = GR_FMA(htt_y, vt2, GR_FMA(htp_y, vtp2, hpp_y * vp2));
            const real B0i = B0 * iD;
            const real Se_h = -(s * Se_y);               // ∂_θ Se
            const real vr2iD = vr2 * iDel;
            const real Del_r = tr - tM;
            ar = -((Del * iSe) * (GR_FMA(0.5 * tr, vr2iD - vh2, GR_FMA(Se_h * iDel, vrvh, -((0.5 * Se) * (Del_r * iDel)) * vr2iD))
                                  - (0.5 * iD) * GR_FMA(-D_r, B0i, Br)));
            ah = -(iSe * (GR_FMA(tr, vrvh, (0.5 * Se_h) * (vh2 - vr2iD)) + (0.5 * (s * iD)) * GR_FMA(-D_y, B0i, By)));
            return;
        } else if constexpr (ID == GR_METRIC_KERR_DARK_MATTER) {
            // Kerr with the enclosed mass M(r) (kerr-dark-matter.jl:6-49): every component depends on r through M as well, so
            // ∂_r g = ∂_r g|_M + M'(r) ∂_M g, and the right-hand side is Kerr's at M = M(r) (kerr_core: the fused form of
            // KerrFamily::rhs with the mass per lane) plus what M' ∂_M g contributes.  With e = 2r/Σ: ∂_M (g_tt, g_tϕ, g_ϕϕ, g_rr) =
            // (e, -a s² e, a² s⁴ e, 2rΣ/Δ²), so  δT_t = M' e U v^r,  δT_ϕ = -a s² δT_t,  δD_r = M'(e U² + (2rΣ/Δ²) v_r²)  and
            //   δa^t = δT_t (B/Δ - a² s² w/Δ),   δa^ϕ = δT_t a/Δ,   δa^r = -M' r (v_r²/Δ - Δ U²/Σ²),   δa^θ = 0
            // (δT joins the linear forms inside kerr_core, before the inverse block).  Outside the shell (M' = 0) it IS Kerr.
            const real Mbh = P[0], Mdm = P[2], dR = P[3], rs = P[4];
            real M = Mbh, dM = 0.0;
            if (r >= rs + dR) {
                M = Mbh + Mdm;
            } else if (r >= rs) {
                const real idR = rcp_full(dR);
                const real xr = (r - rs) * idR;
                M = GR_FMA(Mdm, (3.0 - 2.0 * xr) * (xr * xr), Mbh);
                dM = (6.0 * Mdm * idR) * (xr * (1.0 - xr));
            }
            KerrMid k;
            kerr_core(M, dM, r, s, c, vt, vr, vh, vp, at, ar, ah, ap, k);
            return;
        } else if constexpr (ID == GR_METRIC_KERR_REFRACTIVE) {
            // Kerr with g_tt / n², g_tϕ / n inside the corona (kerr-refractive-ad.jl:8-33), n(r) = n0 + (1 - n0) t(r), t the smooth
            // step of utils.jl:158-168.  The inverse block is (n² g^tt, n g^tϕ, g^ϕϕ), and with ṽ^t = v^t / n every Kerr form is
            // Kerr's at (ṽ^t, v^r, v^θ, v^ϕ): g̃_tt (v^t)² = g_tt (ṽ^t)², T̃_t = T_t/n - (ṅ/n²) L, T̃_ϕ = T_ϕ - (ṅ/n) g_tϕ ṽ^t,
            // L = 2 g_tt ṽ^t + g_tϕ v^ϕ, D̃_r = D_r - 2 (n'/n)(g_tt (ṽ^t)² + g_tϕ ṽ^t v^ϕ).  So
            //   a^t = n a^t_K + ṅ (g^tt L + g^tϕ g_tϕ ṽ^t),   a^ϕ = a^ϕ_K + (ṅ/n)(g^tϕ L + g^ϕϕ g_tϕ ṽ^t),
            //   a^r = a^r_K - g^rr (n'/n)(g_tt (ṽ^t)² + g_tϕ ṽ^t v^ϕ),   a^θ = a^θ_K,       ṅ = n' v^r.
            // n' lives in the 2.5-wide band round the corona radius; outside it the metric is Kerr at a rescaled v^t.
            const real M = P[0], n0 = P[2], rc = P[3];
            real n = n0, dn = 0.0, in = kin0;
            if (r <= rc - 1.25) {
                n = 1.0; in = 1.0;
            } else if (r <= rc + 1.25) {
                const real z = 4e3 * (r - rc);
                const real t = 0.5 - 0.3183098861837907 * GR_ATAN(z);
                const real tp = -(0.3183098861837907 * 4e3) * rcp_full(GR_FMA(z, z, 1.0));
                n = GR_FMA(1.0 - n0, t, n0);
                dn = (1.0 - n0) * tp;
                in = rcp_full(n);
            }
            const real wt = vt * in;                        // ṽ^t
            KerrMid k;
            kerr_core(M, 0.0, r, s, c, wt, vr, vh, vp, at, ar, ah, ap, k);
            at = n * at;
            if (dn != 0.0) {
                const real gtt = k.w - 1.0, gtp = k.gtp;
                const real gitt = -(k.B * k.iDel), gitp = gtp * k.iDs, gipp = (1.0 - k.w) * k.iDs;
                const real Lf = GR_FMA(2.0 * gtt, wt, gtp * vp);
                const real gw = gtp * wt;
                const real nd = dn * vr, ndn = nd * in;
                at = GR_FMA(nd, GR_FMA(gitt, Lf, gitp * gw), at);
                ap = GR_FMA(ndn, GR_FMA(gitp, Lf, gipp * gw), ap);
                ar = GR_FMA(-((k.Del * k.iSig) * (dn * in)), wt * GR_FMA(gtt, wt, gtp * vp), ar);
            }
            return;
        } else if constexpr (ID == GR_METRIC_BUMBLEBEE) {
            const real tM = ktM, K = ktM * P[1];
            const real rm = r - tM;
            const real Q = rcp_rhs(r * rm);                  // 1/(r (r - 2M))
            const real ir = Q * rm, irm = Q * r;
            const real s2 = s * s, S2 = 2.0 * (s * c);
            const real u = tM * ir, ir2 = ir * ir;
            const real Kir = K * ir;                         // -g_tϕ / s²
            const real r2 = r * r;
            const real Dp = GR_FMA(u - 1.0, r2, -((K * Kir) * (s2 * ir)));       // D' = (u - 1) r² - K² s²/r²
            const real R = rcp_rhs(Dp * s2);                 // 1/(D' s²)
            const real iDp = R * s2, is2Dp = R;
            // metric gradients
            const real gtt_r = -(u * ir);
            const real gtp_r = (Kir * ir) * s2;              // K s²/r²
            const real gtp_h = -(Kir * S2);
            const real gpp_r = (2.0 * r) * s2;
            const real gpp_h = r2 * S2;
            // dots along the ray and the linear forms
            const real gtt_d = gtt_r * vr;
            const real gtp_d = GR_FMA(gtp_r, vr, gtp_h * vh);
            const real gpp_d = GR_FMA(gpp_r, vr, gpp_h * vh);
            const real Tt = GR_FMA(gtt_d, vt, gtp_d * vp);
            const real Tp = GR_FMA(gtp_d, vt, gpp_d * vp);
            at = -(GR_FMA(r2, Tt, Kir * Tp) * iDp);
            ap = -GR_FMA(Kir * iDp, Tt, ((u - 1.0) * is2Dp) * Tp);
            // quadratic forms
            const real vt2 = vt * vt, vp2 = vp * vp, vtp = vt * vp, vh2 = vh * vh;
            const real tr = 2.0 * r;
            const real Dr = GR_FMA(gtt_r, vt2, GR_FMA(tr, vh2, GR_FMA(gpp_r, vp2, (2.0 * gtp_r) * vtp)));     // without the g_rr term
            const real girr = (rm * ir) * kik;                 // g^rr = (r - 2M)/(k r)
            ar = 0.5 * GR_FMA((tM * ir) * irm, vr * vr, girr * Dr);
            const real Dh = GR_FMA(gpp_h, vp2, (2.0 * gtp_h) * vtp);
            ah = GR_FMA(-(2.0 * ir), vr * vh, (0.5 * ir2) * Dh);
            return;
        }
        const real a = P[1];
        const real a2 = ka2, tM = ktM, eps = keps;
        const real s2 = s * s, S2 = 2.0 * (s * c), tr = 2.0 * r;
        const real rho2 = GR_FMA(r, r, a2);
        const real a2S2 = a2 * S2;                         // -Σ_θ
        const real a2s2 = a2 * s2;
        const real Sig = rho2 - a2s2;                      // r² + a² cos²θ (Σ >= r²: no cancellation)
        const real iSig = rcp_rhs(Sig);
        const real Del = GR_FMA(-tM, r, rho2);
        // w, h, η and their gradients
        const real w = (tM * r) * iSig;
        const real w_r = iSig * GR_FMA(-w, tr, tM);
        const real w_h = (w * a2S2) * iSig;
        const real eta = (eps * r) * iSig;                 // h Σ
        const real h = eta * iSig;
        const real eta_r = iSig * GR_FMA(-eta, tr, eps);
        const real eta_h = (eta * a2S2) * iSig;
        const real h_r = iSig * GR_FMA(-2.0 * h, tr, eps * iSig);
        const real h_h = 2.0 * ((h * a2S2) * iSig);
        const real H = 1.0 + h;
        const real Dt = GR_FMA(a2s2, h, Del);              // Δ̃
        const real Dt_r = GR_FMA(a2s2, h_r, tr - tM);
        const real Dt_h = GR_FMA(a2S2, h, a2s2 * h_h);
        // reciprocals
        const real HD = H * Dt;
        const real R = rcp_rhs(HD * s2);                  // 1/(H Δ̃ s²)
        const real iH = R * (Dt * s2), iDt = R * (H * s2), is2Dt = R * H;
        // dots along the ray
        const real hd = GR_FMA(h_r, vr, h_h * vh);
        const real wd = GR_FMA(w_r, vr, w_h * vh);
        const real s2d = S2 * vh;
        const real es2_r = s2 * eta_r;                      // (η s²)_r
        const real es2_h = GR_FMA(S2, eta, s2 * eta_h);    // (η s²)_θ
        const real es2d = GR_FMA(es2_r, vr, es2_h * vh);
        // linear forms
        const real avp = a * vp;
        const real U = GR_FMA(-avp, s2, vt);
        const real wU = w * U;
        const real l = wU - vt;
        const real B = GR_FMA(rho2, vp, -(a * wU));         // ℓ_ϕ^K / s²
        const real lpK = s2 * B;
        const real aws2d = (a * w) * s2d;
        const real Tt = GR_FMA(hd, l, H * GR_FMA(wd, U, -(aws2d * vp)));
        const real Bd = GR_FMA(tr * vr, vp, GR_FMA(-(a * wd), U, (a * aws2d) * vp));      // Ḃ at fixed velocity
        const real lpKd = GR_FMA(s2d, B, s2 * Bd);
        const real Tp = GR_FMA(hd, lpK, GR_FMA(H, lpKd, -(es2d * vp)));
        at = GR_FMA(GR_FMA(a2s2, w, rho2) - eta * iH, Tt, (a * w) * Tp) * iDt;
        ap = GR_FMA(w - 1.0, Tp, ((a * s2) * w) * Tt) * is2Dt;
        // quadratic forms
        const real vp2 = vp * vp, U2 = U * U;
        const real PhiK = GR_FMA(wU, U, GR_FMA(s2 * rho2, vp2, -(vt * vt)));
        const real PhiK_r = GR_FMA(w_r, U2, (tr * s2) * vp2);
        const real PhiK_h = GR_FMA(w_h, U2, S2 * GR_FMA(rho2, vp2, -2.0 * (wU * avp)));
        const real Phi_r = GR_FMA(h_r, PhiK, GR_FMA(H, PhiK_r, -(es2_r * vp2)));
        const real Phi_h = GR_FMA(h_h, PhiK, GR_FMA(H, PhiK_h, -(es2_h * vp2)));
        const real L_r = GR_FMA(tr, iSig, GR_FMA(h_r, iH, -(Dt_r * iDt)));
        const real L_h = GR_FMA(-a2S2, iSig, GR_FMA(h_h, iH, -(Dt_h * iDt)));
        const real vr2 = vr * vr, vh2 = vh * vh, vrh = vr * vh;
        const real girr = (Dt * iSig) * iH;                 // g^rr
        ar = GR_FMA(girr, 0.5 * GR_FMA(tr, vh2, Phi_r), -GR_FMA(0.5 * L_r, vr2, L_h * vrh));
        const real grr = (Sig * H) * iDt;
        ah = iSig * GR_FMA(0.5, GR_FMA(grr * L_h, vr2, GR_FMA(a2S2, vh2, Phi_h)), -(tr * vrh));
    }

    GR_DEV void eval(real r, real s, real c, real g[5], real gr[5], real gt[5], real gi[5]) const
    {
        Dual2 gd[5];
        // seeds: r = (r;1,·), sinθ = (s;·,c), cosθ = (c;·,-s)
        components(DualR{ r, 1.0, 0.0 }, DualT{ s, 0.0, c }, DualT{ c, 0.0, -s }, gd);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            g[i] = gd[i].v;
            gr[i] = gd[i].a;
            gt[i] = gd[i].b;
        }
        inverse_generic(g, gi);
    }
};
typedef GenericMetricT<-1> GenericMetric;      // run-time switch over the catalogue

// geodesic_equation (auto-diff.jl:213-226) with the sparse contraction of SURVEY App. B.1 from the components' Jacobian and
// inverse.  The factors 2 and -½ of the reference's form cancel:
//   a^t = -(g^tt T_t + g^tϕ T_ϕ), a^r = -g^rr (ġ_rr v^r - ½ D_r), ... with T_t = ġ_tt v^t + ġ_tϕ v^ϕ.
GR_DEV void geodesic_contract(const real j1[5], const real j2[5], const real gi[5], real vt, real vr, real vh, real vp,
                              real& at, real& ar, real& ah, real& ap)
{
    real gd[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) gd[k] = GR_FMA(j1[k], vr, j2[k] * vh);
    const real vt2 = vt * vt, vr2 = vr * vr, vh2 = vh * vh, vp2 = vp * vp, vtp = 2.0 * vt * vp;
    const real Dr = j1[0] * vt2 + j1[1] * vr2 + j1[2] * vh2 + j1[3] * vp2 + j1[4] * vtp;
    const real Dh = j2[0] * vt2 + j2[1] * vr2 + j2[2] * vh2 + j2[3] * vp2 + j2[4] * vtp;
    const real Tt = GR_FMA(gd[0], vt, gd[4] * vp);
    const real Tp = GR_FMA(gd[4], vt, gd[3] * vp);
    const real Tr = GR_FMA(gd[1], vr, -0.5 * Dr);
    const real Th = GR_FMA(gd[2], vh, -0.5 * Dh);
    at = -GR_FMA(gi[0], Tt, gi[4] * Tp);
    ar = -(gi[1] * Tr);
    ah = -(gi[2] * Th);
    ap = -GR_FMA(gi[4], Tt, gi[3] * Tp);
}

// ---------------------------------------------------------------------------------------
// GR_METRIC_TABULATED: a user-defined metric from the piecewise-polynomial table of gr_tabmetric.hpp -- the AbstractMetric plugin
// interface (metric_components(m, (r, θ)), src/metrics/kerr-metric.jl:62-70) on the device.
//
// Memory system.  A patch is 1440 bytes of coefficients (degree 7) and a right-hand side reads all of it for 385 FP64 operations.
// Per-LANE fetches of that volume from L1 / L2 would be bound by the return path (64 bytes per clock and CU against 64 FP64 FMAs
// per clock).  But the 64 rays of a wave are an 8 x 8 pixel tile: they sit in ONE patch in 58 % of the wave-steps of the 2048²
// bench plane and in <= 2 in 91 %.  So every wave keeps a CACHE OF PATCHES IN LDS (TabLds): kTabSlots slots of one patch, tags in
// a 64-byte head; a lane compares its patch number with the tags (read as vectors: one LDS latency per look-up), and reads its
// coefficients from its slot with ds_read_b128 -- lanes in one slot read one address (a broadcast), lanes in different slots
// different bank groups (the slot stride is 208 mod 256 bytes), so the cost of an evaluation does not depend on how many patches
// the wave straddles.  The reads run GR_TAB_LOOKAHEAD coefficients ahead of the arithmetic, pinned there by a data dependence
// (CoefStream).  A patch that is not resident is copied by all active lanes into a slot no lane of this evaluation reads; a wave
// that straddles more patches than it has slots (the shadow's edge) sends the lanes left over to global memory through one
// out-of-line copy of the evaluation.  The scalar path that was built first -- s_load the patch of the first unfinished lane,
// coefficients as SGPR operands, lanes of other patches waiting their turn -- lost by 1.7x: the scalar data cache keeps nothing
// between two evaluations of a wave (DESIGN_measurements.md §M16).
//
// The scalar type.  `real` is double in the fp64 kernels and a value with two tangents in the tangent flavour (gr_tangent.hpp): the
// patch is located from VALUES, the local coordinates are lifted (du = su dr) and the same recurrences run on the lifted numbers --
// the tangents of g and of ∂g come out of the polynomial's own second derivatives, as the reference's nested ForwardDiff does for
// a closure (src/tracing/precision-solvers.jl:401-451).  In the fp32 kernels (`real` = float: tolerance sweeps, gr_ctx_set
// "precision" 32) the table stays what it is -- fp64 data -- and its polynomials are evaluated in DOUBLE (tab_real): converting the
// 105 coefficients of an evaluation would cost as many FP64-rate instructions as the 210 operations themselves; the components
// and their Jacobian are rounded to float once and the inverse, the contraction and the whole step run in single precision.
// ---------------------------------------------------------------------------------------
#define GR_HAS_TABULATED 1      // (every flavour of the kernels: fp64, fp32, tangents)
#ifndef GR_TAB_SLOTS
#define GR_TAB_SLOTS 12
#endif
#ifndef GR_TAB_FETCH
#define GR_TAB_FETCH 1      // missing patches a wave copies into its cache side by side; 2 and 4 measured equal (27.7 / 28.3 ms against
#endif                      // 27.4 at 1024², profiles/r5q_tab_slots_ab.log): the slow waves are slow in the global-memory evaluation

#ifndef GR_TAB_LANE_WAVES
#ifdef GR_REAL_IS_TAN2
#define GR_TAB_LANE_WAVES 1
#else
#define GR_TAB_LANE_WAVES 2
#endif
#endif
#ifndef GR_TAB_PARK
#define GR_TAB_PARK 0       // stage accelerations parked in LDS (ParkA); measured equal at two waves per SIMD (profiles/r5e_tab_ab.log)
#endif
#ifndef GR_TAB_LOOKAHEAD_GLOBAL
#define GR_TAB_LOOKAHEAD_GLOBAL 32
#endif
#ifndef GR_TAB_LOOKAHEAD
#define GR_TAB_LOOKAHEAD 16     // coefficients the LDS reads of an evaluation run ahead of its arithmetic (LdsCoef)
#endif
// The wave's patch cache in LDS: [ 16 ints: kTabSlots tags, -1 up to index 14, the round-robin counter at 15 | kTabSlots slots of
// kTabSlotBytes ].  The tags are read four at a time (ds_read_b128), so a look-up costs one LDS latency whatever the number of slots.
// The slot stride is 208 mod 256 bytes: slot k starts 208 k bytes (mod 256) into the 64 banks -- distinct 16-byte columns for k < 16,
// so the lanes of one ds_read group that sit in different slots do not collide.
constexpr int kTabSlots = GR_TAB_SLOTS;
constexpr int kTabTagVecs = (kTabSlots + 3) / 4;
constexpr int kTabRR = 15;              // index of the round-robin counter among the 16 ints of the head
constexpr int kTabFetch = GR_TAB_FETCH;
constexpr int kTabCopyDepth = 8 / kTabFetch;        // loads per patch a lane keeps in flight while it copies (8 in flight in all)
constexpr int kTabHeadBytes = 64;
// (the smallest size >= a patch that is 208 mod 256: 1488 at degree 7, 1232 at 6, 976 at 5)
constexpr int kTabSlotBytes = gr_tab::kPatchDoubles * 8 + ((208 - gr_tab::kPatchDoubles * 8 % 256) + 256) % 256;
static_assert(kTabSlotBytes % 256 == 208 && kTabSlotBytes % 16 == 0, "slot stride");
constexpr size_t kTabLdsBytesPerWave = kTabHeadBytes + (size_t)kTabSlots * kTabSlotBytes;
static_assert(kTabSlots >= 1 && kTabSlots <= 12 && gr_tab::kPatchDoubles * 8 <= kTabSlotBytes, "patch cache geometry");
static_assert(kTabFetch == 1 || kTabFetch == 2 || kTabFetch == 4, "patches copied side by side");
typedef double double2_t __attribute__((ext_vector_type(2)));

// the scalar the patch polynomials are evaluated in
#ifdef GR_REAL_IS_FLOAT
typedef double tab_real;
#else
typedef real tab_real;
#endif
// local coordinate -> that scalar: the tangents of u are du/dr times the tangents of r
GR_DEV tab_real tab_lift(double u, double su, real r)
{
#ifdef GR_REAL_IS_TAN2
    gr_tan2 x(u);
    x.a = su * r.a;
    GR_TAN_B(x.b = su * r.b;)
    return x;
#else
    (void)su; (void)r;
    return u;
#endif
}
// gr_tab::eval_patch's operations on `real`
struct TabRealOps {
#ifndef GR_REAL_IS_FLOAT
    static GR_DEV real fma(real a, real b, real c) { return GR_FMA(a, b, c); }
    static GR_DEV real fmak(real a, real b, double k) { return GR_FMA(a, b, k); }
    static GR_DEV real add(real a, real b) { return a + b; }
    static GR_DEV real mulk(real a, double k) { return a * k; }
    static GR_DEV real addk(real a, double k) { return a + k; }
#endif
    static GR_DEV void row_done(int, int, tab_real&, tab_real&, tab_real&) {}
#if defined(GR_REAL_IS_TAN2) || defined(GR_REAL_IS_FLOAT)
    static GR_DEV double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static GR_DEV double fmak(double a, double b, double k) { return __builtin_fma(a, b, k); }
    static GR_DEV double add(double a, double b) { return a + b; }
    static GR_DEV double mulk(double a, double k) { return a * k; }
    static GR_DEV double addk(double a, double k) { return a + k; }
    static GR_DEV double lift(double k) { return k; }
    static GR_DEV void row_done2(int, int, double&, double&, double&, double&, double&, double&) {}
#endif
};

#ifndef GR_HOST_HARNESS
// the right-hand side with the coefficients read per lane from global memory, NOT inlined: the step loop holds six copies of the
// LDS-fed evaluation already; this one serves lanes that found no cache slot
__device__ __attribute__((noinline)) void tab_rhs_from_global(const double* pc, const double* ax, int form, tab_real u, tab_real v, double su, double sv,
                                                             real s, real c, real vt, real vr, real vh, real vp, real* out);
#endif
struct TabulatedMetric {
    static constexpr int kMinWavesPerSimd = GR_TAB_LANE_WAVES < 2 ? GR_TAB_LANE_WAVES : 2;
    static constexpr int kLaneWavesPerSimd = GR_TAB_LANE_WAVES;
    static constexpr int kParkStages = GR_TAB_PARK;      // stage accelerations parked in LDS while a right-hand side runs (ParkA)
    static constexpr bool kHasForce = false;
    static constexpr bool kFusedRhs = false;
    static constexpr bool kByTheta = true;      // evaluated at (r, θ) -- the integrator hands θ over next to sin θ, cos θ
    gr_tab::GridK gk;
    int32_t form;                               // gr_tab H_POLE_FACTOR: how g_ϕϕ, g_tϕ are stored (0, 1, 2)
    const double* patches;                      // device: first patch of the table
    const double* axis;                         //         the axis terms of form 2 (kAxisDoubles per radial row)
    const gr_tab::SegRec* segs;                 //         the segment records
    // The host unit lays the grid out in cfg.params the way the kernels use it (stage_metric_table): doubles as doubles, the
    // integers packed into the BITS of params[5..7] -- a double -> int conversion would be a vector instruction whose
    // (uniform) result then sits in vector registers for the whole step loop; kernel arguments arrive in scalar registers and
    // bit fields of them stay there.
    GR_DEV void load(const gr_config& c)
    {
        gk.r0 = c.params[0];
        gk.xmin = c.params[1];
        gk.mr = c.params[2];
        gk.nth_over_pi = c.params[3];
        const unsigned long long b5 = __builtin_bit_cast(unsigned long long, c.params[5]), b6 = __builtin_bit_cast(unsigned long long, c.params[6]),
                                 b7 = __builtin_bit_cast(unsigned long long, c.params[7]);
        gk.e_min = (int32_t)(uint32_t)(b6 & 0xffffffffull);
        gk.e_max = (int32_t)(uint32_t)(b6 >> 32);
        gk.m_r = (int32_t)(b7 & 0xffffull);
        gk.n_theta = (int32_t)((b7 >> 16) & 0xffffull);
        form = (int32_t)((b7 >> 32) & 3ull);
        gk.n_seg = (int32_t)((b7 >> 40) & 0xffull);
        patches = c.metric_table + (long long)b5;
        axis = c.metric_table + gr_tab::kBodyOff;
        segs = reinterpret_cast<const gr_tab::SegRec*>(c.metric_table + gr_tab::kSegOff);
    }
    // ---- where a point lies ----
    // One-segment tables (every smooth metric): segment 0 from the kernel arguments, nothing loaded.  Otherwise the segment of each
    // lane is counted from the records' lower ends and the lanes of one segment take its record through SCALAR loads (the record
    // of the first unfinished lane; the others wait their turn -- a wave rarely straddles a segment boundary).
    GR_DEV void locate_any(double r, double th, int& row, int& patch, double& u, double& v, double& su, double& sv) const
    {
        if (gk.n_seg == 1) {
            gr_tab::locate(gk, r, th, row, patch, u, v, su, sv);
            return;
        }
#ifdef GR_HOST_HARNESS
        gr_tab::locate_segments(gk, segs, r, th, row, patch, u, v, su, sv);
#else
        int it;
        gr_tab::locate_theta(gk, th, it, v, sv);
        typedef const gr_tab::SegRec __attribute__((address_space(4))) cseg;
        cseg* sc = (cseg*)(unsigned long long)segs;
        const int sg = gr_tab::segment_of(gk.n_seg, sc, r);
        row = 0; u = 0.0; su = 0.0;
        for (;;) {
            const int s0 = __builtin_amdgcn_readfirstlane(sg);
            if (sg == s0) {
                gr_tab::locate_row(sc[s0].anchor, sc[s0].xmin, sc[s0].e_lo, sc[s0].e_hi, sc[s0].first_row, sc[s0].dir, sc[s0].core, gk.mr, gk.m_r, r, row, u, su);
                break;
            }
        }
        patch = row * gk.n_theta + it;
#endif
    }
    // ---- the axis terms of form 2: K_m, ∂r K_m, K_d, ∂r K_d of g_ϕϕ and g_tϕ at this lane's radius (per-lane loads: 4 (p + 1)
    // doubles that the lanes of a wave mostly share; only metrics whose azimuthal components do not vanish on the axis come here)
    static GR_DEV void axis_terms(const double* ax, tab_real u, double su, tab_real out[8])
    {
#pragma unroll
        for (int q = 0; q < gr_tab::kAxisPolys; ++q) {
            tab_real K, Ku;
            gr_tab::eval_axis_poly<tab_real>([ax](int k) { return ax[k]; }, q * (gr_tab::kDegree + 1), u, K, Ku);
            out[2 * q] = K;
            out[2 * q + 1] = Ku * su;
        }
    }
    // ---- where the coefficients come from ----
    // (a) cold code (initial conditions, point functions, the path and apply kernels): every lane loads its own patch from
    //     global memory.
    // (b) the step loop: the wave's LDS patch cache (above).
#ifndef GR_HOST_HARNESS
    // The coefficient stream of one evaluation out of an LDS slot: ds_read_b128 of the pairs the recurrences consume in
    // order.  Reads run GR_TAB_LOOKAHEAD coefficients ahead of the arithmetic: when a row of a component has been folded in,
    // the pairs up to that distance beyond it are requested, so a read's LDS latency lies behind the FMAs of the rows before
    // it and only the look-ahead (2 registers per coefficient) is held in registers.  Left to the scheduler all reads go to
    // the top of the evaluation (360 registers: 1.3 KB of scratch per lane) -- scheduling barriers do not hold the pure
    // arithmetic in place, a DATA dependence does: the address of the next reads passes through an empty asm that also takes
    // the accumulators of the row just finished.
    // The loads are volatile so that they stay 16-byte reads (the optimiser otherwise narrows them to the doubles used and the
    // back end pairs those as ds_read2_b64: half the LDS bandwidth).
    template <class PairPtr, class AddrInt, int LOOKAHEAD>
    struct CoefStream {
        typedef const double2_t __attribute__((address_space(3))) lds_cdouble2;
        static constexpr int kAll = gr_tab::kComps * gr_tab::kCoefs;       // 180 coefficients = 90 pairs at degree 7
        // (rows p and p - 1 of a component -- 3 coefficients -- and its first Horner row are consumed before the first row_done of
        // the component, the longest row between two calls has p + 1: a shorter look-ahead would read buf[] entries never loaded)
        static_assert(LOOKAHEAD >= gr_tab::kDegree + 1 && LOOKAHEAD % 2 == 0, "the look-ahead must cover one row of coefficients");
        // LDS reads are volatile (see above).  Global ones must NOT be: a volatile global load is a system-scope one (sc0 sc1) that
        // no cache may serve -- every look-ahead batch of the evaluation from global memory then waits for memory itself, 7.4 µs
        // per evaluation (scripts/wave_timeline.py); the load/store vectoriser keeps plain global loads 16 bytes wide.
        static constexpr bool kVolatile = sizeof(AddrInt) == 4;
        static constexpr int kPairs = (kAll + 1) / 2;                       // (an odd count at degree 5: the last pair's second half is padding)
        mutable PairPtr sl;
        mutable double2_t buf[kPairs];
        // coefficients consumed once row `row` of component `comp` is done, and the pairs requested by then
        static constexpr int consumed(int comp, int row) { return gr_tab::kCoefs * comp + gr_tab::row_offset(row) + (gr_tab::kDegree - row + 1); }
        static constexpr int pairs_by(int coefs) { return coefs + LOOKAHEAD >= kAll ? kPairs : (coefs + LOOKAHEAD + 1) / 2; }
        GR_DEV void issue(int from_pair, int to_pair) const
        {
#pragma unroll
            for (int j = from_pair; j < to_pair; ++j) {
                if constexpr (kVolatile) buf[j] = *(volatile typename std::remove_pointer<PairPtr>::type*)(sl + j);
                else buf[j] = sl[j];
            }
        }
        GR_DEV void start() const { issue(0, pairs_by(0)); }
        GR_DEV double operator()(int kk) const { return buf[kk >> 1][kk & 1]; }
        // the five operations of gr_tab::eval_patch, and the hook that keeps the stream ahead
#ifndef GR_REAL_IS_FLOAT
        static GR_DEV real fma(real a, real b, real c) { return GR_FMA(a, b, c); }
        static GR_DEV real fmak(real a, real b, double k) { return GR_FMA(a, b, k); }
        static GR_DEV real add(real a, real b) { return a + b; }
        static GR_DEV real mulk(real a, double k) { return a * k; }
        static GR_DEV real addk(real a, double k) { return a + k; }
#endif
#if defined(GR_REAL_IS_TAN2) || defined(GR_REAL_IS_FLOAT)
        // the tangent flavour evaluates the patch on PLAIN numbers with second derivatives (gr_tab::eval_patch2) and contracts;
        // the fp32 kernels evaluate it in double (tab_real)
        static GR_DEV double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
        static GR_DEV double fmak(double a, double b, double k) { return __builtin_fma(a, b, k); }
        static GR_DEV double add(double a, double b) { return a + b; }
        static GR_DEV double mulk(double a, double k) { return a * k; }
        static GR_DEV double addk(double a, double k) { return a + k; }
        static GR_DEV double lift(double k) { return k; }
#endif
#ifdef GR_REAL_IS_TAN2
        GR_DEV void row_done2(int comp, int row, double& p0, double& p1, double& p2, double& p3, double& p4, double& p5) const
        {
            const int before = (row == gr_tab::kDegree - 2) ? (comp == 0 ? 0 : consumed(comp - 1, 0)) : consumed(comp, row + 1);
            const int from = pairs_by(before), to = pairs_by(consumed(comp, row));
            if (to <= from) return;
            AddrInt a = (AddrInt)(unsigned long long)sl;
            asm volatile("" : "+v"(a), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5));      // (all six chains: see row_done)
            sl = (PairPtr)(unsigned long long)a;
            issue(from, to);
        }
#endif
        GR_DEV void row_done(int comp, int row, tab_real& acc, tab_real& acc_u, tab_real& acc_v) const
        {
            // (rows kDegree and kDegree - 1 of a component are folded together before the first call for it)
            const int before = (row == gr_tab::kDegree - 2) ? (comp == 0 ? 0 : consumed(comp - 1, 0)) : consumed(comp, row + 1);
            const int from = pairs_by(before), to = pairs_by(consumed(comp, row));
            if (to <= from) return;
            AddrInt a = (AddrInt)(unsigned long long)sl;      // (a 32-bit LDS address, or a 64-bit global one)
            // (all three accumulators: a chain left out is deferred by the scheduler to the end of the evaluation, with every
            // intermediate of the value chain it reads kept alive -- spilled -- until then)
#ifdef GR_REAL_IS_TAN2
            asm volatile("" : "+v"(a), "+v"(acc.v), "+v"(acc_u.v), "+v"(acc_v.v), "+v"(acc.a), "+v"(acc_u.a), "+v"(acc_v.a));
            GR_TAN_B(asm volatile("" : "+v"(a), "+v"(acc.b), "+v"(acc_u.b), "+v"(acc_v.b));)
#else
            asm volatile("" : "+v"(a), "+v"(acc), "+v"(acc_u), "+v"(acc_v));
#endif
            sl = (PairPtr)(unsigned long long)a;
            issue(from, to);
        }
    };
    typedef CoefStream<const double2_t __attribute__((address_space(3)))*, unsigned, GR_TAB_LOOKAHEAD> LdsCoef;      // out of a cache slot
    // ... out of the table in global memory (tab_rhs_from_global): memory latency is ten times the LDS's, and the lanes that take
    // this path are few and often alone in their wave -- twice the look-ahead
    // (an address-space-1 pointer: through a generic one the loads are flat_load's, which count on the LDS counter as well and
    // return out of order there -- every wait for one of them then waits for all of them, and the look-ahead is gone)
    typedef CoefStream<const double2_t __attribute__((address_space(1)))*, unsigned long long, GR_TAB_LOOKAHEAD_GLOBAL> GlobalCoef;
#endif
    // components and Jacobian from a coefficient source: the polynomials, the chain rule, the axis forms
    // (ax: this lane's row of the axis terms, read in form 2 only)
    template <class Ld, class Ops_>
    static GR_DEV void horner(const Ld& ld, const Ops_& ops, int form, const double* ax, tab_real u, tab_real v, double su, double sv, real s, real c,
                              real g[5], real gr[5], real gt[5])
    {
#ifdef GR_REAL_IS_TAN2
        // Value + two tangents: the patch on plain numbers with its second derivatives, then the chain rule.  With
        // δu = su δr, δv = sv δθ (the tangents the lifted u, v carry):
        //   δg = Pu δu + Pv δv,   δ(∂r g) = su (Puu δu + Puv δv),   δ(∂θ g) = sv (Puv δu + Pvv δv)
        // -- 372 operations per evaluation where the recurrences on lifted numbers took ~1050 (73.6 ms and 1.3 KB of scratch per lane
        // for 1024² rays).
        double P[5], Pu[5], Pv[5], Puu[5], Puv[5], Pvv[5];
        gr_tab::eval_patch2<double>(ld, ops, u.v, v.v, P, Pu, Pv, Puu, Puv, Pvv);
        const double dua = u.a, dva = v.a;
        const double s2ua = 2.0 * su * dua, suva = su * dva, svua = sv * dua, s2va = 2.0 * sv * dva;      // (eval_patch2 returns ½ Puu, ½ Pvv)
#if GR_TAN_W == 2
        const double dub = u.b, dvb = v.b;
        const double s2ub = 2.0 * su * dub, suvb = su * dvb, svub = sv * dub, s2vb = 2.0 * sv * dvb;
#endif
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            g[k] = real(P[k]);
            g[k].a = __builtin_fma(Pu[k], dua, Pv[k] * dva);
            gr[k] = real(Pu[k] * su);
            gr[k].a = __builtin_fma(Puu[k], s2ua, Puv[k] * suva);
            gt[k] = real(Pv[k] * sv);
            gt[k].a = __builtin_fma(Puv[k], svua, Pvv[k] * s2va);
#if GR_TAN_W == 2
            g[k].b = __builtin_fma(Pu[k], dub, Pv[k] * dvb);
            gr[k].b = __builtin_fma(Puu[k], s2ub, Puv[k] * suvb);
            gt[k].b = __builtin_fma(Puv[k], svub, Pvv[k] * s2vb);
#endif
        }
        if (form != 0) {
            gr_tab::pole_factor_apply<real>(s * s, 2.0 * (s * c), g, gr, gt);
            if (form == 2) {
                real at8[8];
                axis_terms(ax, u, su, at8);
                gr_tab::axis_terms_apply<real>(at8, s, c, g, gr, gt);
            }
        }
#else
        // (tab_real is `real` in the fp64 kernels; in the fp32 ones everything down to the last loop is double)
        tab_real G[5], Gr[5], Gt[5];
        {
            tab_real P[5], Pu[5], Pv[5];
            gr_tab::eval_patch<tab_real>(ld, ops, u, v, P, Pu, Pv);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                G[k] = P[k];
                Gr[k] = Pu[k] * su;
                Gt[k] = Pv[k] * sv;
            }
        }
        if (form != 0) {
            const tab_real sd = s, cd = c;
            gr_tab::pole_factor_apply<tab_real>(sd * sd, 2.0 * (sd * cd), G, Gr, Gt);
            if (form == 2) {
                tab_real at8[8];
                axis_terms(ax, u, su, at8);
                gr_tab::axis_terms_apply<tab_real>(at8, sd, cd, G, Gr, Gt);
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            g[k] = (real)G[k];
            gr[k] = (real)Gr[k];
            gt[k] = (real)Gt[k];
        }
#endif
    }
    // components and Jacobian at (r, θ); s, c = sin θ, cos θ of the same θ (the axis factor of g_ϕϕ and g_tϕ)
    GR_DEV void poly(real r, real th, real s, real c, real g[5], real gr[5], real gt[5]) const
    {
        int row, patch;
        double u, v, su, sv;
        locate_any((double)r, (double)th, row, patch, u, v, su, sv);
        const double* pc = patches + (int64_t)patch * gr_tab::kPatchDoubles;
        horner([pc](int k) { return pc[k]; }, TabRealOps{}, form, axis + (int64_t)row * gr_tab::kAxisDoubles, tab_lift(u, su, r), tab_lift(v, sv, th),
               su, sv, s, c, g, gr, gt);
    }
    // the right-hand side of the geodesic equation from the table's components: inverse and contraction
    static GR_DEV void finish_rhs(const real g[5], const real gr[5], const real gt[5], real vt, real vr, real vh, real vp,
                                  real& at, real& ar, real& ah, real& ap)
    {
        real gi[5];
        inverse_generic(g, gi);
        geodesic_contract(gr, gt, gi, vt, vr, vh, vp, at, ar, ah, ap);
    }
    // ... inside the step loop: through the wave's patch cache (cs.tab, TabLds).  The branches below enclose the WHOLE right-hand
    // side: four accelerations are carried out of them, not fifteen components.
    template <class Cold_>
    GR_DEV void rhs_th(const Cold_& cs, real r, real th, real s, real c, real vt, real vr, real vh, real vp,
                       real& at, real& ar, real& ah, real& ap) const
    {
#ifdef GR_HOST_HARNESS
        real g[5], gr[5], gt[5];
        poly(r, th, s, c, g, gr, gt);
        finish_rhs(g, gr, gt, vt, vr, vh, vp, at, ar, ah, ap);
#else
        if constexpr (!Cold_::kTabLds) {
            real g[5], gr[5], gt[5];
            poly(r, th, s, c, g, gr, gt);
            finish_rhs(g, gr, gt, vt, vr, vh, vp, at, ar, ah, ap);
        } else {
            int row, patch;
            double u, v, su, sv;
            locate_any((double)r, (double)th, row, patch, u, v, su, sv);
            typedef volatile int __attribute__((address_space(3))) lds_vint;
            lds_vint* tags = (lds_vint*)cs.tab;
            // Lanes are threads to the compiler: it orders ONE lane's memory operations, not one lane's reads against another
            // lane's writes.  Every hand-over between lanes below (a patch copied by some lanes and read by others, a slot read by
            // some lanes and replaced for others) is therefore fenced at wave scope, and the tags are read as volatile.
#define GR_TAB_WAVE_SYNC()                                        \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
            // -- phase 1: a slot for every lane
            typedef int int4_t __attribute__((ext_vector_type(4)));
            typedef volatile int4_t __attribute__((address_space(3))) lds_vint4;
            int slot = -1;
#pragma unroll
            for (int q = 0; q < kTabTagVecs; ++q) {
                const int4_t t4 = ((lds_vint4*)cs.tab)[q];
                if (4 * q + 0 < kTabSlots) slot = (t4.x == patch) ? 4 * q + 0 : slot;
                if (4 * q + 1 < kTabSlots) slot = (t4.y == patch) ? 4 * q + 1 : slot;
                if (4 * q + 2 < kTabSlots) slot = (t4.z == patch) ? 4 * q + 2 : slot;
                if (4 * q + 3 < kTabSlots) slot = (t4.w == patch) ? 4 * q + 3 : slot;
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(slot < 0);
            if (todo != 0ull) {
#ifdef GR_WAVE_TIMELINE
                const unsigned long long tl_c0 = wall_clock64();
#endif
                // Slots that lanes of THIS evaluation read must stay; the others are replaced round robin.  Up to kTabFetch missing
                // patches are chosen first (scalar work) and copied side by side by all active lanes, their loads in flight together.
                unsigned used = 0;
#pragma unroll
                for (int k = 0; k < kTabSlots; ++k) used |= __builtin_amdgcn_ballot_w64(slot == k) != 0ull ? (1u << k) : 0u;
                const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
                const int n_act = __builtin_popcountll(act);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
                int rr = tags[kTabRR];
                typedef double2_t __attribute__((address_space(3))) lds_double2;
                bool room = true;
                while (todo != 0ull && room) {
                    int pp[kTabFetch], ss[kTabFetch];
                    int nf = 0;
#pragma unroll
                    for (int f = 0; f < kTabFetch; ++f) {
                        pp[f] = -1; ss[f] = 0;
                        if (todo != 0ull && room) {
                            int sidx = -1;
                            for (int tries = 0; tries < kTabSlots && sidx < 0; ++tries) {
                                const int cand = rr;
                                rr = rr + 1 >= kTabSlots ? 0 : rr + 1;
                                if (!(used & (1u << cand))) sidx = cand;
                            }
                            if (sidx < 0) {
                                room = false;      // every slot is read by this evaluation: the lanes left over go to global memory
                            } else {
                                const int first = (int)__builtin_ctzll(todo);
                                pp[f] = __builtin_amdgcn_readlane(patch, first);
                                ss[f] = sidx;
                                used |= 1u << sidx;
                                todo &= ~__builtin_amdgcn_ballot_w64(patch == pp[f]);
                                nf = f + 1;
                            }
                        }
                    }
                    // a patch in pieces of 16 bytes, piece q by the active lane of rank q mod n_act
                    for (int q0 = rank; q0 < gr_tab::kPatchDoubles / 2; q0 += kTabCopyDepth * n_act) {
                        double2_t piece[kTabFetch][kTabCopyDepth];
#pragma unroll
                        for (int f = 0; f < kTabFetch; ++f) {
                            if (f < nf) {
                                const double2_t* src = (const double2_t*)(patches + (int64_t)pp[f] * gr_tab::kPatchDoubles);
#pragma unroll
                                for (int jj = 0; jj < kTabCopyDepth; ++jj) {
                                    const int q = q0 + jj * n_act;
                                    if (q < gr_tab::kPatchDoubles / 2) piece[f][jj] = src[q];
                                }
                            }
                        }
#pragma unroll
                        for (int f = 0; f < kTabFetch; ++f) {
                            if (f < nf) {
                                lds_double2* dst = (lds_double2*)(cs.tab + kTabHeadBytes + ss[f] * kTabSlotBytes);
#pragma unroll
                                for (int jj = 0; jj < kTabCopyDepth; ++jj) {
                                    const int q = q0 + jj * n_act;
                                    if (q < gr_tab::kPatchDoubles / 2) dst[q] = piece[f][jj];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int f = 0; f < kTabFetch; ++f) {
                        if (f < nf) {
                            tags[ss[f]] = pp[f];
                            if (patch == pp[f]) slot = ss[f];
                        }
                    }
#ifdef GR_WAVE_TIMELINE      // (debug builds: patches copied by this wave, scripts/wave_timeline.py)
                    if (rank == 0) tags[12] = tags[12] + nf;
#endif
                }
                tags[kTabRR] = rr;
                GR_TAB_WAVE_SYNC();
#ifdef GR_WAVE_TIMELINE      // (debug builds: clock ticks this wave spent fetching patches)
                if (rank == 0) tags[13] = tags[13] + (int)(wall_clock64() - tl_c0);
#endif
            }
            const tab_real ul = tab_lift(u, su, r), vl = tab_lift(v, sv, th);
            // -- phase 2: the lanes that have a slot evaluate out of LDS ...
            if (slot >= 0) {
                LdsCoef lc;
                lc.sl = (typename LdsCoef::lds_cdouble2*)(cs.tab + kTabHeadBytes + slot * kTabSlotBytes);
                lc.start();
                real g[5], gr[5], gt[5];
                horner(lc, lc, form, axis + (int64_t)row * gr_tab::kAxisDoubles, ul, vl, su, sv, s, c, g, gr, gt);
                finish_rhs(g, gr, gt, vt, vr, vh, vp, at, ar, ah, ap);
            } else {
                // ... and a wave that straddles more patches than it has slots sends the lanes left over to global memory, each
                // for itself, through ONE out-of-line copy of the evaluation (the shadow's edge, where neighbouring rays part: a
                // few waves per launch; replacing slots for them instead costs ~7 patch copies per evaluation -- those waves then
                // run 100 µs per step and set the duration of every launch, 45 ms at any image size: profiles/r5f_tab256_*)
                real out[4];
#ifdef GR_WAVE_TIMELINE      // (debug builds: clock ticks in the evaluation from global memory)
                const unsigned long long tl_f0 = wall_clock64();
#endif
                tab_rhs_from_global(patches + (int64_t)patch * gr_tab::kPatchDoubles, axis + (int64_t)row * gr_tab::kAxisDoubles, form, ul, vl, su, sv,
                                    s, c, vt, vr, vh, vp, out);
#ifdef GR_WAVE_TIMELINE
                {
                    const unsigned long long fa = __builtin_amdgcn_ballot_w64(true);
                    if ((int)__builtin_amdgcn_mbcnt_hi((unsigned)(fa >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fa, 0u)) == 0)
                        tags[14] = tags[14] + (int)(wall_clock64() - tl_f0);
                }
#endif
                at = out[0]; ar = out[1]; ah = out[2]; ap = out[3];
            }
            GR_TAB_WAVE_SYNC();      // the slots just read may be replaced by the next evaluation
#undef GR_TAB_WAVE_SYNC
        }
#endif
    }
    GR_DEV void eval_th(real r, real th, real s, real c, real g[5], real gr[5], real gt[5], real gi[5]) const
    {
        poly(r, th, s, c, g, gr, gt);
        inverse_generic(g, gi);
    }
    GR_DEV void comps_th(real r, real th, real s, real c, real g[5]) const
    {
        real gr[5], gt[5];
        poly(r, th, s, c, g, gr, gt);
    }
};
#ifndef GR_HOST_HARNESS
__device__ __attribute__((noinline)) void tab_rhs_from_global(const double* pc, const double* ax, int form, tab_real u, tab_real v, double su, double sv,
                                                             real s, real c, real vt, real vr, real vh, real vp, real* out)
{
    // (the same coefficient stream as out of LDS: a few loads ahead of the arithmetic, ~100 registers -- with all loads hoisted
    // this function needs 254 registers and saves / restores a hundred callee-saved ones around its body)
    real g[5], gr[5], gt[5];
    TabulatedMetric::GlobalCoef gc;
    gc.sl = (const double2_t __attribute__((address_space(1)))*)(unsigned long long)pc;
    gc.start();
    TabulatedMetric::horner(gc, gc, form, ax, u, v, su, sv, s, c, g, gr, gt);
    TabulatedMetric::finish_rhs(g, gr, gt, vt, vr, vh, vp, out[0], out[1], out[2], out[3]);
}
#endif

// metric id -> functor type (the ids of include/gradus_mi355x.h)
template <int ID> struct MetricOf { typedef GenericMetricT<ID> type; };
template <> struct MetricOf<GR_METRIC_KERR> { typedef KerrFamily<false> type; };
template <> struct MetricOf<GR_METRIC_KERR_NEWMAN> { typedef KerrFamily<true> type; };
template <> struct MetricOf<GR_METRIC_JOHANNSEN> { typedef JohannsenMetric type; };
#if GR_HAS_TABULATED
template <> struct MetricOf<GR_METRIC_TABULATED> { typedef TabulatedMetric type; };
#endif

// does a metric want (r, θ) instead of (r, sin θ, cos θ)?  (TabulatedMetric)
template <class Metric, class = void>
struct ByThetaOf { static constexpr bool value = false; };
template <class Metric>
struct ByThetaOf<Metric, decltype((void)Metric::kByTheta)> { static constexpr bool value = Metric::kByTheta; };

// components / components + Jacobian + inverse at a point whose θ AND sin θ, cos θ the caller has
template <class Metric>
GR_DEV void metric_comps(const Metric& m, real r, real th, real s, real c, real g[5])
{
    if constexpr (ByThetaOf<Metric>::value) m.comps_th(r, th, s, c, g);
    else m.comps(r, s, c, g);
}
template <class Metric>
GR_DEV void metric_eval(const Metric& m, real r, real th, real s, real c, real g[5], real j1[5], real j2[5], real gi[5])
{
    if constexpr (ByThetaOf<Metric>::value) m.eval_th(r, th, s, c, g, j1, j2, gi);
    else m.eval(r, s, c, g, j1, j2, gi);
}

// ... at a point given by r and (sinθ, cosθ)
template <class Metric>
GR_DEV void geodesic_rhs_generic(const Metric& m, real r, real s, real c, real vt, real vr, real vh, real vp,
                                 real& at, real& ar, real& ah, real& ap)
{
    real g[5], j1[5], j2[5], gi[5];
    m.eval(r, s, c, g, j1, j2, gi);
    geodesic_contract(j1, j2, gi, vt, vr, vh, vp, at, ar, ah, ap);
    if constexpr (Metric::kHasForce) m.add_force(r, s, c, gi, vt, vr, vh, vp, at, ar, ah, ap);
}

// ... at a point given by r and θ, for the metrics that are evaluated there (ByThetaOf)
template <class Metric>
GR_DEV void geodesic_rhs_th(const Metric& m, real r, real th, real s, real c, real vt, real vr, real vh, real vp,
                            real& at, real& ar, real& ah, real& ap)
{
    real g[5], j1[5], j2[5], gi[5];
    m.eval_th(r, th, s, c, g, j1, j2, gi);
    geodesic_contract(j1, j2, gi, vt, vr, vh, vp, at, ar, ah, ap);
}
// ... inside the step loop: `cs` carries the wave's LDS storage (the metric's patch cache, TabLds)
template <class Metric, class Cold_>
GR_DEV void geodesic_rhs_th(const Metric& m, const Cold_& cs, real r, real th, real s, real c, real vt, real vr, real vh, real vp,
                            real& at, real& ar, real& ah, real& ap)
{
    m.rhs_th(cs, r, th, s, c, vt, vr, vh, vp, at, ar, ah, ap);
}

// the right-hand side the integrator calls: the metric's own fused form where it has one
template <class Metric>
GR_DEV void geodesic_rhs_sc(const Metric& m, real r, real s, real c, real vt, real vr, real vh, real vp,
                            real& at, real& ar, real& ah, real& ap)
{
#ifndef GR_NO_FUSED_RHS
    if constexpr (Metric::kFusedRhs) {
        m.rhs(r, s, c, vt, vr, vh, vp, at, ar, ah, ap);
        return;
    }
#endif
    geodesic_rhs_generic(m, r, s, c, vt, vr, vh, vp, at, ar, ah, ap);
}

template <class Metric>
GR_DEV void geodesic_rhs(const Metric& m, real r, real th, real vt, real vr, real vh, real vp,
                         real& at, real& ar, real& ah, real& ap, real& s, real& c)
{
    sincos_fast(th, s, c);
    if constexpr (ByThetaOf<Metric>::value) geodesic_rhs_th(m, r, th, s, c, vt, vr, vh, vp, at, ar, ah, ap);
    else geodesic_rhs_sc(m, r, s, c, vt, vr, vh, vp, at, ar, ah, ap);
}

// constrain_time, auto-diff.jl:161-179
GR_DEV real constrain_time(const real g[5], real vr, real vh, real vp, real mu)
{
    const real disc = -g[0] * g[1] * vr * vr - g[0] * g[2] * vh * vh - g[0] * mu * mu
                        - (g[0] * g[3] - g[4] * g[4]) * vp * vp;
    return -(g[4] * vp + sqrt_fast(disc)) * rcp_full(g[0]);
}

// ---------------------------------------------------------------------------------------
// Tsit5 tableau, dense output
// ---------------------------------------------------------------------------------------
struct Ts {
    static constexpr creal A[7][6] = {
        { 0, 0, 0, 0, 0, 0 },
        { 0.161, 0, 0, 0, 0, 0 },
        { -0.008480655492356989, 0.335480655492357, 0, 0, 0, 0 },
        { 2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0 },
        { 5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0 },
        { 5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0 },
        { 0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774 },
    };
    static constexpr creal BT[7] = { -0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                                      -0.1447110071732629,     0.5823571654525552,     -0.45808210592918697,
                                      0.015151515151515152 };
    // b_i(Θ) = Σ_m R[i][m] Θ^(m+1)
    static constexpr creal R[7][4] = {
        { 1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216 },
        { 0.0, 0.13169999999999998, -0.2234, 0.1017 },
        { 0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253 },
        { 0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902 },
        { 0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928 },
        { 0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661 },
        { 0.0, 1.5, -4.0, 2.5 },
    };
};


// Derived tables for the second-order (x' = v, v' = a) structure of the geodesic ODE.  Stage
// velocities are v_j = v + h Σ_{i<j} a_ji A_i, so every Σ_j w_j v_j that Tsit5 needs for the
// position half of the state collapses to (Σ_j w_j) v + h Σ_i (Σ_{j>i} w_j a_ji) A_i.  Only the
// stage ACCELERATIONS A_i are stored (28 doubles instead of 56); results differ from the
// first-order bookkeeping by rounding only.
struct TsX {
    creal C[7];       // c_s = Σ_j a_sj
    creal AX[7][7];   // ā_si = Σ_{i<j<s} a_sj a_ji
    creal SBT;        // Σ_j b̃_j (zero up to rounding of the published coefficients)
    creal BTX[7];     // Σ_{j>i} b̃_j a_ji
    creal SR[4];      // Σ_j R[j][m]  (1, ~0, ~0, ~0)
    creal RX[7][4];   // Σ_{j>i} R[j][m] a_ji
    creal K2;         // max_j Σ_i |Σ_m RX[i][m] Θ_j^(m+1)| over the sample points Θ_j = j/7
    // Every weighted sum of the step, Σ_q w_q A_q, is formed as w_0 (A_0 + Σ_{q>0} (w_q / w_0) A_q): the chain starts at
    // A_0 without a multiplication, and w_0 joins the factor (h, h²) that multiplies the sum anyway -- one product per
    // stage instead of one per component (31 FP64 instructions per step fewer; the sums differ by rounding only)
    creal AR[7][6];   // a_sq / a_s0
    creal AXR[7][7];  // ā_sq / ā_s0
    creal BTR[7];     // b̃_q / b̃_0
    creal BTXR[7];    // BTX_q / BTX_0
};
constexpr TsX make_tsx()
{
    TsX t{};
    for (int s = 0; s < 7; ++s) {
        creal c = 0.0;
        for (int j = 0; j < s && j < 6; ++j) c += Ts::A[s][j];
        t.C[s] = c;
        for (int i = 0; i < 7; ++i) {
            creal a = 0.0;
            for (int j = i + 1; j < s && j < 6; ++j) a += Ts::A[s][j] * Ts::A[j][i];
            t.AX[s][i] = a;
        }
    }
    creal sb = 0.0;
    for (int j = 0; j < 7; ++j) sb += Ts::BT[j];
    t.SBT = sb;
    for (int i = 0; i < 7; ++i) {
        creal a = 0.0;
        for (int j = i + 1; j < 7; ++j) a += Ts::BT[j] * (i < 6 ? Ts::A[j][i] : 0.0);
        t.BTX[i] = a;
    }
    for (int m = 0; m < 4; ++m) {
        creal sr = 0.0;
        for (int j = 0; j < 7; ++j) sr += Ts::R[j][m];
        t.SR[m] = sr;
        for (int i = 0; i < 7; ++i) {
            creal a = 0.0;
            for (int j = i + 1; j < 7; ++j) a += Ts::R[j][m] * (i < 6 ? Ts::A[j][i] : 0.0);
            t.RX[i][m] = a;
        }
    }
    creal k2 = 0.0;
    for (int jj = 1; jj <= 6; ++jj) {
        const creal th = (creal)jj / 7.0;
        creal sum = 0.0;
        for (int i = 0; i < 7; ++i) {
            const creal b = th * (t.RX[i][0] + th * (t.RX[i][1] + th * (t.RX[i][2] + th * t.RX[i][3])));
            sum += b < 0.0 ? -b : b;
        }
        if (sum > k2) k2 = sum;
    }
    t.K2 = k2;
    for (int s = 1; s < 7; ++s)
        for (int q = 0; q < 6; ++q) t.AR[s][q] = Ts::A[s][q] / Ts::A[s][0];
    for (int s = 2; s < 7; ++s)
        for (int q = 0; q < 7; ++q) t.AXR[s][q] = t.AX[s][q] / t.AX[s][0];
    for (int q = 0; q < 7; ++q) {
        t.BTR[q] = Ts::BT[q] / Ts::BT[0];
        t.BTXR[q] = t.BTX[q] / t.BTX[0];
    }
    return t;
}
struct TsD {
    static constexpr TsX X = make_tsx();
};

// PI controller constants (App. A.3)
constexpr creal PI_BETA1 = 7.0 / 50.0;
constexpr creal PI_BETA2 = 2.0 / 25.0;
constexpr creal PI_GAMMA = 0.9;
constexpr creal PI_QMIN = 0.2;
constexpr creal PI_QMAX = 10.0;
constexpr creal LOG2_QOLDINIT = -13.287712379549449;  // log2(1e-4)
// bounds used to skip the event sampling (see Ray::sample_event); 1e-6 of slack for rounding
constexpr creal DENSE_K1 = 1.000001;
constexpr creal DENSE_K2 = TsD::X.K2 * 1.000001;

// ---------------------------------------------------------------------------------------
// kernel parameter block (uniform, lives in the kernarg segment / SGPRs)
// ---------------------------------------------------------------------------------------
struct PfDev {
    int32_t pf_id, filter_id;
    double fill, r_isco;
    int64_t n_plunge;
    const double* plunge_r;   // device
    const double* plunge_vt;
    const double* plunge_vr;
    const double* plunge_vphi;
    int32_t has_u_src;        // 1 = E_start is measured against u_src (energy_ratio, flux-calculations.jl:96-110) instead of (1, 0, 0, 0)
    int32_t pad_u;
    double u_src[4];
};

// Per-launch data that only init()/finalize() touch.  It lives in device memory behind a pointer
// (not in the kernarg segment) and is re-read at each use, so that its ~60 scalars are not kept
// live in SGPRs across the hot step loop.
struct Cold {
    int32_t src_mode;         // 0 = image plane, 1 = (x, v) arrays, 2 = impact-parameter arrays (3 = a source's sky, gr_rayset.sky_*: on the host side only --
                              // launch_trace has a small kernel write the sky's velocities and hands the trace kernels src_mode 1)
    int32_t out_mode;         // 0 = fused point function image, 1 = endpoint records, 2 = binned line profile, 3 = (g, ρ) pairs, 4 = (g, ρ, t, status), 5 = the same with ∂/∂α, ∂/∂β (tangent build only)
    int32_t swizzle;          // log2 rows of the pixel tile a wave owns: 3 = 8 x 8, 4 = 16 x 4 (0 = none)
    int32_t idx32;            // 1 = every ray / pixel index fits 31 bits: 32-bit divisions in the index maps
    gr_plane plane;
    gr_range range;
    const double* x;          // device
    int64_t x_stride;
    const double* v;          // device
    double* image;            // device
    gr_point* points;         // device
    PfDev pf;
    // longest-first scheduling of 8x8 tiles: `tile_perm` (may be null) maps queue order -> tile;
    // `tile_cost` (may be null) receives the step count of one representative ray per tile (the lane kernel of a tabulated
    // metric: the wave's lifetime in 160 ns units, which its step count does not predict -- gr_kernels.hpp) so
    // the host can build the permutation for the next render of the same plane
    const uint32_t* tile_perm;
    uint32_t* tile_cost;
    // src_mode 2: rays given by impact parameters (an AbstractImagePlane, image-planes/planes.jl:180-184)
    const double* alpha;      // device, n
    const double* beta;       // device, n
    const double* area;       // device, n (unnormalized_areas) or null = 1
    const double* height;     // device, n per-ray DatumPlane heights or null = cfg.disc_params[0]
    // separable ray set (gr_rayset.sep_*: a PolarPlane as three small tables): α = r_i cos θ_j, β = r_i sin θ_j, area = r_i²
    const double* sep_r;      // device, sep_nr (null: alpha / beta / area arrays)
    const double* sep_cos;    // device, sep_nt
    const double* sep_sin;    // device, sep_nt
    int64_t sep_nr, sep_nt;
    int64_t sep_first;        // local ray jl is ray sep_first + (jl / sep_block) sep_stride + jl % sep_block of the set
    int64_t sep_block;        // (0: sep_first + jl)
    int64_t sep_stride;
    int64_t sep_core_rows;    // sep_tiled: rows / columns covered by whole 8 x 8 tiles (0, 0 = column-major order)
    int64_t sep_core_cols;
    double winding_plane;     // TraceWindings.plane_inc (cfg.count_windings)
    // out_mode 2: BinningMethod line profile (line-profiles.jl:152-198) fused into finalize;
    // out_mode 3: (g, ρ) pairs for a host-side emissivity
    double lp_rmin, lp_rmax;  // minrₑ, maxrₑ
    double lp_q;              // ε(r) = r^-q
    const double* lp_eps_r;   // device, lp_eps_n radii of a tabulated emissivity (RadialDiscProfile), or null
    const double* lp_eps_v;   // device, lp_eps_n values
    int64_t lp_eps_n;
    int64_t lp_nbins;
    const double* lp_edges;   // device, lp_nbins
    double* lp_flux;          // device, lp_nbins (accumulated with fp64 atomics)
    double* lp_pairs;         // device, n x 2
    // out_mode 0 / 1 of an image plane: 1 = ray j's pixel / record goes to index range_map(j) -- its place in the WHOLE image --
    // instead of to local index j (several devices storing one plane into one page-locked host block, gr_*_multi)
    int32_t out_global;
    int32_t out_reserved;
    // src_mode 3: rays from a source into its sky (corona-models.jl:1-33, samplers.jl:30-99)
    int32_t sky_sampler, sky_both, sky_generator, sky_reserved;
    double sky_resolution;
    const double* sky_i;      // device, n (sky_generator 2) or null
};

// the error norm's scaled residuals in single precision: the fp64 device kernels only (see Ray::step)
#ifndef GR_NORM_F32
#define GR_NORM_F32 1
#endif
#if GR_NORM_F32 && !defined(GR_REAL_IS_FLOAT) && !defined(GR_REAL_IS_TAN2)
#define GR_NORM_F32_ON 1
#else
#define GR_NORM_F32_ON 0
#endif

constexpr int N_STAT = 9;   // statistics counters of a launch: rays, accepted, rejected, rhs, flagged, status[4]

struct Params {
    gr_config cfg;
    const Cold* cold;         // device
    const double* disc_table; // device copy of cfg.disc_table (GR_DISC_TABULATED)
    const double* chart_table; // device copy of cfg.chart_table (PoloidalShapeChart); cfg.upper_hemisphere bit 1 set
    int64_t n;                // rays in this call
    unsigned long long* stats;  // device: 9 counters (see gr_stats order), may be null
    unsigned long long* queue;  // device: persistent-kernel work counter
    int32_t refill_threshold;
    int32_t lds_plunge_rows;  // rows of the plunging table to stage in LDS (0 = none)
    int32_t lds_bins;         // line-profile bins privatised in LDS (0 = none)
    int32_t maxiters32;       // cfg.maxiters clamped to int32: the per-step test is one 32-bit compare
    double wedge;             // asin(gtol) with a hair of slack: |θ - π/2| beyond it cannot hit the disc
    double dtmax;             // |λ1 - λ0|, formed once on the host instead of once per step per lane
    int32_t tangent_norm;     // tangent build only: 1 = the error norm runs over values AND tangents (gr_ctx_set "tangent_norm")
    int32_t lds_points;       // one-ray-per-lane kernel, end-point output: 1 = a wave's 152-B records leave through LDS as whole
                              // runs (POINT_UNITS doubles + one address slot per lane behind the other LDS regions)
    int32_t xcd_spread;       // one-ray-per-lane kernel on rays in CALLER order: 1 = workgroup b traces chunk xcd_chunk(b) of the
    int32_t lds_tab_off;      // rays instead of chunk b (gr_kernels.hpp) | byte offset of the tabulated metric's patch caches in the workgroup's LDS
};

// the derived fields of Params, from cfg (host side; one place for the library and the two host harnesses)
static inline void derive_params(Params& p)
{
    const double g = p.cfg.gtol < 1.0 ? p.cfg.gtol : 1.0;
    p.wedge = ::asin(g) * (1.0 + 1e-9) + 1e-12;
    const double span = p.cfg.lambda1 - p.cfg.lambda0;
    p.dtmax = span < 0.0 ? -span : span;
    int64_t mi = p.cfg.maxiters < 0 ? 0 : p.cfg.maxiters;
    p.maxiters32 = (int32_t)(mi > 0x7fffffff ? 0x7fffffff : mi);
}

// Small read-mostly tables staged in LDS by the kernel prologue (null = use the global copy):
// the PlungingInterpolation table of the non-Kerr redshift and the per-workgroup private copy of
// the line-profile histogram.
struct LdsView {
    const double* pl_r;
    const double* pl_vt;
    const double* pl_vr;
    const double* pl_vp;
    double* hist;
    double* point;            // this lane's record inside its wave's region (POINT_UNITS doubles), or null: direct stores
    uint64_t* point_addr;     // this lane's slot for the record's destination address (0 = nothing to store)
};
constexpr int POINT_UNITS = 19;     // sizeof(gr_point) / 8
static_assert(sizeof(gr_point) == 8 * POINT_UNITS, "gr_point is 19 eight-byte units");

// Lanes per ray and this lane's tangent direction.  GR_TAN_W = 1 on the device: rays are traced by PAIRS of neighbouring
// lanes (gr_tangent.hpp), the even lane carries ∂/∂α, the odd one ∂/∂β; local ray index = global work-item index >> 1.
#if defined(GR_REAL_IS_TAN2) && GR_TAN_W == 1 && !defined(GR_HOST_HARNESS)
constexpr int LANES_PER_RAY_LOG2 = 1;
GR_DEV int tan_dir() { return (int)(threadIdx.x & 1u); }
#else
constexpr int LANES_PER_RAY_LOG2 = 0;
GR_DEV int tan_dir() { return 0; }
#endif

// read the cold block through a pointer the optimiser cannot hoist loads from
GR_DEV const Cold& cold_of(const Params& p)
{
    const Cold* c = p.cold;
#ifndef GR_HOST_HARNESS
    asm volatile("" : "+s"(c));
#endif
    return *c;
}

// local ray index -> swizzled local index so that 64 consecutive work items cover an 8x8 tile
// integer division in the index maps: 64-bit division costs ~100 instructions on the VALU, 32-bit ~20
GR_DEV int64_t idx_div(const Cold& p, int64_t a, int64_t b)
{
    return p.idx32 ? (int64_t)((uint32_t)a / (uint32_t)b) : a / b;
}

// `swizzle` = log2 of the tile's ROWS (consecutive pixels of a column, i.e. consecutive doubles of the image): 3 = 8 x 8
// tiles, 4 = 16 rows x 4 columns -- a wave's stores are then four whole 128-byte lines instead of eight half lines
// (gr_ctx_set "tile_rows").  0 = no tiling.
GR_DEV int64_t tile_swizzle(const Cold& p, int64_t j)
{
    if (!p.swizzle) return j;
    const int tr = p.swizzle, tc = 6 - tr;
    const int64_t H = p.plane.height;
    int64_t tile = j >> 6;
    if (p.tile_perm) tile = p.tile_perm[tile];
    const int lane = (int)(j & 63);
    const int64_t tiles_per_col = H >> tr;
    const int64_t tx = idx_div(p, tile, tiles_per_col), ty = tile - tx * tiles_per_col;
    return ((tx << tc) + (lane >> tr)) * H + (ty << tr) + (lane & ((1 << tr) - 1));
}

GR_DEV int64_t range_map(const Cold& p, int64_t j)
{
    const gr_range& rg = p.range;
    const int64_t b = idx_div(p, j, rg.block);
    return rg.first + b * rg.stride_blocks * rg.block + (j - b * rg.block);
}

GR_DEV real range_at(real a, real b, int64_t n, int64_t k)
{
    if (n <= 1) return a;
    const real t = (real)k / (real)(n - 1);
    return (1.0 - t) * a + t * b;
}

// ---------------------------------------------------------------------------------------
// point functions on a finished ray
// ---------------------------------------------------------------------------------------
GR_DEV real kerr_plunge_Le(real M, real rms, real a)
{
    // Lₑ, redshift.jl:93
    return GR_SQRT(M) * (rms * rms - 2.0 * a * GR_SQRT(M * rms) + a * a)
           / (rms * GR_SQRT(rms) - 2.0 * M * GR_SQRT(rms) + a * GR_SQRT(M));
}

GR_DEV real nan_linear_interp(const double* t, const double* y, int64_t n, real x)
{
    // NaNLinearInterpolator, interpolations.jl:7-29
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (t[mid] <= x) lo = mid + 1; else hi = mid;
    }
    int64_t idx = lo < 1 ? 1 : lo;
    if (idx > n - 1) idx = n - 1;
    const real x1 = t[idx - 1], x2 = t[idx], y1 = y[idx - 1], y2 = y[idx];
    const real w = (x - x1) / (x2 - x1);
    const real v = (1.0 - w) * y1 + w * y2;
    if (!(v == v)) {
        if (w < 0.5) return (y1 == y1) ? y1 : 0.0;
        return (y2 == y2) ? y2 : 0.0;
    }
    return v;
}

// CircularOrbits.fourvelocity(m, ρ) at θ = π/2, circular-orbits.jl:11-37,58-61,114-121
template <class Metric>
GR_DEV void circular_fourvelocity(const Metric& m, real rho, real& vt, real& vp)
{
    real g[5], j1[5], j2[5], gi[5];
    metric_eval(m, rho, (real)1.5707963267948966, (real)1.0, (real)0.0, g, j1, j2, gi);
    const real Dl = sqrt_fast(j1[4] * j1[4] - j1[0] * j1[3]);
    const real Om = -(j1[4] - Dl) * rcp_full(j1[3]);
    const real A = -(Om * gi[0] - gi[4]);
    const real B = (Om * gi[4] - gi[3]);
    const real den = B * B * gi[0] + 2.0 * A * B * gi[4] + A * A * gi[3];
    const real d = -(real)sgn(den) * sqrt_fast(rcp_full(GR_FABS(den)));
    const real ut = B * d, up = A * d;
    vt = gi[0] * ut + gi[4] * up;
    vp = gi[4] * ut + gi[3] * up;
}

// redshift_function(m, gp) / interpolate_redshift closure; redshift.jl:192-220,246-276
template <class Metric>
GR_DEV real redshift_pf(const Metric& m, const Params& pp, const Cold& p, const LdsView& lds, const real x0[4],
                        const real v0[4], const real x[4], const real v[4])
{
    real s, c;
    sincos_fast(x[2], s, c);
    const real rho = x[1] * GR_FABS(s);
    real dt_, dr_, dp_;
    const bool kerr_analytic = (pp.cfg.metric_id == GR_METRIC_KERR) && (p.pf.n_plunge == 0);
    const real isco = p.pf.r_isco;
    if (rho < isco) {
        if (kerr_analytic) {
            const real M = pp.cfg.params[0], a = pp.cfg.params[1];
            const real Le = kerr_plunge_Le(M, isco, a);
            const real H = (2.0 * M * rho - a * Le) / (rho * rho - 2.0 * M * rho + a * a);
            const real ge = GR_SQRT(1.0 - (2.0 * M) / (3.0 * isco));
            const real q = isco / rho - 1.0;
            const real ur = -GR_SQRT((2.0 * M) / (3.0 * isco)) * q * GR_SQRT(q);
            dt_ = ge * (1.0 + 2.0 * M * (1.0 + H) / rho);
            dr_ = -ur;
            dp_ = ge / (rho * rho) * (Le + a * H);
        } else {
            real rb = rho;
            const int64_t n = p.pf.n_plunge;
            const double* tr = lds.pl_r ? lds.pl_r : p.pf.plunge_r;
            const double* tvt = lds.pl_r ? lds.pl_vt : p.pf.plunge_vt;
            const double* tvr = lds.pl_r ? lds.pl_vr : p.pf.plunge_vr;
            const double* tvp = lds.pl_r ? lds.pl_vp : p.pf.plunge_vphi;
            if (rb < tr[0]) rb = tr[0];
            if (rb > tr[n - 1]) rb = tr[n - 1];
            dt_ = nan_linear_interp(tr, tvt, n, rb);
            dr_ = -nan_linear_interp(tr, tvr, n, rb);
            dp_ = nan_linear_interp(tr, tvp, n, rb);
        }
    } else {
        circular_fourvelocity(m, rho, dt_, dp_);
        dr_ = 0.0;
    }
    // _redshift_dotproduct: E_obs / E_disc with v_obs = (1,0,0,0)
    real g[5];
    metric_comps(m, x[1], x[2], s, c, g);
    const real E_disc = (g[0] * v[0] + g[4] * v[3]) * dt_ + g[1] * v[1] * dr_ + (g[4] * v[0] + g[3] * v[3]) * dp_;
    real s0, c0, g0[5];
    sincos_fast(x0[2], s0, c0);
    metric_comps(m, x0[1], x0[2], s0, c0, g0);
    real E_obs = g0[0] * v0[0] + g0[4] * v0[3];
    if (p.pf.has_u_src) {
        // energy_ratio (flux-calculations.jl:96-110): the photon's energy at its start in the frame of a moving source
        const real u0 = p.pf.u_src[0], u1 = p.pf.u_src[1], u2 = p.pf.u_src[2], u3 = p.pf.u_src[3];
        E_obs = (g0[0] * v0[0] + g0[4] * v0[3]) * u0 + g0[1] * v0[1] * u1 + g0[2] * v0[2] * u2 + (g0[4] * v0[0] + g0[3] * v0[3]) * u3;
    }
    return E_obs * rcp_full(E_disc);
}

// ---------------------------------------------------------------------------------------
// Cold lane storage: LDS as the spill space the compiler does not have.
//
// The step's hot region (five stages + the right-hand side at the new state) needs r, θ, the four velocities and the
// stage accelerations; the rest of the ray -- t, dt, x^t, x^ϕ, the disc condition at the step's start, the controller's
// memory and the counters -- is touched before and after it only.  At the register budget of three waves per SIMD
// (168 VGPRs) the compiler spilled into scratch (56-88 B per lane, HBM-backed: 44-60 MB of write-back per launch).  A
// lane-private LDS slot costs neither HBM traffic nor a VALU instruction: the cold values are parked at the top of
// step() and reloaded behind the last right-hand side (9 ds_write_b64 + 9 ds_read_b64 per step on a unit that is
// otherwise idle; 72 B per lane = 4.6 KB per wave of the CU's 160 KB).  NoColdStore keeps everything in registers
// (host harness, tangent flavour, k_trace_path).
// ---------------------------------------------------------------------------------------
constexpr int COLD_SLOTS = 9;      // 8-byte slots per lane
struct NoColdStore {
    static constexpr bool kOn = false;
    static constexpr bool kHead = false;
    static constexpr int kParkA = 0;
    static constexpr bool kTabLds = false;      // (a tabulated metric's patch cache: TabLds below)
};
template <bool HEAD>
struct LdsColdStoreT {
    static constexpr bool kTabLds = false;
    static constexpr bool kOn = true;       // parked around the event sampling (the rarely taken branch)
    static constexpr bool kHead = HEAD;     // ... and across the whole hot region of every step
    static constexpr int kParkA = 0;        // (stage accelerations: ParkA below)
    static constexpr int kStride = 64;     // one wave per region: slot k of the wave's lanes is one conflict-free 512-byte row,
                                           // and k * 512 is an immediate offset of the ds instruction (no address arithmetic)
    double* lane;        // this lane's slot 0 inside its wave's region; slot k is lane[k * 64]
    template <class T> GR_DEV void st(int k, T v) const { *reinterpret_cast<T*>(lane + k * kStride) = v; }
    template <class T> GR_DEV T ld(int k) const { return *reinterpret_cast<const T*>(lane + k * kStride); }
    GR_DEV void st2(int k, int32_t a, int32_t b) const
    {
        reinterpret_cast<int32_t*>(lane + k * kStride)[0] = a;
        reinterpret_cast<int32_t*>(lane + k * kStride)[1] = b;
    }
    GR_DEV void ld2(int k, int32_t& a, int32_t& b) const
    {
        a = reinterpret_cast<const int32_t*>(lane + k * kStride)[0];
        b = reinterpret_cast<const int32_t*>(lane + k * kStride)[1];
    }
    // the compiler may neither forward a parked value to its reload nor move the accesses into the hot region
    static GR_DEV void fence()
    {
#ifndef GR_HOST_HARNESS
        asm volatile("" ::: "memory");
#endif
    }
};
typedef LdsColdStoreT<true> LdsColdStore;
typedef LdsColdStoreT<false> LdsColdStoreRare;

// STAGE ACCELERATIONS PARKED IN LDS (round 4).  A[1..5] are written once per stage and read only by the sums of the later
// stages, the error estimate and the dense output -- never inside a right-hand side, which is where the register pressure
// peaks.  With ParkA<Base, N> the accelerations of stages 1..N leave for LDS as soon as they are formed (Ray::step, GR_PARK) and
// come back, behind a compiler fence, where a sum is about to read them (GR_UNPARK): no value of A[1..N] is alive across a
// right-hand side.  N x 4 scalars per lane: 32 B per stage for the fp64 kernels (Johannsen: 3 stages = 6 KB per wave, 72 KB per
// CU at three waves per SIMD), 64 B for the tangent build's (value, ∂) pairs (4 stages = 16 KB per wave, 128 KB at two).
// Element (q, i) of the wave is 64 consecutive scalars: one conflict-free ds_read / ds_write per element.
template <class Base, int NPARK>
struct ParkA : Base {
    static constexpr int kParkA = NPARK;
    real* park;          // this lane's element (1, 0); element (q, i) is park[((q - 1) * 4 + i) * 64]
    GR_DEV void stA(int q, int i, real v) const { park[((q - 1) * 4 + i) * 64] = v; }
    GR_DEV real ldA(int q, int i) const { return park[((q - 1) * 4 + i) * 64]; }
    static GR_DEV void park_fence()
    {
#ifndef GR_HOST_HARNESS
        asm volatile("" ::: "memory");      // no forwarding of a parked value to its reload, no reload hoisted above a right-hand side
#endif
    }
};

// GR_METRIC_TABULATED: the wave's patch cache rides along with the other per-wave LDS storage of the step loop
template <class Base>
struct TabLds : Base {
    static constexpr bool kTabLds = true;
#ifdef GR_HOST_HARNESS
    char* tab;
#else
    char __attribute__((address_space(3)))* tab;      // this wave's region (kTabLdsBytesPerWave bytes)
#endif
};

// ---------------------------------------------------------------------------------------
// The per-lane integrator.
// ---------------------------------------------------------------------------------------
enum : int32_t { RAY_EVENT = 0x100 };   // bit in Ray::flags while a disc event awaits its root find

template <class Metric, int DISC>
struct Ray {
    real x[4];        // (t, r, θ, ϕ) at the start of the current step
    real v[4];        // (v^t, v^r, v^θ, v^ϕ)
    real A[7][4];     // stage accelerations; A[0] is FSAL
    real t;           // affine time
    hreal dt, h;      // proposed step, last used step
    real cprev;       // disc condition at x
    real sth, cth;    // sin θ, cos θ at x (base of the stage rotations)
    RotK rotk;        // register-resident constants of the stage rotations
#ifdef GR_CONTROLLER_F64
    double lq_old;      // log2(qold)
#else
    float lq_old;       // log2(qold)
#endif
    int32_t ev_top;     // upper bracket j of Θ = j/7 when an event is pending
    int64_t j;          // local (swizzled) ray index
    int32_t status, flags;      // flags: GR_FLAG_* | RAY_EVENT in bits 0..15, TraceWindings count in bits 16..31
    int32_t nacc, nrej;
    real hdat;          // GR_DISC_DATUM: this ray's plane height (dead in every other instantiation)
    // GR_DISC_COMPOSITE (dead in every other instantiation): the conditions of components 1.. at x (component 0's is cprev) and
    // which components the pending event belongs to
    real cprev_more[GR_COMP_MAX - 1];
    int32_t ev_mask;
    // GR_DISC_MESH (dead in every other instantiation): the Cartesian position at the start of the step and at its end; whether
    // the end lies inside the mesh's bounding box (a test is due); whether the user's or the chart's callback ended the ray at
    // that step; and whether the kernel runs the due tests wave-wide after the step (mesh_phase) instead of inside it
    real qprev[3], qnew[3];
    int32_t mesh_need, mesh_cb_term, mesh_coop;
#ifdef GR_HOST_HARNESS
    real dbg_e2;
    real dbg_dmax;          // per attempted step: largest |θ_stage - θ_base| (harness statistics)
    mutable int dbg_bits;   // per attempted step: bit s = stage s took the full sincos; 8 = event sampling past the reach bound; 9 = past the θ samples
#endif

    // a geometry with a ContinuousCallback (distance_to_disc); a mesh is a DiscreteCallback on the step's line element instead
    static constexpr bool kContinuous = (DISC != GR_DISC_NONE && DISC != GR_DISC_MESH);

#if GR_HAS_MESH
    // ---- MeshAccretionGeometry (geometry/meshes.jl:1-80): after every accepted step the Cartesian line element
    // (to_cartesian(u_prev), to_cartesian(u)) (geometry.jl:13-16,38-40) is tested against the triangles -- intersects_geometry
    // (intersections.jl:7-16) = in_nearby_region && has_intersect; the ray ends AT the step's end (no root finding) with
    // IntersectedWithGeometry.  The triangles sit in HBM behind p.disc_table, sorted into a uniform grid (gr_mesh_grid.hpp).
    static GR_DEV void to_cartesian3(real r, real s, real c, real ph, real q[3])
    {
        real sp, cp;
        sincos_fast(ph, sp, cp);
        const real rs = r * s;
        q[0] = rs * cp;
        q[1] = rs * sp;
        q[2] = r * c;
    }
    static GR_DEV void cross3(const real a[3], const real b[3], real o[3])
    {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    }
    static GR_DEV real dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    // jsf_algorithm (intersections.jl:58-101; Jiménez, Segura & Feito 2010), ϵ = 1e-8: does the segment Q1 -> Q2 pass through the
    // triangle from its front side (w > 0: Q1 in front of the plane, Q2 not); a segment that starts behind the plane never hits
    static GR_DEV bool jsf_hit(const real V1[3], const real V2[3], const real V3[3], const real Q1[3], const real Q2[3])
    {
        const real eps = 1e-8;
        real A[3], B[3], C[3], D[3], W1[3], W2[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { A[i] = Q1[i] - V3[i]; B[i] = V1[i] - V3[i]; C[i] = V2[i] - V3[i]; D[i] = Q2[i] - V3[i]; }
        cross3(B, C, W1);
        const real w = dot3(A, W1);
        if (w > eps) {
            const real sg = dot3(D, W1);
            if (sg > eps) return false;
            cross3(A, D, W2);
            const real tt = dot3(W2, C);
            if (tt < -eps) return false;
            const real uu = -dot3(W2, B);
            if (uu < -eps) return false;
            if (w < sg + tt + uu) return false;
            return true;
        } else if (w < -eps) {
            return false;
        }
        const real sg = dot3(D, W1);       // Q1 in the triangle's plane
        if (sg > eps) return false;
        if (sg < -eps) {
            cross3(D, A, W2);
            const real tt = dot3(W2, C);
            if (tt > eps) return false;
            const real uu = -dot3(W2, B);
            if (uu > eps) return false;
            if (-sg > tt + uu) return false;
            return true;
        }
        return false;
    }
    // p.disc_table: the table gr_mesh_grid.hpp builds -- the bounding box, a uniform grid over the triangles' first vertices
    // with cells of (just over) 3, the triangles sorted by cell (first vertices in one run, the other two in a second)
    // in_nearby_region (meshes.jl:46-51): the step's END strictly inside the bounding box
    static GR_DEV bool mesh_inside(const Params& p, const real Q2[3])
    {
        const double* tb = p.disc_table;
        return (real)tb[0] < Q2[0] && Q2[0] < (real)tb[1] && (real)tb[2] < Q2[1] && Q2[1] < (real)tb[3] && (real)tb[4] < Q2[2] && Q2[2] < (real)tb[5];
    }
    // the grid cells around a point: per axis the cell of the point and its two neighbours, clipped to the grid (the box may
    // reach beyond the grid of first vertices: an empty range then)
    static GR_DEV void mesh_cells(const double* tb, const real Q2[3], int lo[3], int hi[3])
    {
        const real icell = (real)tb[9];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int dim = (int)tb[10 + a];
            real f = GR_FLOOR((Q2[a] - (real)tb[6 + a]) * icell);
            f = GR_FMAX((real)-2.0, GR_FMIN(f, (real)dim + (real)1.0));
            const int i = (int)f;
            lo[a] = i - 1 < 0 ? 0 : i - 1;
            hi[a] = i + 1 > dim - 1 ? dim - 1 : i + 1;
        }
    }
    // has_intersect (meshes.jl:53-64) for one candidate: the triangle counts if its FIRST vertex is within 3 of the step's end
    static GR_DEV bool mesh_candidate(const Params& p, const double* T0, uint32_t k, const real Q1[3], const real Q2[3])
    {
        const double* T = T0 + 3 * (int64_t)k;
        const real V1[3] = { (real)T[0], (real)T[1], (real)T[2] };
        const real dx = V1[0] - Q2[0], dy = V1[1] - Q2[1], dz = V1[2] - Q2[2];
        if (!(dx * dx + dy * dy + dz * dz < 9.0)) return false;
        const double* W = T0 + 3 * (int64_t)p.cfg.disc_table_n + 6 * (int64_t)k;
        const real V2[3] = { (real)W[0], (real)W[1], (real)W[2] };
        const real V3[3] = { (real)W[3], (real)W[4], (real)W[5] };
        return jsf_hit(V1, V2, V3, Q1, Q2);
    }
    // ONE lane walks the candidates of its own step: the point's cell and its 26 neighbours; per (y, z) neighbour the three
    // x-cells are one run of triangles.  (The persistent and path kernels, and the lane kernel when many lanes of a wave have a
    // test due at once.)
    static GR_DEV bool mesh_lane_query(const Params& p, const real Q1[3], const real Q2[3])
    {
        const double* tb = p.disc_table;
        const int nx = (int)tb[10], ny = (int)tb[11];
        const uint32_t* cs = reinterpret_cast<const uint32_t*>(tb + 16);
        const double* T0 = tb + (int64_t)tb[13];
        int lo[3], hi[3];
        mesh_cells(tb, Q2, lo, hi);
        if (hi[0] < lo[0]) return false;
        for (int iz = lo[2]; iz <= hi[2]; ++iz)
            for (int iy = lo[1]; iy <= hi[1]; ++iy) {
                const int64_t row = ((int64_t)iz * ny + iy) * nx;
                const uint32_t k1 = cs[row + hi[0] + 1];
                for (uint32_t k = cs[row + lo[0]]; k < k1; ++k)
                    if (mesh_candidate(p, T0, k, Q1, Q2)) return true;
            }
        return false;
    }
#ifndef GR_HOST_HARNESS
    // The WAVE walks the candidates of one lane's step (Q1, Q2 are the same in every lane): the up to nine runs of triangles are
    // laid end to end and dealt to the 64 lanes, so a step with ~100 candidates costs two rounds of coalesced loads instead
    // of a hundred dependent ones.  What the tail of a mesh launch is made of: a few rays that wind round the hole inside the
    // bounding box, alone in their waves (DESIGN_measurements.md §M15).
    static GR_DEV bool mesh_wave_query(const Params& p, const real Q1[3], const real Q2[3], int lane)
    {
        const double* tb = p.disc_table;
        const int nx = (int)tb[10], ny = (int)tb[11];
        const uint32_t* cs = reinterpret_cast<const uint32_t*>(tb + 16);
        const double* T0 = tb + (int64_t)tb[13];
        int lo[3], hi[3];
        mesh_cells(tb, Q2, lo, hi);
        if (hi[0] < lo[0]) return false;
        uint32_t first[9], upto[10];
        upto[0] = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int iz = lo[2] + q / 3, iy = lo[1] + q % 3;
            uint32_t k0 = 0, k1 = 0;
            if (iz <= hi[2] && iy <= hi[1]) {
                const int64_t row = ((int64_t)iz * ny + iy) * nx;
                k0 = cs[row + lo[0]];
                k1 = cs[row + hi[0] + 1];
            }
            first[q] = k0;
            upto[q + 1] = upto[q] + (k1 - k0);
        }
        for (uint32_t base = 0; base < upto[9]; base += 64) {
            const uint32_t j = base + (uint32_t)lane;
            bool h = false;
            if (j < upto[9]) {
                uint32_t k = first[0] + j;
#pragma unroll
                for (int q = 1; q < 9; ++q)
                    if (j >= upto[q]) k = first[q] + (j - upto[q]);
                h = mesh_candidate(p, T0, k, Q1, Q2);
            }
            if (__ballot(h)) return true;
        }
        return false;
    }
#endif
#endif

    // distance_to_disc(::DatumPlane), datum-plane.jl:6-10 ; distance_to_disc(::ThinDisc), thin-disc.jl:20-26 ; distance_to_disc(::AbstractThickAccretionDisc),
    // thick-disc.jl:60-66 with cross_section(::ShakuraSunyaev), shakura-sunyaev.jl:28-33
    GR_DEV real disc_cond(const Params& p, real r, real s, real c) const
    {
        if (DISC == GR_DISC_DATUM) return r * c - hdat;   // signed: no underside
        if (DISC == GR_DISC_ELLIPTICAL) {
            // distance_to_disc(::EllipticalDisc), geometry/discs.jl:57-72 (the radial test is on r itself there)
            const real a = p.cfg.disc_params[0], b = p.cfg.disc_params[1];
            if (a < r || r < (real)p.cfg.disc_r_in) return 1.0;
            const real q = r * rcp_full(a);
            const real y = sqrt_fast(GR_FMAX((1.0 - q * q) * b * b, 0.0));
            return GR_FABS(r * c) - y - p.cfg.gtol * GR_FABS(r);
        }
        const real rho = r * GR_FABS(s);
        if (DISC == GR_DISC_THIN || DISC == GR_DISC_PRECESSING_THIN) {
            if (rho < p.cfg.disc_r_in || rho > p.cfg.disc_r_out) return 1.0;
            return r * GR_FABS(c) - p.cfg.gtol * GR_FABS(r);
        }
        real height;
        if (DISC == GR_DISC_TABULATED) {
            // cross_section(::ThickDisc) sampled on a uniform grid (thick-disc.jl:57-58)
            const real r0 = p.cfg.disc_params[0], r1 = p.cfg.disc_params[1];
            if (rho < r0 || rho > r1 || rho < (real)p.cfg.disc_r_in || rho > (real)p.cfg.disc_r_out) return 1.0;
            const int64_t n = p.cfg.disc_table_n;
            const real u = (rho - r0) * ((real)(n - 1) * rcp_full(r1 - r0));
            int64_t k = (int64_t)u;
            if (k > n - 2) k = n - 2;
            const real w = u - (real)k;
            height = (1.0 - w) * (real)p.disc_table[k] + w * (real)p.disc_table[k + 1];
            // WarpedThinDisc (thin-disc.jl:42-66): the table is the signed height h(ρ) of a thin sheet,
            // hit from either side within the thin disc's tolerance wedge
            if (p.cfg.disc_params[3] != 0.0) return GR_FABS(height - r * c) - p.cfg.gtol * GR_FABS(r);
        } else {
            const real rin = p.cfg.disc_r_in;
            if (rho < rin) return 1.0;
            height = 3.0 * (real)p.cfg.disc_params[1] * (real)p.cfg.disc_params[0] * (1.0 - sqrt_fast(rin * rcp_full(rho)));
        }
        if (height <= 0.0) return 1.0;
        return r * GR_FABS(c) - height;
    }

    // the geometry condition at a full position: axisymmetric discs ignore ϕ; PrecessingDisc(ThinDisc, β, γ)
    // (geometry/discs.jl:74-96) rotates the direction by Rx(-β) after shifting ϕ by γ and hands (r, θ') to the
    // thin disc.  disc_params = {β, γ, cos β, sin β}.
    GR_DEV real disc_cond4(const Params& p, real r, real s, real c, real phi) const
    {
        if (DISC != GR_DISC_PRECESSING_THIN) return disc_cond(p, r, s, c);
        real sp, cp;
        sincos_fast(phi - (real)p.cfg.disc_params[1], sp, cp);
        const real cb = p.cfg.disc_params[2], sb = p.cfg.disc_params[3];
        const real v1 = s * sp, v2 = s * cp;
        const real x2 = cb * v2 - sb * c, x3 = sb * v2 + cb * c;     // R = [1 0 0; 0 cβ -sβ; 0 sβ cβ]
        return disc_cond(p, r, sqrt_fast(GR_FMA(v1, v1, x2 * x2)), x3);
    }

    // ---- CompositeGeometry (src/geometry/composite.jl; geometry_collision_callback(::CompositeGeometry), bootstrap.jl:76-110):
    // a VectorContinuousCallback whose k-th condition is component k's distance_to_disc and whose every affect terminates with
    // IntersectedWithGeometry.  DiffEqBase (third party, as published: determine_event_occurance / find_callback_time for
    // VectorContinuousCallback): sign(c_k) at the step's start is remembered per component; a component has an event at the
    // step's end if its sign changed (or hit zero); unless EVERY component has one, the seven interior samples of the dense
    // output are scanned in order and the first sample at which any component's sign differs from its start decides: the
    // components that changed THERE are the candidates and that sample the bracket's top (no interior change: the end-of-step
    // candidates stand).  Each candidate's root is found on the bracket, the earliest one is the event.
    // Components: thin disc, Shakura-Sunyaev, elliptical disc, datum plane (formulas of disc_cond above, per-component fields).
    GR_DEV real comp_cond(const Params& p, int k, real r, real s, real c) const
    {
        const gr_disc_component& g = p.cfg.comp[k];
        if (g.disc_id == GR_DISC_DATUM) return r * c - (real)g.disc_params[0];
        if (g.disc_id == GR_DISC_ELLIPTICAL) {
            const real a = g.disc_params[0], b = g.disc_params[1];
            if (a < r || r < (real)g.disc_r_in) return 1.0;
            const real q = r * rcp_full(a);
            const real y = sqrt_fast(GR_FMAX((1.0 - q * q) * b * b, 0.0));
            return GR_FABS(r * c) - y - p.cfg.gtol * GR_FABS(r);
        }
        const real rho = r * GR_FABS(s);
        if (g.disc_id == GR_DISC_THIN) {
            if (rho < g.disc_r_in || rho > g.disc_r_out) return 1.0;
            return r * GR_FABS(c) - p.cfg.gtol * GR_FABS(r);
        }
        const real rin = g.disc_r_in;                       // GR_DISC_SHAKURA_SUNYAEV
        if (rho < rin) return 1.0;
        const real height = 3.0 * (real)g.disc_params[1] * (real)g.disc_params[0] * (1.0 - sqrt_fast(rin * rcp_full(rho)));
        if (height <= 0.0) return 1.0;
        return r * GR_FABS(c) - height;
    }
    GR_DEV real& comp_prev(int k) { return k == 0 ? cprev : cprev_more[k > 0 ? k - 1 : 0]; }
    GR_DEV real comp_prev(int k) const { return k == 0 ? cprev : cprev_more[k > 0 ? k - 1 : 0]; }

    // the interior samples Θ = 1/7 .. 6/7: first sample at which a component's sign differs from its sign at the step's start;
    // returns that sample's index (0: none) and the components that changed there
    GR_DEV int sample_event_composite(const Params& p, real hh, int32_t& mask) const
    {
        real Ct[4], Cr[4];
        dense_coeffs(2, hh, Ct);
        dense_coeffs(1, hh, Cr);
        const int K = p.cfg.comp_n;
        for (int jj = 1; jj <= 6; ++jj) {
            const real th = (real)jj / 7.0;
            real s, c;
            sincos_fast(dense_eval(x[2], hh, Ct, th), s, c);
            const real rr = dense_eval(x[1], hh, Cr, th);
            int32_t mk = 0;
#pragma unroll
            for (int k = 0; k < GR_COMP_MAX; ++k) {       // (constant indices: the per-component state stays in registers)
                if (k >= K) continue;
                const int ps = sgn(comp_prev(k));
                if (ps != 0 && (real)ps * comp_cond(p, k, rr, s, c) <= 0.0) mk |= 1 << k;   // (<=: findall_events! of the vector callback)
            }
            if (mk) { mask = mk; return jj; }
        }
        mask = 0;
        return 0;
    }

    // left-biased root of a condition on the dense output inside Θ in [0, hi] (the bracketing of resolve_event below, for a
    // condition given as a functor): sign(f(0)) = ps
    template <class F>
    GR_DEV real root_on_dense(F f, int ps, real flo, real hi) const
    {
        real lo = 0.0, fhi = f(hi);
        if (fhi == 0.0) return hi;
        int side = 0;
        for (int it = 0; it < 48; ++it) {
            const real w = hi - lo;
            if (w <= 1.0e-13) break;
            real mid = lo - flo * w / (fhi - flo);
            const bool plateau = (flo == 1.0) || (fhi == 1.0);
            if (plateau || !(mid > lo && mid < hi)) {
                mid = lo + 0.5 * w;
                if (!(mid > lo && mid < hi)) break;
            }
            const real fm = f(mid);
            if (sgn(fm) == ps) {
                lo = mid; flo = fm;
                if (side < 0) fhi *= 0.5;
                side = -1;
            } else {
                hi = mid; fhi = fm;
                if (side > 0) flo *= 0.5;
                side = 1;
            }
        }
        return lo;
    }

    // DiscreteCallbacks in CallbackSet order: domain_upper_hemisphere, then the chart
    // (the library keeps "a PoloidalShapeChart is active" in bit 1 and "count windings" in bit 2 of its private
    // copy of cfg.upper_hemisphere, so the common PolarChart case tests one already-resident scalar)
    static GR_DEV bool discrete_cb(const Params& p, real r, real th, real c, int32_t& st, int32_t& fl)
    {
        bool term = false;
        const int32_t cb = p.cfg.upper_hemisphere;
        real rmin = p.cfg.r_inner;
        if (cb) {
            // the optional callbacks sit behind ONE scalar test, so the plain PolarChart trace pays nothing for them
            if (cb & 4) {
                // winding_callback (photon-rings.jl:1-15): the count lives in the upper half of the ray's flag word,
                // the plane in the cold block
                const real wp = (real)cold_of(p).winding_plane;
                if ((fl & 0x10000) ? th < wp : th > wp) {
                    if ((fl & 0xFFFF0000) != 0xFFFF0000) fl += 0x10000;
                }
            }
            if ((cb & 1) && r * c < p.cfg.hemi_delta) { st = GR_STATUS_OUT_OF_DOMAIN; term = true; }
            if (cb & 2) {
                // PoloidalShapeChart (charts.jl:26-48): r_min(θ) by linear interpolation of the table
                const int64_t n = p.cfg.chart_table_n;
                const real idth = (real)((double)(n - 1) / (p.cfg.chart_theta1 - p.cfg.chart_theta0));
                const real f = (th - (real)p.cfg.chart_theta0) * idth;
                int64_t k = (int64_t)GR_FLOOR(f);
                k = k < 0 ? 0 : (k > n - 2 ? n - 2 : k);
                const real y0 = (real)p.chart_table[k], y1 = (real)p.chart_table[k + 1];
                rmin = GR_FMA(f - (real)k, y1 - y0, y0);
            }
        }
        if (r <= rmin || r > p.cfg.r_outer) {
            st = (r <= rmin) ? GR_STATUS_WITHIN_INNER_BOUNDARY : GR_STATUS_OUT_OF_DOMAIN;
            term = true;
        }
        return term;
    }

    // ray k of a separable set -> (radius index i, angle index j).  Tiled order (lineprofiles._tile_order): whole 8 x 8
    // tiles first -- tiles down a column strip, lanes column-major inside a tile --, then the rays no whole tile covers in
    // column-major order; untiled: k = i + nr j.
    static GR_DEV void sep_index(const Cold& p, int64_t k, int64_t& i, int64_t& j)
    {
        const int64_t R = p.sep_core_rows, Cc = p.sep_core_cols, nr = p.sep_nr;
        const int64_t core = R * Cc;
        if (k < core) {
            const int64_t rr = k & 7, cc = (k >> 3) & 7, t = k >> 6;
            const int64_t tiles_down = R >> 3;
            const int64_t tcol = t / tiles_down, trow = t - tcol * tiles_down;
            i = (trow << 3) + rr;
            j = (tcol << 3) + cc;
            return;
        }
        k -= core;
        const int64_t tail = nr - R;                       // rows below the tiled block, in each of its Cc columns
        if (k < Cc * tail) {
            j = k / tail;
            i = R + (k - j * tail);
        } else {
            k -= Cc * tail;
            const int64_t jj = k / nr;
            j = Cc + jj;
            i = k - jj * nr;
        }
    }

    // local ray of a launch -> position in the order of the separable set (gr_rayset.sep_first / sep_block / sep_stride)
    static GR_DEV int64_t sep_global(const Cold& p, int64_t jl)
    {
        if (p.sep_block <= 0) return p.sep_first + jl;
        const int64_t b = jl / p.sep_block;
        return p.sep_first + b * p.sep_stride + (jl - b * p.sep_block);
    }

    // impact parameters of ray jl of an impact-parameter set (src_mode 2)
    static GR_DEV void impact_parameters_of(const Cold& p, int64_t jl, double& al, double& be)
    {
        if (p.sep_r) {
            int64_t i, j;
            sep_index(p, sep_global(p, jl), i, j);
            const double r = p.sep_r[i];
            al = r * p.sep_cos[j];
            be = r * p.sep_sin[j];
        } else {
            al = p.alpha[jl];
            be = p.beta[jl];
        }
    }

    // initial position / unconstrained velocity of local ray jl
    static GR_DEV void initial_conditions(const Params& pp, int64_t jl, real x0[4], real v0[4])
    {
        const Cold& p = cold_of(pp);
        if (p.src_mode == 0) {
            // _render_velocity_function, rendering.jl:140-163 ; local_momentum, utility.jl:13-20
            const int64_t i = range_map(p, jl);
            const int64_t H = p.plane.height;
            const int64_t xi = idx_div(p, i, H), yi = i - xi * H;
            const real alpha = range_at(p.plane.alpha0, p.plane.alpha1, p.plane.width, xi) + p.plane.offset;
            const real beta = range_at(p.plane.beta0, p.plane.beta1, H, yi) + p.plane.offset;
            const real ro = p.plane.x_obs[1];
            const real iro = rcp_full(ro);
            const real b = beta * iro, a = alpha * iro;
            const real pr = -rcp_full(sqrt_fast(1.0 + a * a + b * b));
            const real pb[4] = { 1.0, pr, b * pr, a * pr };
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x0[q] = p.plane.x_obs[q];
                v0[q] = p.plane.Mx[q * 4 + 0] * pb[0] + p.plane.Mx[q * 4 + 1] * pb[1] + p.plane.Mx[q * 4 + 2] * pb[2]
                        + p.plane.Mx[q * 4 + 3] * pb[3];
            }
        } else if (p.src_mode == 2) {
            // promote_velfunc: map_impact_parameters(m, x, αs[i], βs[i]) -- no pixel offset
            const real ro = p.plane.x_obs[1];
            const real iro = rcp_full(ro);
            double al_, be_;
            impact_parameters_of(p, jl, al_, be_);
#ifdef GR_REAL_IS_TAN2
            // the tangent directions of this build: ∂/∂α and ∂/∂β of everything downstream (GR_TAN_W = 1: this lane's one)
            const real al = gr_t_seed(al_, 0, tan_dir()), be = gr_t_seed(be_, 1, tan_dir());
#else
            const real al = (real)al_, be = (real)be_;
#endif
            const real b = be * iro, a = al * iro;
            const real pr = -rcp_full(sqrt_fast(1.0 + a * a + b * b));
            const real pb[4] = { 1.0, pr, b * pr, a * pr };
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x0[q] = p.plane.x_obs[q];
                v0[q] = p.plane.Mx[q * 4 + 0] * pb[0] + p.plane.Mx[q * 4 + 1] * pb[1] + p.plane.Mx[q * 4 + 2] * pb[2]
                        + p.plane.Mx[q * 4 + 3] * pb[3];
            }
        } else {
            const double* xs = p.x + jl * p.x_stride;
            const double* vs = p.v + jl * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) { x0[q] = xs[q]; v0[q] = vs[q]; }
        }
    }

    // The slot of this ray's tile in Cold::tile_cost if the ray is its tile's representative (the first row of the first
    // column), else -1.
    GR_DEV int64_t tile_cost_index(const Cold& cd) const
    {
        const int tr = cd.swizzle, tc = 6 - tr;
        const int64_t H = cd.plane.height;
        const int64_t col = idx_div(cd, j, H), row = j - col * H;
        if ((col & ((1 << tc) - 1)) == 0 && (row & ((1 << tr) - 1)) == 0) return (col >> tc) * (H >> tr) + (row >> tr);
        return -1;
    }

    // constrain_all (constraints.jl:14-15): v^t from the null/mass-shell condition
    static GR_DEV void constrained_u0(const Metric& m, const Params& p, int64_t jl, real x0[4], real v0[4])
    {
        initial_conditions(p, jl, x0, v0);
        real s, c, g[5];
        sincos_fast(x0[2], s, c);
        metric_comps(m, x0[1], x0[2], s, c, g);
        v0[0] = constrain_time(g, v0[1], v0[2], v0[3], p.cfg.mu);
    }

    static GR_DEV void accel(const Metric& m, real r, real th, const real vv[4], real a[4], real& s, real& c)
    {
        geodesic_rhs(m, r, th, vv[0], vv[1], vv[2], vv[3], a[0], a[1], a[2], a[3], s, c);
    }

    // reinit! + auto_dt_reset! (tracing.jl:234-243; App. A.4)
    GR_DEV void init(const Metric& m, const Params& p, int64_t jl)
    {
        j = jl;
        status = GR_STATUS_NO_STATUS;
        flags = 0; nacc = 0; nrej = 0; ev_top = 0;
        rotk.load();
        constrained_u0(m, p, jl, x, v);
        t = p.cfg.lambda0;
        h = 0.0;
        lq_old = (float)LOG2_QOLDINIT;
        real s, c;
        accel(m, x[1], x[2], v, A[0], s, c);
        sth = s; cth = c;
        if (DISC == GR_DISC_DATUM) {
            // datumplane(d, rₑ) (datum-plane.jl:14-17): one plane per ray when the set carries heights
            const Cold& cd = cold_of(p);
            hdat = (cd.src_mode == 2 && cd.height) ? (real)cd.height[jl] : (real)p.cfg.disc_params[0];
        }
        cprev = kContinuous ? disc_cond4(p, x[1], s, c, x[3]) : 1.0;
        ev_mask = 0;
#if GR_HAS_MESH
        if constexpr (DISC == GR_DISC_MESH) {
            to_cartesian3(x[1], s, c, x[3], qprev);
#pragma unroll
            for (int i = 0; i < 3; ++i) qnew[i] = qprev[i];
            mesh_need = 0; mesh_cb_term = 0; mesh_coop = 0;
        }
#endif
        if constexpr (DISC == GR_DISC_COMPOSITE) {
#pragma unroll
            for (int k = 0; k < GR_COMP_MAX; ++k)
                if (k < p.cfg.comp_n) comp_prev(k) = comp_cond(p, k, x[1], s, c);
        }

        const real abstol = p.cfg.abstol, reltol = p.cfg.reltol;
        const real dtmax = (real)p.dtmax;
        real iskx[4], iskv[4], d0s = 0.0, d1s = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            iskx[i] = rcp_full(abstol + GR_FABS(x[i]) * reltol);
            iskv[i] = rcp_full(abstol + GR_FABS(v[i]) * reltol);
            const real a0 = x[i] * iskx[i], b0 = v[i] * iskv[i];
            const real a1 = v[i] * iskx[i], b1 = A[0][i] * iskv[i];
            d0s = GR_FMA(a0, a0, d0s);
            d0s = GR_FMA(b0, b0, d0s);
            d1s = GR_FMA(a1, a1, d1s);
            d1s = GR_FMA(b1, b1, d1s);
        }
        const real d0 = sqrt_fast(d0s * 0.125), d1 = sqrt_fast(d1s * 0.125);
        real dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 * rcp_full(d1));
        dt0 = GR_FMIN(dt0, dtmax);
        if (dt0 < 10.0 * GR_EPS) {
            dt = 1e-6;
        } else {
            real v1[4], a1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v1[i] = GR_FMA(dt0, A[0][i], v[i]);
            accel(m, GR_FMA(dt0, v[1], x[1]), GR_FMA(dt0, v[2], x[2]), v1, a1, s, c);
            real d2s = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const real ex = (v1[i] - v[i]) * iskx[i];
                const real ev = (a1[i] - A[0][i]) * iskv[i];
                d2s = GR_FMA(ex, ex, d2s);
                d2s = GR_FMA(ev, ev, d2s);
            }
            const real d2 = sqrt_fast(d2s * 0.125) * rcp_full(dt0);
            const real dm = GR_FMAX(d1, d2);
            // 10^(-(2 + log10 dm)/5) = 10^-0.4 dm^-0.2 ; single-precision hardware log2/exp2 is ample
            // for a starting step size
#ifdef GR_CONTROLLER_F64
            const real dt1 = (dm <= 1e-15) ? GR_FMAX(1e-6, dt0 * 1e-3)
                                             : 0.39810717055349726 * (real)::exp2(-0.2 * ::log2((double)dm));
#else
            const real dt1 = (dm <= 1e-15) ? GR_FMAX(1e-6, dt0 * 1e-3)
                                             : 0.39810717055349726 * (real)fast_exp2f(-0.2f * fast_log2f((float)dm));
#endif
            dt = (hreal)GR_FMIN(GR_FMIN(100.0 * dt0, dt1), dtmax);
        }
    }

#ifdef GR_HOST_HARNESS
#define GR_DBG_BIT(b) dbg_bits |= (b)
#define GR_DBG_DMAX(d) dbg_dmax = GR_FMAX(dbg_dmax, (d))
#else
#define GR_DBG_BIT(b)
#define GR_DBG_DMAX(d)
#endif
    // One attempted Tsit5 step.  Returns true when the ray has finished (terminated by a
    // callback, reached λ1, or hit an anomaly).
    GR_DEV bool step(const Metric& m, const Params& p) { return step(m, p, NoColdStore{}); }

    template <class Cold_>
    GR_DEV bool step(const Metric& m, const Params& p, const Cold_& cs)
    {
        const real tend = p.cfg.lambda1;
        const hreal dtmax = (hreal)p.dtmax;
        GR_DBG_BIT(0);
#ifdef GR_HOST_HARNESS
        dbg_bits = 0;
        dbg_dmax = 0.0;
#endif
        if (nacc + nrej >= p.maxiters32) { flags |= GR_FLAG_MAXITERS; return true; }
        hreal hh = GR_FMIN(dt, dtmax);
        // one comparison on the common path: a NaN step size fails it as well and is told apart inside
        if (!(hh >= (hreal)(4.0 * GR_EPS * GR_FMAX(GR_FABS(t), 1.0)))) {
            flags |= (hh == hh) ? GR_FLAG_DTMIN : GR_FLAG_NAN;
            return true;
        }
        hh = GR_FMIN(hh, (hreal)(tend - t));
        h = hh;
        const hreal h2 = hh * hh;
        const bool resync = (nacc & 63) == 63;      // full sin/cos at the new state (decided while nacc is in a register)
        if constexpr (Cold_::kHead) {
            // park what the hot region does not read (see LdsColdStore)
            cs.template st<real>(0, t);
            cs.template st<real>(1, dt);
            cs.template st<real>(2, x[0]);
            cs.template st<real>(3, x[3]);
            cs.template st<real>(4, cprev);
            cs.st2(5, status, flags);
            cs.st2(6, nacc, nrej);
            cs.st2(7, ev_top, __builtin_bit_cast(int32_t, (float)lq_old));
            cs.template st<int64_t>(8, j);
            Cold_::fence();
        }

        real s, c;
        // stage accelerations parked in LDS (ParkA): store A[S] behind its right-hand side, reload A[1..UPTO] where sums read them
#define GR_PARK(S)                                                                             \
    if constexpr (Cold_::kParkA >= (S)) {                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) cs.stA((S), i_, A[(S)][i_]);          \
        Cold_::park_fence();                                                                   \
    }
#define GR_UNPARK(UPTO)                                                                        \
    if constexpr (Cold_::kParkA > 0) {                                                         \
        Cold_::park_fence();                                                                   \
        _Pragma("unroll") for (int q_ = 1; q_ <= ((UPTO) < Cold_::kParkA ? (UPTO) : Cold_::kParkA); ++q_)   \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) A[q_][i_] = cs.ldA(q_, i_);       \
    }
#ifdef GR_REAL_IS_TAN2
#define GR_PIN4(a) { gr_t_pin((a)[0]); gr_t_pin((a)[1]); gr_t_pin((a)[2]); gr_t_pin((a)[3]); }     // see gr_t_pin
#else
#define GR_PIN4(a)
#endif
        // stages 2..6: arguments need r, θ and the four velocities only (the RHS does not
        // depend on t or ϕ)
#if GR_PK_F32
#define GR_STAGE_VSUM(S)                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; i += 2)                                              \
        {                                                                                             \
            gr_f2 acc = GR_PK2(A[0], i);                                                              \
            _Pragma("unroll") for (int q = 1; q < S; ++q) acc = GR_PKFMA(TsD::X.AR[S][q], GR_PK2(A[q], i), acc); \
            const gr_f2 w = GR_PKFMA(ha, acc, GR_PK2(v, i));                                          \
            vs[i] = w.x; vs[i + 1] = w.y;                                                             \
        }
#else
#define GR_STAGE_VSUM(S)                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
        {                                                                                             \
            real acc = A[0][i];                                                                     \
            _Pragma("unroll") for (int q = 1; q < S; ++q) acc = GR_FMA(TsD::X.AR[S][q], A[q][i], acc); \
            vs[i] = GR_FMA(ha, acc, v[i]);                                                     \
        }
#endif
#define GR_STAGE(S)                                                                                   \
    {                                                                                                 \
        GR_UNPARK((S) - 1)                                                                            \
        real vs[4];                                                                                 \
        const hreal ha = Ts::A[S][0] * hh;                                                          \
        GR_STAGE_VSUM(S)                                                                              \
        real rs = GR_FMA(TsD::X.C[S] * hh, v[1], x[1]);                                           \
        real ts = GR_FMA(TsD::X.C[S] * hh, v[2], x[2]);                                           \
        if (S > 1) {                                                                                  \
            real ar = A[0][1], at = A[0][2];                                                        \
            _Pragma("unroll") for (int q = 1; q < S - 1; ++q)                                         \
            {                                                                                         \
                ar = GR_FMA(TsD::X.AXR[S][q], A[q][1], ar);                                         \
                at = GR_FMA(TsD::X.AXR[S][q], A[q][2], at);                                         \
            }                                                                                         \
            const hreal h2a = TsD::X.AX[S][0] * h2;                                                 \
            rs = GR_FMA(h2a, ar, rs);                                                          \
            ts = GR_FMA(h2a, at, ts);                                                          \
        }                                                                                             \
        sincos_rot_stage(rotk, x[2], sth, cth, ts, s, c);                                             \
        GR_DBG_BIT((GR_FABS(ts - x[2]) <= SINCOS_ROT_MAX) ? 0 : (1 << S));                                    \
        GR_DBG_DMAX(GR_FABS(ts - x[2]));                                                              \
        if constexpr (ByThetaOf<Metric>::value)                                                       \
            geodesic_rhs_th(m, cs, rs, ts, s, c, vs[0], vs[1], vs[2], vs[3], A[S][0], A[S][1], A[S][2], A[S][3]); \
        else                                                                                          \
        geodesic_rhs_sc(m, rs, s, c, vs[0], vs[1], vs[2], vs[3], A[S][0], A[S][1], A[S][2], A[S][3]); \
        GR_PIN4(A[S]);                                                                                \
        GR_PARK(S)                                                                                    \
    }
        GR_STAGE(1)
        GR_STAGE(2)
        GR_STAGE(3)
        GR_STAGE(4)
        GR_STAGE(5)
#undef GR_STAGE
#undef GR_STAGE_VSUM
        // stage 7 argument = the new state.  Its right-hand side reads r, θ and the velocities; t and ϕ of the new state are
        // formed behind it (their old values are parked when the cold store is on)
        real xn[4], vn[4];
        GR_UNPARK(5)
        const hreal ha6 = Ts::A[6][0] * hh, hc6 = TsD::X.C[6] * hh, h2a6 = TsD::X.AX[6][0] * h2;
#if GR_PK_F32
        // (all four positions here, in pairs: t and ϕ of the new state then live across the last right-hand side -- two registers)
        static_assert(Cold_::kParkA == 0 && !Cold_::kHead, "the packed sums keep the whole state in registers");
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            gr_f2 acc = GR_PK2(A[0], i), ax = GR_PK2(A[0], i);
#pragma unroll
            for (int q = 1; q < 6; ++q) acc = GR_PKFMA(TsD::X.AR[6][q], GR_PK2(A[q], i), acc);
#pragma unroll
            for (int q = 1; q < 5; ++q) ax = GR_PKFMA(TsD::X.AXR[6][q], GR_PK2(A[q], i), ax);
            const gr_f2 wv = GR_PKFMA(ha6, acc, GR_PK2(v, i));
            const gr_f2 wx = GR_PKFMA(h2a6, ax, GR_PKFMA(hc6, GR_PK2(v, i), GR_PK2(x, i)));
            vn[i] = wv.x; vn[i + 1] = wv.y;
            xn[i] = wx.x; xn[i + 1] = wx.y;
        }
#define GR_NEW_POSITION(i)
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            real acc = A[0][i];
#pragma unroll
            for (int q = 1; q < 6; ++q) acc = GR_FMA(TsD::X.AR[6][q], A[q][i], acc);
            vn[i] = GR_FMA(ha6, acc, v[i]);
        }
#define GR_NEW_POSITION(i)                                                                        \
    {                                                                                             \
        real ax = A[0][i];                                                                        \
        _Pragma("unroll") for (int q = 1; q < 5; ++q) ax = GR_FMA(TsD::X.AXR[6][q], A[q][i], ax); \
        xn[i] = GR_FMA(h2a6, ax, GR_FMA(hc6, v[i], x[i]));                                        \
    }
        GR_NEW_POSITION(1)
        GR_NEW_POSITION(2)
        if constexpr (Cold_::kParkA > 0) {      // t and ϕ of the new state while the parked accelerations are here
            GR_NEW_POSITION(0)
            GR_NEW_POSITION(3)
        }
#endif
        // sin/cos at the new state by rotating the step's base as well (the RHS at the new state is stage 7 and the
        // base of the next step).  Rotation errors random-walk by ~1 ulp per step, so the base is re-synchronised
        // with a full evaluation every 64 accepted steps (and whenever the rotation falls back to it anyway).
        real sn, cn;
#ifndef GR_NO_ROT_FINAL
        GR_DBG_DMAX(GR_FABS(xn[2] - x[2]));
        if (resync) sincos_fast(xn[2], sn, cn);
        else sincos_rot(rotk, x[2], sth, cth, xn[2], sn, cn);
        if constexpr (ByThetaOf<Metric>::value) geodesic_rhs_th(m, cs, xn[1], xn[2], sn, cn, vn[0], vn[1], vn[2], vn[3], A[6][0], A[6][1], A[6][2], A[6][3]);
        else geodesic_rhs_sc(m, xn[1], sn, cn, vn[0], vn[1], vn[2], vn[3], A[6][0], A[6][1], A[6][2], A[6][3]);
        GR_PIN4(A[6]);
#else
        accel(m, xn[1], xn[2], vn, A[6], sn, cn);
#endif
        if constexpr (Cold_::kHead) {
            Cold_::fence();
            t = cs.template ld<real>(0);
            dt = cs.template ld<real>(1);
            x[0] = cs.template ld<real>(2);
            x[3] = cs.template ld<real>(3);
            cprev = cs.template ld<real>(4);
            cs.ld2(5, status, flags);
            cs.ld2(6, nacc, nrej);
            int32_t lqb;
            cs.ld2(7, ev_top, lqb);
            lq_old = __builtin_bit_cast(float, lqb);
            j = cs.template ld<int64_t>(8);
        }
        if constexpr (Cold_::kParkA == 0) {
            GR_NEW_POSITION(0)
            GR_NEW_POSITION(3)
        }
#undef GR_NEW_POSITION
        GR_UNPARK(5)          // the error estimate reads every stage

        // error estimate, squared RMS norm over all eight components: ũ_v = h Σ b̃_q A_q, ũ_x = h (Σb̃ · v + h Σ b̄_i A_i);
        // the common factor h² and the 1/8 of the mean are applied once to the sum.
        // The Σb̃ · v term looks negligible (Σb̃ = 1.4e-17: the rounding residue of the published coefficients) and is
        // NOT: in the first steps of a ray the true error estimate is ~1e-10 of the tolerance scale, t starts at 0 (scale =
        // abstol), and h Σb̃ v^t / abstol ≈ 1e-8 is then the largest contribution -- it decides whether the controller's
        // growth factor hits its clamp (10x) or not (5x).  The reference's first-order form carries the same kind of
        // residue (rounding of Σ b̃_j v_j ≈ 4e-16 |v|) and grows 4x there.  Dropping the term kept every result within
        // tolerance but moved the device's early step sequence away from the oracle's: median redshift difference on C2
        // 4.3e-11 with it, 1.0e-10 without (tests/test_gpu_baseline_configs.py).
        const real abstol = p.cfg.abstol, reltol = p.cfg.reltol;
        real e2v = 0.0, e2x = 0.0;        // Σ (ũ_v / b̃_0 / scale)², Σ (ũ_x / scale)² (before the common h² / 8)
#if GR_NORM_F32_ON
        float e2vf = 0.f, e2xf = 0.f;
#endif
#ifdef GR_REAL_IS_TAN2
        double e2n = 0.0;
#endif
        const hreal hbx = TsD::X.BTX[0] * hh;
#if GR_PK_F32
        real ev4[4], ex4[4];
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            gr_f2 ev2 = GR_PK2(A[0], i), ex2 = GR_PK2(A[0], i);
#pragma unroll
            for (int q = 1; q < 7; ++q) ev2 = GR_PKFMA(TsD::X.BTR[q], GR_PK2(A[q], i), ev2);
#pragma unroll
            for (int q = 1; q < 6; ++q) ex2 = GR_PKFMA(TsD::X.BTXR[q], GR_PK2(A[q], i), ex2);
            ex2 = GR_PKFMA(hbx, ex2, (gr_f2)(TsD::X.SBT) * GR_PK2(v, i));
            ev4[i] = ev2.x; ev4[i + 1] = ev2.y;
            ex4[i] = ex2.x; ex4[i + 1] = ex2.y;
        }
#endif
#if GR_PK_F32
        // ... and the scaled residuals of the pairs: scale = reltol max(|u0|, |u1|) + abstol, residual / scale, squares summed -- packed
        // FMA / MUL around two scalar reciprocals
        {
            gr_f2 s2v = { 0.f, 0.f }, s2x = { 0.f, 0.f };
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const gr_f2 mv = { absmax(v[i], vn[i]), absmax(v[i + 1], vn[i + 1]) }, mx = { absmax(x[i], xn[i]), absmax(x[i + 1], xn[i + 1]) };
                const gr_f2 kv = __builtin_elementwise_fma(mv, (gr_f2)(reltol), (gr_f2)(abstol));
                const gr_f2 kx = __builtin_elementwise_fma(mx, (gr_f2)(reltol), (gr_f2)(abstol));
                const gr_f2 av = GR_PK2(ev4, i) * gr_f2{ rcp_raw(kv.x), rcp_raw(kv.y) };
                const gr_f2 ax = GR_PK2(ex4, i) * gr_f2{ rcp_raw(kx.x), rcp_raw(kx.y) };
                s2v = __builtin_elementwise_fma(av, av, s2v);
                s2x = __builtin_elementwise_fma(ax, ax, s2x);
            }
            e2v = s2v.x + s2v.y;
            e2x = s2x.x + s2x.y;
        }
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if GR_PK_F32
            const real ev = ev4[i], ex = ex4[i];
#else
            real ev = A[0][i];               // ũ_v / (h b̃_0)
#pragma unroll
            for (int q = 1; q < 7; ++q) ev = GR_FMA(TsD::X.BTR[q], A[q][i], ev);
            real ex = A[0][i];
#pragma unroll
            for (int q = 1; q < 6; ++q) ex = GR_FMA(TsD::X.BTXR[q], A[q][i], ex);
            ex = GR_FMA(hbx, ex, TsD::X.SBT * v[i]);
#endif
#ifdef GR_REAL_IS_TAN2
            if (p.tangent_norm) {
                // The reference integrates Dual state through OrdinaryDiffEq (precision-solvers.jl:73-131,401-451) and
                // DiffEqBase's ForwardDiff extension (third party; as published) lets the partials into the controller:
                //   internalnorm(u::Dual) = sqrt(value² + Σ partials²),   internalnorm(array) = sqrt(Σ sse / (N (1 + P))),
                // so the residual of component i is ũ_i / (abstol + reltol max(|u0_i|_D, |u1_i|_D)) as a Dual and EEst² is
                // the mean over all 8 x 3 entries.  Steps shorten where the tangents are the stiffer part.
                // max(‖u0‖_D, ‖u1‖_D) = sqrt(max of the squares): one square root per scale
                const double qv0 = __builtin_fma(v[i].v, v[i].v, gr_t_tan_sq(v[i]));
                const double qv1 = __builtin_fma(vn[i].v, vn[i].v, gr_t_tan_sq(vn[i]));
                const double qx0 = __builtin_fma(x[i].v, x[i].v, gr_t_tan_sq(x[i]));
                const double qx1 = __builtin_fma(xn[i].v, xn[i].v, gr_t_tan_sq(xn[i]));
                const double sv = abstol.v + reltol.v * gr_d_sqrt(qv0 > qv1 ? qv0 : qv1);
                const double sx = abstol.v + reltol.v * gr_d_sqrt(qx0 > qx1 ? qx0 : qx1);
                const double iv = gr_d_rcp(sv) * (double)Ts::BT[0], ix = gr_d_rcp(sx);
                e2n += __builtin_fma(ev.v, ev.v, gr_t_tan_sq(ev)) * (iv * iv) + __builtin_fma(ex.v, ex.v, gr_t_tan_sq(ex)) * (ix * ix);
                continue;
            }
#endif
#if GR_NORM_F32_ON
            // The scaled residuals in SINGLE precision (round 4; -DGR_NORM_F32=0 restores the FP64 form): one v_cvt_f32_f64 per
            // operand, then v_fma_f32 / v_rcp_f32 / v_mul_f32 instead of 2 FMA + 2 quarter-rate v_rcp_f64 + 2 MUL + 2 FMA in FP64
            // per component -- 809 -> 776 FP64 instructions per step, 18.20 -> 17.93 ms interleaved on one box
            // (profiles/r4h_ab_normf32.txt).  The quotient feeds a controller that works in single precision anyway (and whose
            // FP64 form used the bare 4.6e-8 reciprocal seed): on C2 every pixel keeps its class (814 flips against the oracle
            // with either form, 0 between them), steps per ray agree to 1e-8, the soak's classes are unchanged.
            {
                const float smv = (float)absmax(v[i], vn[i]), smx = (float)absmax(x[i], xn[i]);
                const float avf = (float)ev * GR_RCPF(__builtin_fmaf(smv, (float)reltol, (float)abstol));
                const float axf = (float)ex * GR_RCPF(__builtin_fmaf(smx, (float)reltol, (float)abstol));
                e2vf = __builtin_fmaf(avf, avf, e2vf);
                e2xf = __builtin_fmaf(axf, axf, e2xf);
            }
#else
            const real skv = GR_FMA(absmax(v[i], vn[i]), reltol, abstol);
            const real skx = GR_FMA(absmax(x[i], xn[i]), reltol, abstol);
            // the bare v_rcp_f64 seed is good to 4.6e-8 (measured, scripts/rcp_accuracy.hip): ample for
            // a quantity that only feeds the step-size controller and the accept test
            const real av = ev * rcp_raw(skv);
            const real ax = ex * rcp_raw(skx);
            e2v = GR_FMA(av, av, e2v);
            e2x = GR_FMA(ax, ax, e2x);
#endif
        }
#endif
#if GR_NORM_F32_ON
        real e2 = (real)__builtin_fmaf((float)(Ts::BT[0] * Ts::BT[0]), e2vf, e2xf);
#else
        real e2 = GR_FMA((real)(Ts::BT[0] * Ts::BT[0]), e2v, e2x);
#endif
#ifdef GR_REAL_IS_TAN2
        if (p.tangent_norm) e2 = real(e2n * (1.0 / 3.0));
#endif
        e2 *= 0.125 * h2;   // EEst² ; accept iff EEst <= 1
#ifdef GR_HOST_HARNESS
        dbg_e2 = e2;
#endif

        // PI controller in log2 space: q = EEst^β1 / qold^β2 / γ.  The step factor is formed with
        // single-precision hardware log2/exp2 (relative error ~1e-6 in dt, far below anything the
        // 1e-9 tolerance can see; the reference's DiffEqBase.fastpow is itself Float32-based).
        // GR_CONTROLLER_F64 builds the same controller on double-precision log2/exp2 (A/B of the disc-rim
        // classification against the oracle, scripts/controller_ab.py).
#ifdef GR_CONTROLLER_F64
        typedef double ctl_t;
#define GR_CTL_LOG2(x) ::log2((double)(x))
#define GR_CTL_EXP2(x) ::exp2((double)(x))
#define GR_CTL_MIN(a, b) ::fmin((double)(a), (double)(b))
#define GR_CTL_MAX(a, b) ::fmax((double)(a), (double)(b))
#else
        typedef float ctl_t;
#define GR_CTL_LOG2(x) fast_log2f((float)(x))
#define GR_CTL_EXP2(x) fast_exp2f((float)(x))
#define GR_CTL_MIN(a, b) ::fminf((float)(a), (float)(b))
#define GR_CTL_MAX(a, b) ::fmaxf((float)(a), (float)(b))
#endif
        const ctl_t lE = (ctl_t)0.5 * GR_CTL_LOG2(e2);       // log2(EEst); -inf when e2 underflows
        if (e2 <= 1.0) {
            // dt_next = dt / q with q = clamp(EEst^β1 / qold^β2 / γ, 1/qmax, 1/qmin): formed directly as the growth factor
            // 1/q = clamp(γ 2^(β2 log2 qold - β1 log2 EEst), qmin, qmax), so the accepted step needs no reciprocal
            // (round 2 clamped q and then divided: one v_rcp_f64, four Newton FMAs and a conversion more per step)
            ctl_t gf = GR_CTL_EXP2((ctl_t)PI_BETA2 * (ctl_t)lq_old - (ctl_t)PI_BETA1 * lE) * (ctl_t)PI_GAMMA;
            gf = GR_CTL_MIN((ctl_t)PI_QMAX, GR_CTL_MAX((ctl_t)PI_QMIN, gf));   // e2 == 0 -> lE = -inf -> qmax
            nacc++;
            lq_old = GR_CTL_MAX(lE, (ctl_t)LOG2_QOLDINIT);
            hreal dtnew = hh * (hreal)gf;
            real tnew = t + hh;
            if (GR_FABS(tnew - tend) < 100.0 * GR_EPS * GR_FMAX(GR_FABS(tnew), GR_FABS(tend))) tnew = tend;

            if constexpr (DISC == GR_DISC_COMPOSITE) {
                const int K = p.cfg.comp_n;
                real cnx[GR_COMP_MAX];
                int32_t mask_end = 0;
#pragma unroll
                for (int k = 0; k < GR_COMP_MAX; ++k) {
                    if (k >= K) continue;
                    cnx[k] = comp_cond(p, k, xn[1], sn, cn);
                    const real cp = comp_prev(k);
                    const bool pos = cp > 0.0, neg = cp < 0.0;
                    if ((pos && !(cnx[k] > 0.0)) || (neg && !(cnx[k] < 0.0))) mask_end |= 1 << k;
                }
                int top = 0;
                int32_t mask = 0;
                if (__builtin_popcount((unsigned)mask_end) != K) {
                    GR_UNPARK(5)
                    top = sample_event_composite(p, hh, mask);
                }
                if (!top && mask_end) { top = 7; mask = mask_end; }
                if (top) {
                    flags |= RAY_EVENT;
                    ev_top = top;
                    ev_mask = mask;
                    status = GR_STATUS_INTERSECTED_WITH_GEOMETRY;
                    return true;
                }
#pragma unroll
                for (int k = 0; k < GR_COMP_MAX; ++k)
                    if (k < K) comp_prev(k) = cnx[k];
            } else if (kContinuous) {
                const real cnext = disc_cond4(p, xn[1], sn, cn, xn[3]);
                // prev = sign(c(u_prev)), event at the step's end iff prev != 0 and prev * sign(c(u_new)) <= 0 (a NaN
                // condition has sign 0 and counts as a crossing): four comparisons, no integer sign arithmetic
                const bool pos = cprev > 0.0, neg = cprev < 0.0;
                int top = 0;
                if (pos || neg) {
                    const bool crossed = pos ? !(cnext > 0.0) : !(cnext < 0.0);
                    if (crossed) top = 7;
                    else if (!sample_reach_excludes(p, pos ? 1 : -1, hh)) {
                        // the interior samples are evaluated on ~2 % of the wave-steps and need a dozen registers of
                        // their own: what they do not read waits in the cold store meanwhile (no scratch spill)
                        if constexpr (Cold_::kOn) {
                            cs.template st<real>(0, tnew);
                            cs.template st<real>(1, dtnew);
                            cs.template st<real>(2, x[0]);
                            cs.template st<real>(3, x[3]);
                            cs.template st<real>(4, xn[0]);
                            cs.template st<real>(5, xn[3]);
                            cs.template st<real>(6, vn[0]);
                            cs.template st<real>(7, vn[3]);
                            cs.template st<real>(8, t);
                            Cold_::fence();
                        }
                        GR_UNPARK(5)          // the dense output reads the member array
                        top = sample_event(p, pos ? 1 : -1, hh);
                        if constexpr (Cold_::kOn) {
                            Cold_::fence();
                            tnew = cs.template ld<real>(0);
                            dtnew = cs.template ld<real>(1);
                            x[0] = cs.template ld<real>(2);
                            x[3] = cs.template ld<real>(3);
                            xn[0] = cs.template ld<real>(4);
                            xn[3] = cs.template ld<real>(5);
                            vn[0] = cs.template ld<real>(6);
                            vn[3] = cs.template ld<real>(7);
                            t = cs.template ld<real>(8);
                        }
                    }
                }
                if (top) {
                    // leave (x, v, A, h) in place; finalize() root-finds on the dense output
                    flags |= RAY_EVENT;
                    ev_top = top;
                    status = GR_STATUS_INTERSECTED_WITH_GEOMETRY;
                    return true;
                }
                cprev = cnext;
            }
            // the discrete callbacks in CallbackSet order (callbacks.jl:19-38, bootstrap.jl:11-21): the geometry's, the user's,
            // the chart's -- each one whose condition holds applies its affect!, so a later one's status stands [3P]
            bool term = false;
#if GR_HAS_MESH
            if constexpr (DISC == GR_DISC_MESH) {
                to_cartesian3(xn[1], sn, cn, xn[3], qnew);
                mesh_need = mesh_inside(p, qnew) ? 1 : 0;
                if (!mesh_coop) {
                    if (mesh_need && mesh_lane_query(p, qprev, qnew)) { status = GR_STATUS_INTERSECTED_WITH_GEOMETRY; term = true; }
                    mesh_need = 0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) qprev[i] = qnew[i];
                }
            }
#endif
            const bool cb_term = discrete_cb(p, xn[1], xn[2], cn, status, flags);
            term |= cb_term;
            if constexpr (DISC == GR_DISC_MESH) mesh_cb_term = cb_term ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = xn[i]; v[i] = vn[i]; A[0][i] = A[6][i]; }
            sth = sn; cth = cn;
            t = tnew;
            dt = GR_FMIN(dtmax, dtnew);
            return term || !(t < tend);
        } else {
            // step_reject_controller!; a NaN state lands here too (e2 is NaN).  fminf(1/qmin, NaN) returns
            // 1/qmin, so the step size would stay finite and the ray would be rejected ~25 times down to
            // dt < dtmin: test for it here (rejected steps only, so the accepted path pays nothing)
            nrej++;
            if constexpr (DISC == GR_DISC_MESH) mesh_need = 0;
            if (!(e2 == e2)) {
#if GR_NORM_F32_ON
                // The single-precision norm also turns NaN when a trial state is astronomically far off (tolerance 1e-3 next to
                // the hole: v_new ~ 1e44 overflows the scale to inf, inf x 1/inf): that is a step to reject, not a NaN state --
                // the FP64 norm called it 1e160 and rejected it.  A state that really holds a NaN shows it in (x_new, v_new) or
                // in the right-hand side there; only then is the ray flagged.  (fminf below returns 1/qmin for the NaN factor.)
                bool state_nan = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) state_nan |= !(xn[i] == xn[i]) || !(vn[i] == vn[i]) || !(A[6][i] == A[6][i]);
                if (state_nan) { flags |= GR_FLAG_NAN; return true; }
#else
                flags |= GR_FLAG_NAN;
                return true;
#endif
            }
            const ctl_t q11 = GR_CTL_EXP2((ctl_t)PI_BETA1 * lE);
            dt = hh / (hreal)GR_CTL_MIN((ctl_t)(1.0 / PI_QMIN), q11 * (ctl_t)(1.0 / PI_GAMMA));
            return false;
        }
#undef GR_CTL_LOG2
#undef GR_CTL_EXP2
#undef GR_CTL_MIN
#undef GR_CTL_MAX
#undef GR_PARK
#undef GR_UNPARK
    }

#if GR_HAS_MESH && !defined(GR_HOST_HARNESS)
    // GR_DISC_MESH in the one-ray-per-lane kernel: ALL 64 lanes of the wave come here after every step (finished lanes too: they
    // are the helpers).  A few tests due: the wave takes them one after the other, 64 candidates at a time (mesh_wave_query);
    // many due at once: every lane walks its own candidates, in parallel (mesh_lane_query).  The outcome is the inline test's:
    // a hit ends the ray with IntersectedWithGeometry unless a later callback of the set ended it at the same step.
    GR_DEV bool mesh_phase(const Params& p, bool live, bool fin)
    {
        const bool need = live && mesh_need;
        unsigned long long due = __ballot(need);
        bool hit = false;
        if (due) {
            // q tests due, each over c candidates: lane by lane in parallel costs ~c dependent rounds, the wave on one test after
            // the other ~q (c / 64 + a few) rounds.  Measured on a ring slab of 960 / 3840 triangles at 1024², by wave up to
            // q = 0 / 2 / 8 / 16 / 32 / 64: 11.4 / 10.1 / 9.4 / 9.1 / 10.7 / 17.7 ms and 25.0 / 22.4 / 18.2 / 15.7 / 14.4 / 24.7 ms
            // (a rule that also counts the candidates first costs more than it finds: 9.9 and 16.7-18.0 ms)
            const bool by_wave = __popcll(due) <= GR_MESH_WAVE_MAX;
            if (!by_wave) {
                if (need) hit = mesh_lane_query(p, qprev, qnew);
            } else {
                const int lane = (int)(threadIdx.x & 63);
                do {
                    const int src = __ffsll((long long)due) - 1;
                    due &= due - 1;
                    real Q1[3], Q2[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) { Q1[i] = __shfl(qprev[i], src, 64); Q2[i] = __shfl(qnew[i], src, 64); }
                    const bool h = mesh_wave_query(p, Q1, Q2, lane);
                    if (lane == src) hit = h;
                } while (due);
            }
        }
        if (live) {
            mesh_need = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) qprev[i] = qnew[i];
            if (hit) {
                if (!mesh_cb_term) status = GR_STATUS_INTERSECTED_WITH_GEOMETRY;
                fin = true;
            }
        }
        return fin;
    }
#endif

    // dense-output polynomial coefficients of component `comp` (0..3 position, 4..7 velocity):
    // y(Θ) = y0 + h Σ_m C[m] Θ^(m+1)
    GR_DEV void dense_coeffs(int comp, real hh, real C[4]) const
    {
        if (comp >= 4) {
            const int q = comp - 4;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                real acc = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    if (Ts::R[i][mm] != 0.0) acc = GR_FMA(Ts::R[i][mm], A[i][q], acc);
                C[mm] = acc;
            }
        } else {
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                real acc = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (TsD::X.RX[i][mm] != 0.0) acc = GR_FMA(TsD::X.RX[i][mm], A[i][comp], acc);
                C[mm] = GR_FMA(hh, acc, TsD::X.SR[mm] * v[comp]);
            }
        }
    }
    static GR_DEV real dense_eval(real u0, real hh, const real C[4], real th)
    {
        const real poly = th * (C[0] + th * (C[1] + th * (C[2] + th * C[3])));
        return GR_FMA(hh, poly, u0);
    }

    // ContinuousCallback safety sampling at Θ = j/7, j = 1..6 (Θ = 1 was tested by the caller).
    // Returns the first j whose condition has the opposite sign of `ps`, or 0.
    // A sample can only change sign if it lies inside the |cosθ| < gtol wedge.  Cheap exits first:
    // (1) a bound on how far θ can move inside the step, (2) θ alone at the six samples; the full
    // condition is evaluated only for steps that come near the equatorial plane.
    // cheap exit of the event sampling, on every accepted step: true = no interior sample can change the condition's sign
    GR_DEV bool sample_reach_excludes(const Params& p, int ps, real hh) const
    {
        constexpr bool thin = (DISC == GR_DISC_THIN);
        if (thin && ps > 0) {
            // |θ(Θ_j) - θ_0| <= h (Θ_j |v^θ| + h Σ_i |RXΣ_i(Θ_j)| |A_i^θ|) <= h (|v^θ| + K h max_i |A_i^θ|)
            real amax = absmax(A[0][2], A[1][2]);
#pragma unroll
            for (int i = 2; i < 6; ++i) amax = absmax(amax, A[i][2]);
            const real reach = hh * (GR_FABS(v[2]) * DENSE_K1 + DENSE_K2 * hh * amax);
            // distance of θ from the equatorial plane (mod π) is asin|cosθ| >= |cosθ|, and cosθ of the step's base
            // is at hand: |cosθ| - reach > wedge rules every sample out
            if (GR_FABS(cth) - reach > (real)p.wedge) return true;
        }
        return false;
    }

    GR_DEV int sample_event(const Params& p, int ps, real hh) const
    {
        const real wedge = p.wedge;
        constexpr bool thin = (DISC == GR_DISC_THIN);
        GR_DBG_BIT(1 << 8);
        real Ct[4], Cr[4];
        dense_coeffs(2, hh, Ct);
        bool any = (ps < 0);
        if (thin) {
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                real d = dense_eval(x[2], hh, Ct, (real)(jj + 1) / 7.0) - 1.5707963267948966;
                d -= 3.141592653589793 * GR_RINT(d * 0.3183098861837907);
                any |= (GR_FABS(d) < wedge);
            }
            if (!any) return 0;
            GR_DBG_BIT(1 << 9);
            dense_coeffs(1, hh, Cr);
        } else if (DISC == GR_DISC_DATUM || DISC == GR_DISC_PRECESSING_THIN || DISC == GR_DISC_ELLIPTICAL) {
            dense_coeffs(1, hh, Cr);      // no pre-filter: every sample is evaluated
        } else {
            // thick disc of bounded height Hmax: a sample can only be inside if |z| = r |sin d| < Hmax,
            // and |sin d| >= (2/π)|d|, so |d| r >= (π/2) Hmax rules it out
            dense_coeffs(1, hh, Cr);
            const real hmax = 1.5707963267948966 * 1.000001
                              * (DISC == GR_DISC_TABULATED ? (real)p.cfg.disc_params[2]
                                                           : 3.0 * (real)p.cfg.disc_params[1] * (real)p.cfg.disc_params[0]);
            // a warped thin sheet is also hit within gtol |r| of its surface
            const real warp = (DISC == GR_DISC_TABULATED && p.cfg.disc_params[3] != 0.0)
                                  ? (real)(1.5707963267948966 * 1.000001) * (real)p.cfg.gtol : (real)0.0;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const real th = (real)(jj + 1) / 7.0;
                real d = dense_eval(x[2], hh, Ct, th) - 1.5707963267948966;
                d -= 3.141592653589793 * GR_RINT(d * 0.3183098861837907);
                const real rr = GR_FABS(dense_eval(x[1], hh, Cr, th));
                any |= (GR_FABS(d) * rr < hmax + warp * rr);
            }
            if (!any) return 0;
        }
        real Cp[4] = { 0.0, 0.0, 0.0, 0.0 };
        if (DISC == GR_DISC_PRECESSING_THIN) dense_coeffs(3, hh, Cp);
        for (int jj = 1; jj <= 6; ++jj) {
            const real th = (real)jj / 7.0;
            real s, c;
            sincos_fast(dense_eval(x[2], hh, Ct, th), s, c);
            const real cj = disc_cond4(p, dense_eval(x[1], hh, Cr, th), s, c,
                                       DISC == GR_DISC_PRECESSING_THIN ? dense_eval(x[3], hh, Cp, th) : (real)0.0);
            if ((real)ps * cj < 0.0) return jj;
        }
        return 0;
    }

    // Root-find the event on the dense output (left-biased), move the state there, run the
    // discrete callbacks on it.  Result: final (t, x, v).
    GR_DEV void resolve_event(const Params& p)
    {
        real Cr[4], Ct[4], Cp[4] = { 0.0, 0.0, 0.0, 0.0 };
        dense_coeffs(1, h, Cr);
        dense_coeffs(2, h, Ct);
        if (DISC == GR_DISC_PRECESSING_THIN) dense_coeffs(3, h, Cp);
        const int ps = sgn(cprev);
#ifdef GR_HOST_HARNESS
        dbg_e2 = 0.0;
#endif
        real lo = 0.0, hi = (ev_top >= 7) ? 1.0 : (real)ev_top / 7.0;
        real flo = cprev, fhi = 0.0;
        real theta = hi;
        int kbest = 0;           // GR_DISC_COMPOSITE: the component whose root is the event
        if constexpr (DISC == GR_DISC_COMPOSITE) {
            // every candidate component's root on [0, hi]; the earliest is the event (find_callback_time, VectorContinuousCallback)
            bool first = true;
#pragma unroll
            for (int k = 0; k < GR_COMP_MAX; ++k) {
                if (k >= p.cfg.comp_n || !((ev_mask >> k) & 1)) continue;
                auto f = [&](real th) {
                    real s, c;
                    sincos_fast(dense_eval(x[2], h, Ct, th), s, c);
                    return comp_cond(p, k, dense_eval(x[1], h, Cr, th), s, c);
                };
                const real thk = root_on_dense(f, sgn(comp_prev(k)), comp_prev(k), hi);
                if (first || thk < theta) { theta = thk; kbest = k; }
                first = false;
            }
        } else {
        {
            real s, c;
            sincos_fast(dense_eval(x[2], h, Ct, hi), s, c);
            fhi = disc_cond4(p, dense_eval(x[1], h, Cr, hi), s, c, DISC == GR_DISC_PRECESSING_THIN ? dense_eval(x[3], h, Cp, hi) : (real)0.0);
        }
        if (fhi != 0.0) {
            // Illinois regula falsi: brackets [lo, hi] with sign(f(lo)) == ps throughout, superlinear on
            // the smooth part of the condition and still convergent across its jump at the disc rim
            // The bracket is closed to 1e-13 of the step (Θ in [0, 1]): positions move by < 1e-11, two
            // orders below what differing step sequences do to any two implementations (§4 of DESIGN.md).
            int side = 0;
            for (int it = 0; it < 48; ++it) {
                const real w = hi - lo;
                if (w <= 1.0e-13) break;
                real mid = lo - flo * w / (fhi - flo);
                // on the plateau c = 1 outside the disc's radial range the condition is a step
                // function: bisection is the optimal bracketing method there
                const bool plateau = (flo == 1.0) || (fhi == 1.0);
                if (plateau || !(mid > lo && mid < hi)) {
                    mid = lo + 0.5 * w;
                    if (!(mid > lo && mid < hi)) break;
                }
                real s, c;
                sincos_fast(dense_eval(x[2], h, Ct, mid), s, c);
                const real fm = disc_cond4(p, dense_eval(x[1], h, Cr, mid), s, c,
                                           DISC == GR_DISC_PRECESSING_THIN ? dense_eval(x[3], h, Cp, mid) : (real)0.0);
                if (sgn(fm) == ps) {
                    lo = mid; flo = fm;
                    if (side < 0) fhi *= 0.5;
                    side = -1;
                } else {
                    hi = mid; fhi = fm;
                    if (side > 0) flo *= 0.5;
                    side = 1;
                }
#ifdef GR_HOST_HARNESS
                dbg_e2 += 1.0;
#endif
            }
            theta = lo;
        }
        }
        (void)kbest;
        // change_t_via_interpolation!: every component from the interpolant
        real xe[4], ve[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            real C[4];
            dense_coeffs(i, h, C);
            xe[i] = dense_eval(x[i], h, C, theta);
            dense_coeffs(4 + i, h, C);
            ve[i] = dense_eval(v[i], h, C, theta);
        }
#ifdef GR_REAL_IS_TAN2
        {
            // The event time depends on (α, β): c(x(λ*; α, β)) = 0  =>  ∂λ* = -∂c|_λ / (dc/dλ).  xe, ve above carry the
            // tangents at FIXED λ (Θ is a plain number); the state at the event moves by (ẋ, v̇) ∂λ* on top.  The
            // reference's ForwardDiff pass through the ContinuousCallback carries the same term (without it the
            // Jacobian ∂(ρ, g)/∂(α, β) is off by 13-45 % on the rₑ ≈ 5-8 rays checked in tests/test_tangent_host.py).
            real se, ce;
            sincos_fast(xe[2], se, ce);
            const real cv = (DISC == GR_DISC_COMPOSITE) ? comp_cond(p, kbest, xe[1], se, ce) : disc_cond4(p, xe[1], se, ce, xe[3]);
            // dc/dλ along the ray: the same condition on a state whose first tangent slot holds the velocity
            const real r1 = gr_t_along(xe[1].v, ve[1].v), t1 = gr_t_along(xe[2].v, ve[2].v), p1 = gr_t_along(xe[3].v, ve[3].v);
            real s1, c1;
            sincos_fast(t1, s1, c1);
            const double cdot = ((DISC == GR_DISC_COMPOSITE) ? comp_cond(p, kbest, r1, s1, c1) : disc_cond4(p, r1, s1, c1, p1)).a;
            if (cdot != 0.0) {
                const double la = -cv.a / cdot;
                GR_TAN_B(const double lb = -cv.b / cdot;)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    real C[4];
                    dense_coeffs(4 + i, h, C);
                    // v̇ = (1/h) d/dΘ of the velocity interpolant
                    const double acc = C[0].v + theta.v * (2.0 * C[1].v + theta.v * (3.0 * C[2].v + theta.v * 4.0 * C[3].v));
                    const double vel = ve[i].v;
                    xe[i].a += vel * la; ve[i].a += acc * la;
                    GR_TAN_B(xe[i].b += vel * lb; ve[i].b += acc * lb;)
                }
                t.a += la;
                GR_TAN_B(t.b += lb;)
            }
        }
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = xe[i]; v[i] = ve[i]; }
        t = t + theta * h;
        real s, c;
        sincos_fast(x[2], s, c);
        discrete_cb(p, x[1], x[2], c, status, flags);
    }

    // unpack_solution + apply_to_image!
    GR_DEV void finalize(const Metric& m, const Params& p, const LdsView& lds) { finalize(m, p, lds, NoColdStore{}); }

    template <class Cold_>
    GR_DEV void finalize(const Metric& m, const Params& p, const LdsView& lds, const Cold_& cs)
    {
        if (kContinuous && (flags & RAY_EVENT)) {
            if constexpr (Cold_::kParkA > 0) {
                // the event step's accelerations of stages 1..kParkA are still where step() parked them
                Cold_::park_fence();
#pragma unroll
                for (int q_ = 1; q_ <= Cold_::kParkA; ++q_)
#pragma unroll
                    for (int i_ = 0; i_ < 4; ++i_) A[q_][i_] = cs.ldA(q_, i_);
            }
            resolve_event(p);
            flags &= ~RAY_EVENT;
        }
        if (flags & GR_FLAG_MASK) status = GR_STATUS_NO_STATUS;
        const Cold& cd = cold_of(p);
        if (cd.tile_cost) {
            const int64_t ti = tile_cost_index(cd);
            if (ti >= 0) cd.tile_cost[ti] = (uint32_t)(nacc + nrej);
        }
        if (cd.out_mode == 1) {
            real x0[4], v0[4];
            constrained_u0(m, p, j, x0, v0);
            const int64_t oj = cd.out_global ? range_map(cd, j) : j;
            // with lds.point the record is laid down in LDS and the wave sends all 64 as runs of consecutive
            // addresses afterwards (gr_kernels.hpp, points_epilogue); without it each lane stores its own 152 bytes
            gr_point* o = lds.point ? reinterpret_cast<gr_point*>(lds.point) : cd.points + oj;
            if (lds.point) *lds.point_addr = (uint64_t)(cd.points + oj);
            o->status = status;
            o->flags = flags;
            o->lambda_min = p.cfg.lambda0;
            o->lambda_max = (double)t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o->x_init[q] = (double)x0[q];
                o->v_init[q] = (double)v0[q];
                o->x[q] = (double)x[q];
                o->v[q] = (double)v[q];
            }
        } else if (cd.out_mode >= 2) {
            // lineprofile(bins, ε, m, u, d, BinningMethod()), line-profiles.jl:186-197
            real s, c;
            sincos_fast(x[2], s, c);
            const real rho = x[1] * GR_FABS(s);
            const bool in = status == GR_STATUS_INTERSECTED_WITH_GEOMETRY && rho >= cd.lp_rmin && rho <= cd.lp_rmax;
            real g = 0.0;
            if (in) {
                real x0[4], v0[4];
                constrained_u0(m, p, j, x0, v0);
                g = redshift_pf(m, p, cd, lds, x0, v0, x, v);
            }
#ifdef GR_REAL_IS_TAN2
            if (cd.out_mode == 5) {
                // ray summary with tangents: (g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, t, status) -- what
                // jacobian_∂αβ_∂gr reads off its dual numbers (precision-solvers.jl:401-451)
                double* o = cd.lp_pairs + 8 * j;
#if GR_TAN_W == 1
                // a pair of lanes per ray: the even lane carries ∂/∂α and writes the value columns, the odd one ∂/∂β
                const int dir = tan_dir();
                if (dir == 0) {
                    o[0] = in ? g.v : __builtin_nan("");
                    o[1] = rho.v;
                    o[6] = x[0].v;
                    o[7] = (double)status;
                }
                o[2 + dir] = g.a;
                o[4 + dir] = rho.a;
#else
                o[0] = in ? g.v : __builtin_nan("");
                o[1] = rho.v;
                o[2] = g.a; o[3] = g.b;
                o[4] = rho.a; o[5] = rho.b;
                o[6] = x[0].v;
                o[7] = (double)status;
#endif
            } else
#endif
            if (cd.out_mode == 4) {
                // ray summary for the precision solvers: (g, ρ, t, status) -- 32 B instead of the
                // 152-B end-point record plus a second pass for the redshift
                cd.lp_pairs[4 * j] = in ? (double)g : __builtin_nan("");
                cd.lp_pairs[4 * j + 1] = (double)rho;
                cd.lp_pairs[4 * j + 2] = (double)x[0];
                cd.lp_pairs[4 * j + 3] = (double)status;
            } else if (cd.out_mode == 3) {
                cd.lp_pairs[2 * j] = in ? (double)g : __builtin_nan("");
                cd.lp_pairs[2 * j + 1] = in ? (double)rho : __builtin_nan("");
            } else if (in) {
                real area = 1.0;
                if (cd.sep_r) {
                    int64_t ii, jj;
                    sep_index(cd, sep_global(cd, j), ii, jj);
                    const double rr = cd.sep_r[ii];
                    area = (real)(rr * rr);                    // unnormalized_areas(::PolarPlane) = r_i², planes.jl:127-131
                } else if (cd.area) {
                    area = (real)cd.area[j];
                }
                // ε(r) g³ area with ε(r) = r^-q
                real eps;
                if (cd.lp_eps_n >= 2) {
                    // emissivity_at(prof::RadialDiscProfile, ρ): clamp, then NaNLinearInterpolator (interpolations.jl:7-26)
                    const int64_t ne = cd.lp_eps_n;
                    const double rc = fmin(fmax((double)rho, cd.lp_eps_r[0]), cd.lp_eps_r[ne - 1]);
                    int64_t a = 0, b = ne;                       // searchsortedlast: last index with eps_r[i] <= rc
                    while (a < b) {
                        const int64_t mid = (a + b) >> 1;
                        if (cd.lp_eps_r[mid] <= rc) a = mid + 1; else b = mid;
                    }
                    int64_t i0 = a - 1;
                    i0 = i0 < 0 ? 0 : (i0 > ne - 2 ? ne - 2 : i0);
                    const double x1 = cd.lp_eps_r[i0], x2 = cd.lp_eps_r[i0 + 1], y1 = cd.lp_eps_v[i0], y2 = cd.lp_eps_v[i0 + 1];
                    const double w = (rc - x1) / (x2 - x1);
                    double y = (1.0 - w) * y1 + w * y2;
                    if (y != y) {
                        const double pick = w < 0.5 ? y1 : y2;
                        y = (pick != pick) ? 0.0 : pick;
                    }
                    eps = (real)y;
                } else {
                    eps = (cd.lp_q == 3.0) ? rcp_full(rho * rho * rho) : GR_POW(rho, -cd.lp_q);
                }
                const real f = eps * g * g * g * area;
                // bucket(Simple(), g, f, bins): last edge <= g, clamped to the first / last bin
                // (the convention the reference's emissivity golden pins, test/unit/emissivity.jl:27-48)
                int64_t lo = 0, hi = cd.lp_nbins;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (cd.lp_edges[mid] <= g) lo = mid + 1; else hi = mid;
                }
                lo = lo > 0 ? lo - 1 : 0;
                if (f == f) gr_atomic_add((lds.hist ? lds.hist : cd.lp_flux) + lo, (double)f);
            }
        } else {
            bool pass = true;
            if (cd.pf.filter_id == GR_FILTER_EARLY_TERM) pass = t < p.cfg.lambda1;
            else if (cd.pf.filter_id == GR_FILTER_INTERSECTED) pass = status == GR_STATUS_INTERSECTED_WITH_GEOMETRY;
            real val = cd.pf.fill;
            if (pass) {
                if (cd.pf.pf_id == GR_PF_AFFINE_TIME) val = t;
                else if (cd.pf.pf_id == GR_PF_STATUS) val = (real)status;
                else if (cd.pf.pf_id == GR_PF_WINDING) val = (real)((uint32_t)flags >> 16);
                else if (cd.pf.pf_id == GR_PF_RADIUS) {
                    real s, c;
                    sincos_fast(x[2], s, c);
                    val = x[1] * GR_FABS(s);
                } else {
                    real x0[4], v0[4];
                    constrained_u0(m, p, j, x0, v0);
                    val = redshift_pf(m, p, cd, lds, x0, v0, x, v);
                }
            }
            cd.image[cd.out_global ? range_map(cd, j) : j] = (double)val;
        }
    }
};

}  // namespace GR_NS

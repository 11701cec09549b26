/*
 * gradus_oracle.h -- CPU ORACLE for the image-plane render path of Gradus.jl.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (gradus.jl_amd/csrc/libgradus_mi355x.so) never links,
 * loads or calls anything in this directory.
 *
 * It is a plain-C restatement of the reference algorithm (Julia, /root/reference,
 * Gradus.jl v0.4.30) for the path rendergeodesics/tracegeodesics ->
 * ensemble_solve_tracing_problem(EnsembleEndpointThreads) -> Tsit5 -> PointFunction.
 * Each function cites the reference file:line it follows.  The adaptive stepper,
 * step controller, initial-dt heuristic and callback semantics live in third-party
 * Julia packages that are NOT under /root/reference (OrdinaryDiffEq.jl, DiffEqBase.jl,
 * unpinned versions -- the reference has no Manifest.toml); their published
 * algorithms are restated here and anchored on the reference's
 * own call sites (src/tracing/tracing.jl:198-252, src/tracing/configuration.jl:99-103,
 * src/geometry/bootstrap.jl:43-54, src/tracing/charts.jl:9-23).
 *
 * Pinning: tests/test_oracle_golden.py checks this oracle against every golden
 * value the reference's tests hold for the path (test/smoke-tests/rendergeodesics.jl:43-67,
 * test/image-planes/test-{polar,cartesian}-grids.jl:13-21, test/unit/orthonormalization.jl,
 * test/smoke-tests/special-radii.jl, test/smoke-tests/circular-orbits.jl).
 */
#ifndef GRADUS_ORACLE_H
#define GRADUS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* StatusCodes, src/Gradus.jl:59-64 (EnumX, declaration order) */
enum {
    ORC_OUT_OF_DOMAIN = 0,
    ORC_WITHIN_INNER_BOUNDARY = 1,
    ORC_INTERSECTED_WITH_GEOMETRY = 2,
    ORC_NO_STATUS = 3
};

enum { ORC_METRIC_KERR = 0, ORC_METRIC_JOHANNSEN = 1, ORC_METRIC_MORRIS_THORNE = 2, ORC_METRIC_BUMBLEBEE = 3,
       ORC_METRIC_KERR_NEWMAN = 4, ORC_METRIC_JOHANNSEN_PSALTIS = 5, ORC_METRIC_DILATON_AXION = 6,
       ORC_METRIC_SPHERICAL = 7, ORC_METRIC_KERR_DARK_MATTER = 8, ORC_METRIC_KERR_REFRACTIVE = 9, ORC_METRIC_NOZ = 10,
       ORC_METRIC_TEST_BUMP = 100 /* a stand-in for a user-defined metric (metrics_tmpl.h): tests of the tabulated-metric path */ };
/* TABULATED mirrors the product's sampled ThickDisc; TORUS is the closure `_thick_disc` of the
 * reference's own smoke test (test/smoke-tests/rendergeodesics.jl:7-14) restated exactly, used to
 * pin the thick-disc golden value. */
enum { ORC_DISC_NONE = 0, ORC_DISC_THIN = 1, ORC_DISC_SHAKURA_SUNYAEV = 2, ORC_DISC_TABULATED = 3, ORC_DISC_TORUS = 4,
       ORC_DISC_DATUM = 5 /* DatumPlane(height = disc_params[0]), datum-plane.jl:1-10 */,
       ORC_DISC_ELLIPTICAL = 6, ORC_DISC_PRECESSING_THIN = 7 /* geometry/discs.jl:57-96 */,
       ORC_DISC_COMPOSITE = 8 /* CompositeGeometry, geometry/composite.jl + bootstrap.jl:76-110: comp_n components in comp[] */,
       ORC_DISC_MESH = 9 /* MeshAccretionGeometry, geometry/meshes.jl: disc_table = x/y/z extents (6) + 9 doubles per triangle,
                            disc_table_n = number of triangles; a DiscreteCallback on the step's Cartesian line element */ };
#define ORC_COMP_MAX 4
typedef struct {
    int32_t disc_id;        /* ORC_DISC_THIN | SHAKURA_SUNYAEV | ELLIPTICAL | DATUM */
    int32_t _pad;
    double disc_r_in, disc_r_out;
    double disc_params[4];
} orc_disc_component;

/* per-ray anomaly flags (SciML retcodes that EnsembleEndpointThreads swallows) */
enum { ORC_FLAG_MAXITERS = 1, ORC_FLAG_DTMIN = 2, ORC_FLAG_NAN = 4 };

typedef struct {
    int32_t metric_id;      /* ORC_METRIC_* */
    int32_t disc_id;        /* ORC_DISC_* */
    double params[8];       /* Kerr: M, a ; Johannsen: M, a, a13, a22, a52, e3 */
    double r_inner;         /* PolarChart.inner_radius (already x closest_approach) */
    double r_outer;         /* PolarChart.outer_radius */
    double disc_r_in, disc_r_out, gtol;
    double lambda0, lambda1;
    double abstol, reltol;
    double mu;              /* geodesic mass; 0 = null */
    int64_t maxiters;
    int32_t upper_hemisphere; /* domain_upper_hemisphere callback enabled */
    int32_t _pad;
    double hemi_delta;
    double disc_params[4];  /* ShakuraSunyaev: Mdot/Mdot_Edd, 1/eta ; TABULATED: rho0, rho1, hmax ; TORUS: centre, radius */
    const double* disc_table;
    int64_t disc_table_n;
    const double* chart_table;  /* PoloidalShapeChart: r_min(θ_k) on a uniform θ grid (charts.jl:26-48); n = 0: PolarChart */
    int64_t chart_table_n;
    double chart_theta0, chart_theta1;
    double q;               /* test-particle charge (TraceGeodesic.q); Lorentz force for Kerr-Newman only */
    int32_t count_windings; /* TraceWindings (tracing/photon-rings.jl): the count lands in bits 16..31 of orc_point.flags */
    int32_t _pad2;
    double winding_plane;   /* TraceWindings.plane_inc */
    int32_t comp_n;         /* ORC_DISC_COMPOSITE: number of components */
    int32_t _pad3;
    orc_disc_component comp[ORC_COMP_MAX];
} orc_config;

/* GeodesicPoint{Float64,Nothing}, src/solution-processing.jl:15-32; 152 bytes */
typedef struct {
    int32_t status;
    int32_t flags;          /* occupies the padding of the Julia struct */
    double lambda_min, lambda_max;
    double x_init[4], x[4], v_init[4], v[4];
} orc_point;

typedef struct {
    int32_t accepted, rejected, rhs_evals, cond_evals;
} orc_raystats;

/* Tsit5 tableau access for identity tests */
void orc_tsit5_tableau(double c[7], double a[7][7], double btilde[7], double r[7][4]);

/* metric_components (src/metrics/kerr-metric.jl:11-28, johannsen-ad.jl:12-34) with
 * its AD Jacobian (auto-diff.jl:206-211): g[5], dg/dr[5], dg/dtheta[5] */
void orc_metric_jacobian(const orc_config* c, double r, double th, double g[5], double dr[5], double dth[5]);
/* geodesic_equation, auto-diff.jl:213-226 */
void orc_geodesic_equation(const orc_config* c, const double x[4], const double v[4], double acc[4]);
/* constrain_time, auto-diff.jl:161-179 (positive root) */
double orc_constrain_time(const orc_config* c, const double x[4], const double v[4]);
/* lnrbasis (orthonormalization.jl:116-122) columns hcat'ed: Tx[mu][nu] row-major 4x4 */
void orc_lnrbasis(const orc_config* c, const double x[4], double Tx[16]);
void orc_lnrframe(const orc_config* c, const double x[4], double F[16]);
/* lnr_momentum_to_global_velocity_transform, tracing/utility.jl:32-40: Mx = ginv*Tx */
void orc_lnr_transform(const orc_config* c, const double x[4], double Mx[16]);
/* map_impact_parameters (utility.jl:66-87): unconstrained velocity for (alpha,beta) */
void orc_map_impact_parameters(const orc_config* c, const double x[4], double alpha, double beta, double v[4]);
/* _render_velocity_function, rendering/rendering.jl:140-163: v (unconstrained) for
 * 0-based linear pixel index i of an H x W column-major image */
void orc_render_velocities(const orc_config* c, const double x[4], double a0, double a1, double b0,
                           double b1, int64_t W, int64_t H, int64_t i0, int64_t n, double* v_out);

/* ensemble_solve_tracing_problem(::EnsembleEndpointThreads), tracing.jl:151-196.
 * xs: N x 4 positions (or one position if x_stride == 0); vs: N x 4 unconstrained
 * velocities (constrain_all is applied here, constraints.jl:14-15). */
/* jsf_algorithm (geometry/intersections.jl:58-101): segment Q1 -> Q2 against the triangle (V1, V2, V3) */
int orc_jsf(const double V1[3], const double V2[3], const double V3[3], const double Q1[3], const double Q2[3], double* tpar);
int orc_trace(const orc_config* c, const double* xs, int64_t x_stride, const double* vs, int64_t N,
              orc_point* out, orc_raystats* stats /* may be NULL */, int nthreads);

/* special radii: Kerr closed form (kerr-metric-first-order.jl:297-337) or generic
 * root find (special-radii.jl:14-60) */
double orc_isco(const orc_config* c);
/* CircularOrbits.fourvelocity(m, r) at (r, pi/2), circular-orbits.jl:114-123 */
void orc_circular_fourvelocity(const orc_config* c, double r, double v[4]);

/* point functions, const-point-functions.jl:26-79, point-functions.jl:81-127 */
enum { ORC_PF_AFFINE_TIME = 0, ORC_PF_REDSHIFT = 1, ORC_PF_STATUS = 2, ORC_PF_R = 3 };
enum { ORC_FILTER_NONE = 0, ORC_FILTER_EARLY_TERM = 1, ORC_FILTER_INTERSECTED = 2 };
typedef struct {
    int32_t pf_id, filter_id;
    double fill;            /* FilterPointFunction default, NaN normally */
    double r_isco;          /* for redshift */
    int64_t n_plunge;       /* plunging table (non-Kerr redshift); 0 for Kerr */
    const double* plunge_r; /* sorted ascending */
    const double* plunge_vt;
    const double* plunge_vr;
    const double* plunge_vphi;
} orc_pf;
void orc_apply_pf(const orc_config* c, const orc_pf* pf, const orc_point* pts, int64_t N, double max_time,
                  double* out, int nthreads);

/* plunging geodesic table for interpolate_plunging_velocities (orbit-solving.jl:137-167):
 * traces the mu=1 plunge from isco - delta and returns every accepted step. Returns the
 * number of rows written (<= cap). rows: r, vt, vr, vphi (unsorted, in step order) */
int64_t orc_plunging_table(const orc_config* c, double r_isco, double* r, double* vt, double* vr,
                           double* vphi, int64_t cap);

int64_t orc_trace_steps(const orc_config* c, const double x[4], const double v[4], orc_point* out,
                        double* t, double* r, int64_t cap);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif

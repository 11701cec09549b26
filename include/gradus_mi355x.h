/*
 * gradus_mi355x.h -- C ABI of libgradus_mi355x.so, the MI355X (gfx950) backend for the
 * image-plane render path of Gradus.jl.
 *
 * Drop-in boundary: Gradus.ensemble_solve_tracing_problem(ensemble, problem, config)
 * (reference: src/tracing/tracing.jl:113-196; the DiffEqGPU extension overloads the same
 * method at ext/GradusDiffEqGPUExt/GradusDiffEqGPUExt.jl:10-31).  A Julia method for a new
 * ensemble type `EnsembleMI355X` ccall's the entry points below (binding shown in
 * INTEGRATION.md); everything under the boundary -- constrain_all, the Tsit5 integration
 * of the second-order geodesic ODE with chart / disc callbacks, unpack_solution and the
 * built-in PointFunctions -- runs in hand-written HIP kernels.
 *
 * Conventions
 *   - every function returns 0 on success or a negative gr_status; nothing throws;
 *     gr_last_error() returns a thread-local message for the last failure.
 *   - "_device" entry points take DEVICE pointers and a hipStream_t (as void*) and are
 *     asynchronous on that stream; the host variants take HOST pointers, stage through
 *     the context and block until the result is in the caller's buffer.
 *   - the caller owns every buffer passed in; pointers are only used during the call
 *     (host variants) or until the stream work completes (device variants).
 *   - one gr_ctx is bound to one HIP device; a ctx is not thread-safe, distinct ctxs are
 *     independent.  There is NO CPU fallback: without a gfx950 device gr_ctx_create fails.
 */
#ifndef GRADUS_MI355X_H
#define GRADUS_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR_ABI_VERSION 8

typedef enum {
    GR_OK = 0,
    GR_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad size, unsorted limits ...           */
    GR_ERR_UNSUPPORTED = -2,      /* unknown metric / disc / point-function id             */
    GR_ERR_NO_DEVICE = -3,        /* no usable HIP device (this library has no CPU path)   */
    GR_ERR_HIP = -4,              /* a HIP runtime call failed; see gr_last_error()        */
    GR_ERR_OUT_OF_MEMORY = -5
} gr_status;

/* StatusCodes -- src/Gradus.jl:59-64 (EnumX, declaration order) */
enum {
    GR_STATUS_OUT_OF_DOMAIN = 0,
    GR_STATUS_WITHIN_INNER_BOUNDARY = 1,
    GR_STATUS_INTERSECTED_WITH_GEOMETRY = 2,
    GR_STATUS_NO_STATUS = 3
};

/* AbstractStaticAxisSymmetric metrics available on the device.
 *   KERR               src/metrics/kerr-metric.jl:11-28,62-72          params = {M, a}
 *   JOHANNSEN          src/metrics/johannsen-ad.jl:12-34,49-67         params = {M, a, α13, α22, α52, ϵ3}
 *   MORRIS_THORNE      src/metrics/morris-thorne-ad.jl:4-39            params = {b}
 *   BUMBLEBEE          src/metrics/bumblebee-ad.jl:6-52                params = {M, a, l}
 *   KERR_NEWMAN        src/metrics/kerr-newman-ad.jl:6-64 (null rays / q = 0 only)   params = {M, a, Q}
 *   JOHANNSEN_PSALTIS  src/metrics/johannsen-psaltis-ad.jl:4-46        params = {M, a, ϵ3} */
enum {
    GR_METRIC_KERR = 0,
    GR_METRIC_JOHANNSEN = 1,
    GR_METRIC_MORRIS_THORNE = 2,
    GR_METRIC_BUMBLEBEE = 3,
    GR_METRIC_KERR_NEWMAN = 4,
    GR_METRIC_JOHANNSEN_PSALTIS = 5,
    GR_METRIC_DILATON_AXION = 6,       /* DilatonAxion(M, a, β, b) src/metrics/dilaton-axion-ad.jl:49-70 */
    GR_METRIC_SPHERICAL = 7,           /* SphericalMetric() (flat space) src/metrics/minkowski.jl:1-15; no params */
    GR_METRIC_KERR_DARK_MATTER = 8,    /* KerrDarkMatter(M, a, M_dark_matter, Δr, rₛ) src/metrics/kerr-dark-matter.jl:6-70 */
    GR_METRIC_KERR_REFRACTIVE = 9,     /* KerrRefractive(M, a, n, corona_radius) src/metrics/kerr-refractive-ad.jl:8-58 */
    GR_METRIC_NOZ = 10,                /* NoZMetric(M, a, ϵ) src/metrics/noz-metric.jl:7-66 */
    /* ABI 7 -- ANY AbstractStaticAxisSymmetric metric: the reference's plugin contract is a struct and ONE method,
     * metric_components(m, (r, θ)) -> (g_tt, g_rr, g_θθ, g_ϕϕ, g_tϕ) (src/Gradus.jl:78-86, src/metrics/kerr-metric.jl:62-70);
     * ForwardDiff supplies the Jacobian (src/tracing/method-implementations/auto-diff.jl:206-211).  A closure cannot cross this
     * ABI, so the caller samples its metric_components on the nodes gr_metric_grid_nodes() names, gr_metric_table_fit() turns
     * the samples into piecewise polynomials (total degree 5 on patches geometric in r - r0 and uniform in θ), and the kernels
     * evaluate the five components and their (∂r, ∂θ) derivatives from that table (gr_config.metric_table; params unused).
     * See "tabulated metrics" below.  Every flavour of the kernels ("precision" 32: the table stays fp64 and is evaluated in double, the step is fp32). */
    GR_METRIC_TABULATED = 11
};

/* accretion geometry
 *   THIN             ThinDisc        src/geometry/discs/thin-disc.jl:9-26       disc_r_in, disc_r_out, gtol
 *   SHAKURA_SUNYAEV  ShakuraSunyaev  src/geometry/discs/shakura-sunyaev.jl:22-33 disc_r_in = inner_radius,
 *                    disc_params = {Ṁ/Ṁ_Edd, 1/η}; height 3 (1/η)(Ṁ/Ṁ_Edd)(1 - sqrt(r_in/ρ)); thick-disc
 *                    distance_to_disc of src/geometry/discs/thick-disc.jl:60-66 (no gtol)
 *   TABULATED        ThickDisc(f)    src/geometry/discs/thick-disc.jl:30-66: the user's cross_section
 *                    closure cannot run on the device, so the host samples it on a uniform ρ grid:
 *                    disc_params = {ρ_first, ρ_last, max height}, disc_table[disc_table_n] = f(ρ_k);
 *                    linear interpolation, height <= 0 (or ρ outside the grid) = no disc there.
 *                    disc_params[3] != 0: WarpedThinDisc (thin-disc.jl:42-66) -- the table is the SIGNED height
 *                    h(ρ) of a thin sheet, |h(ρ) - r cosθ| < gtol |r| within disc_r_in <= ρ <= disc_r_out
 *                    (disc_params[2] = max |h|)
 *   DATUM            DatumPlane      src/geometry/discs/datum-plane.jl:1-10: the plane z = r cosθ = height
 *                    (disc_params[0]), signed (no underside), no radial extent; what the transfer-function
 *                    solvers trace against for a thin disc (cunningham-transfer-functions.jl:1-5)
 *   ELLIPTICAL       EllipticalDisc  src/geometry/discs.jl:57-72: disc_r_in = inner_radius, disc_params = {semi_major,
 *                    semi_minor}; |z| < sqrt((1 - (r/a)²) b²) + gtol |r| for inner_radius <= r <= semi_major
 *   PRECESSING_THIN  PrecessingDisc(ThinDisc(r_in, r_out), β, γ)  src/geometry/discs.jl:74-96: the thin disc tilted by β
 *                    about the x axis and turned by γ about the spin axis; disc_params = {β, γ, cos β, sin β}
 *   COMPOSITE        CompositeGeometry(d1, d2, ...) = d1 ∘ d2  src/geometry/composite.jl:1-26, the VectorContinuousCallback of
 *                    src/geometry/bootstrap.jl:76-110: comp_n (2..GR_COMP_MAX) components in gr_config.comp[], each a THIN,
 *                    SHAKURA_SUNYAEV, ELLIPTICAL or DATUM geometry with its own radii and parameters (gtol is shared, as in the
 *                    reference); the ray ends at the EARLIEST intersection with any of them
 *   MESH             MeshAccretionGeometry(mesh)  src/geometry/meshes.jl:1-80: a triangle mesh in Cartesian coordinates
 *                    (x, y, z) = (r sinθ cosϕ, r sinθ sinϕ, r cosθ), tested with a DiscreteCallback: after every accepted step the
 *                    segment from the previous to the new position is tested (jsf_algorithm, src/geometry/intersections.jl:58-101,
 *                    ϵ = 1e-8, front faces only) against the triangles whose first vertex lies within 3 of the new position, when
 *                    the new position is strictly inside the bounding box; the ray ends AT the step's end (no root finding) with
 *                    IntersectedWithGeometry.  disc_table = x_min, x_max, y_min, y_max, z_min, z_max (the x/y/z_extent fields of
 *                    the reference's struct), then 9 doubles V1 V2 V3 per triangle; disc_table_n = number of triangles (>= 1);
 *                    gtol and the other disc fields are not read.  fp64 only: GR_ERR_UNSUPPORTED with "precision" 32 and in
 *                    the tangent entry points (gr_ray_tangent*).  Cost per accepted step inside the box: one pass over the
 *                    triangle list per wave (the reference's "naive implementation" loop, meshes.jl:53-64) */
enum { GR_DISC_NONE = 0, GR_DISC_THIN = 1, GR_DISC_SHAKURA_SUNYAEV = 2, GR_DISC_TABULATED = 3, GR_DISC_DATUM = 4,
       GR_DISC_ELLIPTICAL = 5, GR_DISC_PRECESSING_THIN = 6, GR_DISC_COMPOSITE = 7, GR_DISC_MESH = 8 };
#define GR_COMP_MAX 4

/* one component of a GR_DISC_COMPOSITE geometry: the fields of gr_config with the same names, per component */
typedef struct gr_disc_component {
    int32_t disc_id;          /* GR_DISC_THIN | SHAKURA_SUNYAEV | ELLIPTICAL | DATUM               */
    int32_t _pad;
    double disc_r_in, disc_r_out;
    double disc_params[4];
} gr_disc_component;

/* per-ray anomaly bits written next to the status (SciML retcodes MaxIters /
 * DtLessThanMin / Unstable, which EnsembleEndpointThreads discards, tracing.jl:250) */
enum { GR_FLAG_MAXITERS = 1, GR_FLAG_DTMIN = 2, GR_FLAG_NAN = 4, GR_FLAG_MASK = 0xFFFF /* bits 16..31: winding count */ };

/* TracingConfiguration (src/tracing/configuration.jl:3-29) + TraceGeodesic.μ
 * (src/tracing/tracing.jl:1-7) + the geometry callback's gtol (src/geometry/bootstrap.jl:8)
 * flattened to plain data. */
typedef struct gr_config {
    int32_t metric_id;        /* GR_METRIC_*                                             */
    int32_t disc_id;          /* GR_DISC_*                                               */
    double params[8];         /* metric parameters, see GR_METRIC_*                      */
    double r_inner;           /* PolarChart.inner_radius (charts.jl:3-6), already scaled */
    double r_outer;           /* PolarChart.outer_radius                                 */
    double disc_r_in;         /* ThinDisc.inner_radius                                   */
    double disc_r_out;        /* ThinDisc.outer_radius                                   */
    double gtol;              /* geometry tolerance, default 1e-2                        */
    double lambda0, lambda1;  /* λ_domain                                                */
    double abstol, reltol;    /* default 1e-9 (configuration.jl:1)                       */
    double mu;                /* geodesic mass μ (0 = null)                              */
    int64_t maxiters;         /* OrdinaryDiffEq default 1_000_000                        */
    int32_t upper_hemisphere; /* 1 = domain_upper_hemisphere callback (callbacks.jl:31)  */
    int32_t _pad;
    double hemi_delta;        /* its δ, default 1e-4                                     */
    double disc_params[4];    /* extra geometry parameters, see GR_DISC_*                */
    const double* disc_table; /* GR_DISC_TABULATED (disc_table_n samples) / GR_DISC_MESH (6 + 9 disc_table_n doubles): HOST */
    int64_t disc_table_n;     /*   pointer in every entry point (copied into the context); NULL / 0 otherwise            */
    /* PoloidalShapeChart (charts.jl:26-48, event_horizon_chart :61-70): inner boundary r_min(θ),
     * chart_table[k] = r_min(θ_k) on the uniform grid θ_k = chart_theta0 + k (chart_theta1 -
     * chart_theta0)/(n-1), already scaled by closest_approach; linear interpolation (end intervals
     * extended).  HOST pointer, copied into the context.  n = 0: PolarChart with r_inner.       */
    const double* chart_table;
    int64_t chart_table_n;
    double chart_theta0, chart_theta1;
    double q;                 /* test-particle charge, TraceGeodesic.q (tracing.jl:1-7): adds the
                                 Lorentz force q F^μ_ν v^ν (q/μ for μ != 0) for GR_METRIC_KERR_NEWMAN,
                                 kerr-newman-ad.jl:66-100; ignored by metrics without a field     */
    int32_t count_windings;   /* 1 = TraceWindings (src/tracing/photon-rings.jl:1-71): count the crossings of the
                                 cone θ = winding_plane at step ends; the count is returned in bits 16..31 of
                                 gr_point.flags and by GR_PF_WINDING                                  */
    int32_t _pad2;
    double winding_plane;     /* TraceWindings.plane_inc, default π/2                                */
    int32_t comp_n;           /* GR_DISC_COMPOSITE: number of components (2..GR_COMP_MAX), else 0    */
    int32_t _pad3;
    gr_disc_component comp[GR_COMP_MAX];
    /* ABI 7, GR_METRIC_TABULATED: the table gr_metric_table_fit() wrote (HOST pointer in every entry point; the library keeps a
     * device copy per context, keyed by the table's build id, so a table is uploaded once) and its length in doubles.
     * NULL / 0 for every other metric. */
    const double* metric_table;
    int64_t metric_table_n;
} gr_config;

/* GeodesicPoint{Float64,Nothing} -- src/solution-processing.jl:15-32.  152 bytes, same
 * field order and padding as the Julia isbits struct (status::Int32 enum + 4 pad bytes,
 * which carry the anomaly flags here). */
typedef struct gr_point {
    int32_t status;
    int32_t flags;
    double lambda_min, lambda_max;
    double x_init[4], x[4], v_init[4], v[4];
} gr_point;

/* The pixel -> initial-velocity closure of _render_velocity_function
 * (src/rendering/rendering.jl:140-163) as data. */
typedef struct gr_plane {
    double x_obs[4];          /* observer four-position                                   */
    double Mx[16];            /* row-major 4x4: ginv * hcat(lnrbasis(g)...) of
                                 lnr_momentum_to_global_velocity_transform
                                 (src/tracing/utility.jl:32-40), computed by the caller   */
    double alpha0, alpha1;    /* αlims                                                    */
    double beta0, beta1;      /* βlims                                                    */
    int64_t width, height;    /* image_width, image_height                                */
    double offset;            /* the +1e-6 of rendering.jl:158-159                        */
} gr_plane;

/* Which rays of the W*H image (0-based linear index i = x*H + y, Julia's column-major
 * H x W matrix) a call processes: local ray j in [0, count) is image ray
 *     i = first + ((j / block) * stride_blocks + 0) * block + (j % block)
 * i.e. blocks of `block` consecutive rays, every `stride_blocks`-th block.  A contiguous
 * range is {first, count, count, 1}.  Used to shard an image over GPUs. */
typedef struct gr_range {
    int64_t first;
    int64_t count;
    int64_t block;
    int64_t stride_blocks;
} gr_range;

/* Built-in point functions -- src/const-point-functions.jl:26-79, src/redshift.jl:192-276 */
enum {
    GR_PF_AFFINE_TIME = 0,    /* gp.λ_max                                                */
    GR_PF_REDSHIFT = 1,       /* ConstPointFunctions.redshift(m, x)                      */
    GR_PF_STATUS = 2,         /* Float64(gp.status)                                      */
    GR_PF_RADIUS = 3,         /* _equatorial_project(gp.x)                               */
    GR_PF_WINDING = 4         /* gp.aux.winding of a TraceWindings trace (photon-ring order) */
};
enum {
    GR_FILTER_NONE = 0,
    GR_FILTER_EARLY_TERM = 1, /* filter_early_term: gp.λ_max < max_time                  */
    GR_FILTER_INTERSECTED = 2 /* filter_intersected: status == IntersectedWithGeometry   */
};
typedef struct gr_pointfunction {
    int32_t pf_id;
    int32_t filter_id;
    double fill;              /* FilterPointFunction.default (NaN)                       */
    double r_isco;            /* isco(m), host-computed                                  */
    /* PlungingInterpolation table for non-Kerr redshift (src/orbits/orbit-solving.jl:99-131,
     * src/interpolations.jl:1-45); n_plunge = 0 selects the analytic Kerr branch.  HOST
     * pointers in every entry point (the table is copied into the context). */
    int64_t n_plunge;
    const double* plunge_r;
    const double* plunge_vt;
    const double* plunge_vr;
    const double* plunge_vphi;
    /* ABI 7: the four-velocity the photon's energy is measured against at its START, for GR_PF_REDSHIFT:
     * has_u_src = 0 [every image-plane caller]: the static observer (1, 0, 0, 0) of _redshift_dotproduct (redshift.jl:192-220);
     * has_u_src = 1: u_src -- energy_ratio(m, gp, v_src, v_disc) = (g v_init u_src) / (g v v_disc) of
     * src/corona/flux-calculations.jl:96-110, the redshift between a moving coronal source and the disc.                    */
    int32_t has_u_src;
    int32_t _pad_u;
    double u_src[4];
} gr_pointfunction;

/* aggregate counters of one call (device-side reductions) */
typedef struct gr_stats {
    int64_t rays;
    int64_t accepted_steps;
    int64_t rejected_steps;
    int64_t rhs_evals;
    int64_t flagged_rays;     /* rays with a GR_FLAG_* bit set                           */
    int64_t status_count[4];  /* histogram over StatusCodes                              */
    /* host variants only (ABI 5 tells the two apart): device time from the start of the call's work on the context's
     * stream to the end of its LAST TRACE KERNEL (staging of inputs and tables included) ...                            */
    double kernel_ms;
    /* ... and to the end of the last copy back into the caller's buffer: kernel_ms plus the D2H part.  The wall time
     * of the blocking call is this plus the launch / synchronisation latency of the host side.                        */
    double call_ms;
    /* *_multi entry points only (ABI 6; 0 from every other call): host wall-clock time the calling thread spent ENQUEUEING
     * this context's share -- staging its inputs and launching its kernel, before it moved on to the next context.  Small
     * against kernel_ms means the devices ran side by side; comparable to kernel_ms would mean the call had waited for
     * device k before it started device k+1 (tests/test_gpu_multi.py asserts the former, for pageable and pinned results). */
    double enqueue_ms;
} gr_stats;

typedef struct gr_ctx gr_ctx;

int32_t gr_abi_version(void);
const char* gr_last_error(void);

/* Create / destroy a context on HIP device `device`. */
int32_t gr_ctx_create(int32_t device, gr_ctx** out);
int32_t gr_ctx_destroy(gr_ctx* ctx);
/* Tuning knobs: key/value.  ("kernel", 0 = one ray per lane, 1 = persistent with wave-ballot
 * refill, 2 = chosen per launch [default]: image planes -> one ray per lane in one-wave workgroups,
 * line profiles and caller-ordered ray arrays -> persistent); ("block", threads per workgroup,
 * 0 = auto [default]: 64 for kernel 0, 256 for kernel 1); ("refill_threshold", idle lanes that
 * trigger a refill, 0 = auto [default]: 16 for the fp64 kernels, 32 for the fp32 ones); ("waves_per_simd"), ("swizzle"), ("lpt"), ("lpt_lane"), ("lds"),
 * ("precision", 64 | 32); ("pipeline", bands of the end-point return); ("hugepages", 1 [default] = large caller-owned
 * result buffers are madvise(MADV_HUGEPAGE)d before they are pre-faulted, 0 = the caller's mapping is left alone);
 * ("tangent_norm", 1 [default] = gr_ray_tangent's step-size controller sees values AND tangents -- DiffEqBase's norm on
 * Dual state, what the reference's solves under ForwardDiff use (src/tracing/precision-solvers.jl:73-131,401-451); two
 * independent integrators then agree on the Jacobians to 1e-6 (tests/test_oracle_tangent.py) --, 0 = values only: the
 * tangents ride on the very steps of the plain trace and are good to ~1e-5, 4e-3 on rays through the polar axis);
 * ("tangent_pairs", gr_ray_tangent's kernel shape: 0 = one lane carries a ray with both directions of the Jacobian (least work:
 * 1024² rays in 20.9 ms), 1 = a PAIR of lanes per ray, one direction each (shortest step: the reference's default line profile,
 * 909 launches of ~126 rays, in 1.16 s against 1.55 s), 2 [default] = pairs for launches that cannot fill the SIMDs anyway
 * (2 n lanes <= one wave per SIMD), one lane per ray beyond; same results to rounding either way);
 * ("xcd_spread", 1 [default] = rays in caller order on the one-ray-per-lane kernel are dealt to the 8 XCDs by the digit sum of
 * their chunk index instead of round-robin -- a periodic pattern in the rays, such as the polar-axis column of a 1024-wide grid of
 * impact parameters, otherwise lands on one or two XCDs (38 against 21 ms) --, 0 = chunk b to workgroup b);
 * ("sky_deal", 1 [default] = gr_corona_trace deals the rays of a sky source to the waves by what they will cost -- a counting sort
 * by a step count predicted from each ray's initial position and direction (its passage round the polar axis), the longest class
 * first, traced by the one-ray-per-lane kernel in 256-thread workgroups: a wave's rays take nearly the same number of steps and the
 * launch begins with its longest waves (10⁶ lamp-post samples: 5.9 against 8.4 ms in sample order) --, 0 = sample order; same
 * min / max and bins either way);
 * ("lds_points", 1 [default] = the one-ray-per-lane kernel sends a wave's 64 end-point records through LDS as runs of
 * consecutive addresses, 0 = every lane stores its own 152 bytes; same bytes either way); ("direct_host", 1 [default] =
 * gr_render_endpoints into a gr_host_alloc block lets the kernel store across the link itself -- no staging buffer, no
 * copy --, 0 = staged in HBM and copied in bands as for caller-owned memory); ("pinned_pool_mib", process-wide: bytes of freed
 * gr_host_alloc blocks kept page-locked for the next request, default 1024 in at most 4 blocks, 0 = empty the pool and keep nothing;
 * the pool is emptied when the last context is destroyed); ("pinned_max_mib", process-wide: bytes of gr_host_alloc blocks that may
 * be outstanding at once, default 8192 -- a request beyond it is refused with GR_ERR_OUT_OF_MEMORY and the caller falls back to
 * pageable memory: a garbage-collected caller sees a 100-byte wrapper, not the block behind it);
 * ("pinned_huge", process-wide: 1 [default] = gr_host_alloc blocks of 8 MiB and more are mappings on transparent huge pages
 * registered with the runtime (13 ms for 608 MiB), 0 = every block from hipHostMalloc (122-365 ms)). */
int32_t gr_ctx_set(gr_ctx* ctx, const char* key, int64_t value);

/* ---- pinned result buffers (ABI 5).  The reference allocates the result of ensemble_solve_tracing_problem itself
 * (`Vector{GeodesicPoint{T}}(undef, n)`, src/tracing/tracing.jl:179-183): pageable memory, which the D2H copy crosses at
 * ~30 GB/s after its pages have been faulted in.  A binding that lets the LIBRARY allocate that block (page-locked, mapped
 * for DMA once) and wraps it as its array -- Julia: unsafe_wrap(Array, Ptr{GeodesicPoint}(p), n) + a finalizer calling
 * gr_host_free -- lets the trace kernel store the records into it directly, across the link (gr_render_endpoints, gr_render:
 * no staging buffer, no copy; gr_ctx_set "direct_host").  Any host entry point accepts such a pointer wherever it takes a
 * caller-owned output buffer; callers that bring their own memory are served as before.  Blocks are registered
 * process-wide: gr_host_free does not look at `ctx` (it may be NULL, or a context that has been destroyed meanwhile --
 * finalizers of a garbage-collected host language run in no particular order), and destroying a context does not free them.
 * gr_host_free(ctx, NULL) is a no-op.  Blocks of 8 MiB and more are anonymous mappings on transparent huge pages that the
 * library registers with the runtime (608 MiB: 13 ms; hipHostMalloc takes 122-365 ms for the same block), and freed blocks
 * wait in a bounded process-wide pool for the next request of a similar size (gr_ctx_set "pinned_pool_mib"), which then
 * costs nothing.  The memory of a reused block holds the previous result until the next call overwrites it -- a call that
 * writes only PART of a block (a gr_range that is not the whole plane) leaves the rest as it found it.
 * PROCESS-WIDE STATE, the one exception to "no global state but the context handle" (SURVEY §8b): the registry of blocks, the
 * pool and its two limits ("pinned_pool_mib", "pinned_max_mib": set through any context, they apply to all) belong to the
 * process because the blocks must outlive contexts.  The library counts the bytes it has handed out and REFUSES
 * (GR_ERR_OUT_OF_MEMORY) a request that would take them past "pinned_max_mib" (default 8 GiB): bindings for garbage-collected
 * languages collect and retry, or fall back to pageable memory.  The pool is emptied when the last context is destroyed.
 * Only a destination that lies ENTIRELY inside one block is written by a kernel directly; anything else -- a pointer near the end
 * of a block, a block too small for the result -- takes the staged copy like caller-owned memory. */
int32_t gr_host_alloc(gr_ctx* ctx, int64_t bytes, void** out);
int32_t gr_host_free(gr_ctx* ctx, void* p);

/* ---- fused render: rendergeodesics / render_into_image! (rendering.jl:28-54,89-107) ----
 * image[j] (j local, see gr_range) = pf(m, trace(ray i), λ_max). */
int32_t gr_render_device(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane,
                         const gr_pointfunction* pf, const gr_range* range,
                         double* d_image /* device, range->count doubles */,
                         gr_stats* d_stats /* device, may be NULL */, void* hip_stream);
int32_t gr_render(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane,
                  const gr_pointfunction* pf, const gr_range* range,
                  double* image /* host, range->count doubles */, gr_stats* stats /* host, may be NULL */);

/* ---- ONE host thread, SEVERAL devices: what a Julia caller gets from `ensemble = EnsembleMI355X(devices)` at every entry
 * of the boundary (bench.py uses one process per GPU and an RCCL gather instead).  Every *_multi call has the same shape:
 *   phase 1  for each context: stage its inputs, launch its kernel on its own stream -- nothing here waits for a device
 *            (gr_stats.enqueue_ms records the host time each context took);
 *   phase 2  for each context: queue the copy of its share into its place in the caller's buffer (a copy into pageable memory
 *            may block the host until THAT device's kernel has finished -- every other device is already running);
 *   phase 3  wait for all of them.
 * There is no exchange between devices at all.  ctxs must be distinct contexts; they may live on different devices or (for
 * testing) on the same one.  stats[k] receives the counters of ctxs[k] (may be NULL).  With n = 1 a call is equivalent to its
 * single-context entry point and produces the same bytes.
 *
 * Image planes (gr_render_multi, gr_render_endpoints_multi): the columns are dealt to the n contexts block-cyclically in
 * groups of `block_cols` columns (0 = default 8, halved until it divides width / n) -- centre columns hold the long rays, so
 * a contiguous split would leave the middle devices working after the outer ones have finished.  Into caller-owned (pageable)
 * memory each device's share goes home with one strided D2H copy; into a block from gr_host_alloc every device's KERNEL
 * stores its pixels / records at their final place across the link (no staging buffer, no copy). */
int32_t gr_render_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_plane* plane,
                        const gr_pointfunction* pf, int64_t block_cols,
                        double* image /* host, width*height doubles */, gr_stats* stats /* n entries or NULL */);
/* prerendergeodesics / ensemble_solve_tracing_problem on the render closure (src/rendering/rendering.jl:56-87,
 * src/tracing/tracing.jl:151-196): the whole plane's end points, 152 B per ray in image order. */
int32_t gr_render_endpoints_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_plane* plane,
                                  int64_t block_cols, gr_point* points /* host, width*height */,
                                  gr_stats* stats /* n entries or NULL */);
/* tracegeodesics(m, xs, vs, ...) (tracing.jl:151-196 on the array inputs of geodesic-problem.jl:121-150): context k takes
 * the k-th contiguous share of the rays (shares are multiples of 64 rays, the last may be short or empty). */
int32_t gr_trace_endpoints_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const double* x, int64_t x_stride,
                                 const double* v, int64_t n_rays, gr_point* points /* host, n_rays */,
                                 gr_stats* stats /* n entries or NULL */);
/* (the ray-set entry points -- gr_rayset_endpoints_multi, gr_ray_summary_multi, gr_ray_tangent_multi,
 * gr_redshift_radius_multi, gr_lineprofile_multi -- are declared at the end of this header, behind gr_rayset) */

/* ---- endpoints of an image plane: prerendergeodesics / EndpointRenderCache
 * (rendering.jl:56-87,121-138) ---- */
int32_t gr_render_endpoints_device(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane,
                                   const gr_range* range, gr_point* d_points, gr_stats* d_stats,
                                   void* hip_stream);
int32_t gr_render_endpoints(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane,
                            const gr_range* range, gr_point* points, gr_stats* stats);

/* ---- tracegeodesics(m, xs, vs, ...; ensemble) -> Vector{GeodesicPoint}
 * (tracing.jl:151-196; input shapes of geodesic-problem.jl:121-150).
 * x: n x 4 positions, or a single position when x_stride == 0 (else x_stride == 4);
 * v: n x 4 UNCONSTRAINED velocities -- constrain_all (constraints.jl:14-15) is applied on
 * the device. */
int32_t gr_trace_endpoints_device(gr_ctx* ctx, const gr_config* cfg, const double* d_x,
                                  int64_t x_stride, const double* d_v, int64_t n,
                                  gr_point* d_points, gr_stats* d_stats, void* hip_stream);
int32_t gr_trace_endpoints(gr_ctx* ctx, const gr_config* cfg, const double* x, int64_t x_stride,
                           const double* v, int64_t n, gr_point* points, gr_stats* stats);

/* ---- tracegeodesics(m, x::SVector, v::SVector, ...): ONE geodesic with every accepted step
 * saved (the reference's single-problem solve with save_on = true, src/tracing/tracing.jl:88-110).
 * Used on its own and by interpolate_plunging_velocities (src/orbits/orbit-solving.jl:137-167),
 * which traces the μ = 1 plunge from the ISCO on the same integrator.
 * path: cap rows of 9 doubles (λ, x[4], v[4]); row 0 is the initial state, the last row the
 * final state (after any event).  *n_rows receives the number of rows the solve produced
 * (rows beyond cap are dropped but still counted). */
int32_t gr_trace_path(gr_ctx* ctx, const gr_config* cfg, const double* x /* 4 */,
                      const double* v /* 4, unconstrained */, int64_t cap, double* path /* host, cap x 9 */,
                      int64_t* n_rows, gr_point* endpoint /* host, may be NULL */);
/* The same for n geodesics at once (tracegeodesics(m, xs, vs, ...) with save_on = true, the
 * EnsembleProblem solve of src/tracing/tracing.jl:113-149; the reference's benchmark suite
 * benchmark/integrator/benchmark-tracing.jl "many-geodesic" cases): ray j fills path[j * cap * 9 ...],
 * n_rows[j], endpoints[j].  x_stride = 0: one position for all rays, 4: one per ray. */
int32_t gr_trace_paths(gr_ctx* ctx, const gr_config* cfg, const double* x, int64_t x_stride,
                       const double* v /* n x 4, unconstrained */, int64_t n, int64_t cap,
                       double* path /* host, n x cap x 9 */, int64_t* n_rows /* host, n */,
                       gr_point* endpoints /* host, n, may be NULL */);

/* ---- lineprofile(bins, ε, m, x, d, BinningMethod(); plane, callback) --
 * src/line-profiles.jl:152-198.  The rays are an AbstractImagePlane given by its impact parameters
 * (src/image-planes/planes.jl:70-184); velocities are map_impact_parameters(m, x, α_i, β_i) with no
 * pixel offset (planes.jl:180-184). */
typedef struct gr_rayset {
    double x_obs[4];
    double Mx[16];            /* as in gr_plane                                            */
    const double* alpha;      /* n impact parameters α                                     */
    const double* beta;       /* n impact parameters β                                     */
    const double* area;       /* n unnormalized_areas(plane), or NULL for 1                */
    int64_t n;
    const double* height;     /* GR_DISC_DATUM only: n plane heights, one DatumPlane per ray
                                 (datumplane(d, rₑ), datum-plane.jl:14-17: what the thick-disc
                                 transfer-function solvers trace against, one plane per emission
                                 radius), or NULL for cfg->disc_params[0]                   */
    /* Separable ray sets (ABI 4): a PolarPlane (src/image-planes/planes.jl:96-131) is the outer product of Nr radii and
     * Nθ angles -- α = r_i cos θ_j, β = r_i sin θ_j, unnormalized_areas = r_i² -- so its rays need not be materialised:
     * with sep_r != NULL the library forms them on the device from the three small tables (alpha / beta / area are
     * ignored; products are plain IEEE multiplies, i.e. bit-identical to the arrays the
     * reference builds).  Ray k of the set is visited in 8 x 8 tiles of (i, j) (sep_tiled = 1: 64 neighbouring rays per
     * wave; the order in which a histogram receives its rays is immaterial) or column-major, k = i + sep_nr j
     * (sep_tiled = 0: the order of vec(αs), for outputs that are indexed by ray). */
    const double* sep_r;      /* sep_nr radii                                              */
    const double* sep_cos;    /* sep_nt cosines                                            */
    const double* sep_sin;    /* sep_nt sines                                              */
    int64_t sep_nr, sep_nt;
    int32_t sep_tiled;
    int32_t sep_reserved;
    /* Which rays of the set's order this launch traces (one plane sharded over devices, each adding its partial
     * histogram -- gradus.jl_amd/distributed.py): local ray j is ray
     *     k = sep_first + (j / sep_block) * sep_stride + j % sep_block          (sep_block > 0: a block-cyclic deal)
     *     k = sep_first + j                                                     (sep_block = 0: a contiguous range)
     * and every k must be below sep_nr * sep_nt.  All zero and n = sep_nr * sep_nt: the whole plane. */
    int64_t sep_first, sep_block, sep_stride;
    /* Rays FROM A SOURCE INTO ITS SKY (ABI 7): tracegeodesics(m, model::AbstractCoronaModel, ...) / tracecorona
     * (src/corona/corona-models.jl:1-33,143-190) for a source at ONE position (lamp post, beamed point source, a ring's
     * representative point).  sky_sampler != 0 selects it: x_obs is the source position, Mx = T · diag(1, J) with T =
     * tetradframe(g, v_source) and J the Cartesian -> spherical Jacobian at the source (samplers.jl:81-99: v = T (E0, E0 J k̂));
     * ray j (sample number j + 1, 1-based as in samplers.jl:30-44) leaves in the direction k̂ = -(sinθ cosϕ, sinθ sinϕ, cosθ) with
     *   i     = j + 1 (sky_generator 0: GoldenSpiralGenerator) | (j + 1) / n (1: EvenGenerator) | sky_i[j] (2: the caller's
     *           numbers, e.g. RandomGenerator's rand() * n)
     *   θ     = EvenSampler (sky_sampler 1): acos(1 - i / n) lower hemisphere, acos(1 - 2 i / n) both;
     *           WeierstrassSampler (2): 2 atan(sqrt(sky_resolution / i)), mirrored to π - θ for odd i on both hemispheres
     *   ϕ     = mod(π (1 + √5) i, 2π) for the golden spiral, mod(2π i, 2π) otherwise.
     * alpha / beta / area / height / sep_* are not read. */
    int32_t sky_sampler;      /* 0 = off, 1 = EvenSampler, 2 = WeierstrassSampler */
    int32_t sky_both;         /* 0 = LowerHemisphere, 1 = BothHemispheres */
    int32_t sky_generator;    /* 0 = golden spiral, 1 = even, 2 = sky_i */
    int32_t sky_reserved;
    double sky_resolution;
    const double* sky_i;      /* n values for sky_generator 2 (host / device pointer like alpha), else NULL */
    /* ABI 8 -- a SHARE of a source's samples (what the *_multi entry points hand each context; a caller that shards by hand --
     * one process per GPU -- sets them itself): the n rays of this set are samples sky_first + 1 .. sky_first + n of sky_total in
     * all, i.e. the `n` of the formulas above is sky_total and ray j has sample number sky_first + j + 1 (sky_i still has one
     * entry per ray of THIS set).  sky_total = 0: the whole source (first 0, total n). */
    int64_t sky_first, sky_total;
    /* ABI 8 -- a source WITHOUT one position (the reference's DiscCorona, src/corona/models/extended.jl:165-181: every sample leaves
     * from a point of its own, sample_position_velocity per ray, corona-models.jl:1-33): GR_SKY_ROW doubles per ray,
     *     x[4]       the sample's position
     *     Mx[16]     its matrix T diag(1, J), row-major as above (x_obs and Mx of the set are then not read)
     *     u_cov[4]   g_μν(x) u_src^ν, the source's four-velocity there with its index lowered
     *     gt[4]      g_tμ(x)
     * and the rows of gr_ray_summary / gr_corona_trace hold g = energy_ratio against THAT sample's source velocity (pf->has_u_src
     * must be 0: the library forms (u_cov · v) / (gt · v) per ray and scales the static-observer ratio with it).  Host or device
     * pointer like alpha; NULL: one position. */
    const double* sky_rows;
} gr_rayset;
#define GR_SKY_ROW 28

typedef struct gr_binning {
    double r_min, r_max;      /* minrₑ, maxrₑ: only hits with r_min <= ρ <= r_max count    */
    double emissivity_index;  /* ε(r) = r^-q                                               */
    int64_t n_bins;
    const double* bin_edges;  /* n_bins values, ascending (the `bins` argument)            */
    /* ABI 4: a tabulated emissivity instead of the power law -- emissivity_at(prof::RadialDiscProfile, r)
     * (src/corona/radial.jl:15-18): r clamped to [eps_r[0], eps_r[eps_n-1]], then the NaNLinearInterpolator of
     * src/interpolations.jl:1-30 (linear between the neighbouring radii; a NaN node falls back to the nearer finite
     * neighbour, else 0).  eps_n >= 2 selects it; eps_n = 0: ε(r) = r^-emissivity_index. */
    const double* eps_r;      /* eps_n radii, ascending                                    */
    const double* eps_v;      /* eps_n emissivities                                        */
    int64_t eps_n;
} gr_binning;

/* flux[k] += ε(ρ) g³ area for every counted hit whose redshift g falls in bin k
 * (bucket(Simple(), g, f, bins): last edge <= g, clamped to the first / last bin).  NOT normalised;
 * the caller divides by sum(flux) (line-profiles.jl:197).  Device variant: rays->alpha/beta/area,
 * b->bin_edges and d_flux are device pointers; d_flux is zeroed by the call. */
int32_t gr_lineprofile_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                              const gr_pointfunction* pf, const gr_binning* b, double* d_flux,
                              gr_stats* d_stats, void* hip_stream);
int32_t gr_lineprofile(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                       const gr_pointfunction* pf, const gr_binning* b, double* flux, gr_stats* stats);
/* generic emissivity: (g, ρ) per ray (NaN, NaN for rays that do not count), n x 2 doubles */
int32_t gr_redshift_radius_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                                  const gr_pointfunction* pf, double r_min, double r_max,
                                  double* d_pairs, gr_stats* d_stats, void* hip_stream);
int32_t gr_redshift_radius(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                           const gr_pointfunction* pf, double r_min, double r_max, double* pairs,
                           gr_stats* stats);

/* ---- ray summaries for the precision solvers (find_offset_for_radius, jacobian_∂αβ_∂gr,
 * src/tracing/precision-solvers.jl:135-236,401-451): per ray of an impact-parameter set, 4 doubles
 * (g, ρ, t, status) = (redshift of the end point or NaN if the ray did not reach the geometry,
 * r |sinθ| and coordinate time of the end point, status code) -- everything one solver iteration
 * reads, in one launch and 32 B per ray. ---- */
int32_t gr_ray_summary_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                              const gr_pointfunction* pf, double* d_out /* n x 4 */, gr_stats* d_stats,
                              void* hip_stream);
int32_t gr_ray_summary(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                       const gr_pointfunction* pf, double* out /* host, n x 4 */, gr_stats* stats);

/* ---- The same rays with DUAL NUMBERS THROUGH THE INTEGRATOR: what jacobian_∂αβ_∂gr obtains from ForwardDiff.jacobian around
 * tracegeodesics (src/tracing/precision-solvers.jl:401-451) and the Cunningham transfer functions divide by
 * (src/transfer-functions/cunningham-transfer-functions.jl:337-387).  The integrator is instantiated on a scalar that carries
 * ∂/∂α and ∂/∂β (gr_tangent.hpp): the tangent equations are integrated with the very Tsit5 steps of the value, the event
 * time's dependence on (α, β) is added by implicit differentiation at the disc.  out: n x 8 doubles
 * (g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, t, status); g = NaN unless the ray met the geometry.  cfg->disc_id must not be NONE. */
int32_t gr_ray_tangent_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                              const gr_pointfunction* pf, double* d_out /* n x 8 */, gr_stats* d_stats,
                              void* hip_stream);
int32_t gr_ray_tangent(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                       const gr_pointfunction* pf, double* out /* host, n x 8 */, gr_stats* stats);

/* ---- end points of an impact-parameter ray set: tracegeodesics(m, x, i -> map_impact_parameters(m, x,
 * α[i], β[i]), d, ...; ensemble = EnsembleEndpointThreads()) as impact_parameters_for_radius_obscured
 * (src/tracing/precision-solvers.jl:363-372) and the thick-disc transfer-function workhorse
 * (src/transfer-functions/cunningham-transfer-functions.jl:253-300) call it. ---- */

int32_t gr_rayset_endpoints_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                                   gr_point* d_points /* n */, gr_stats* d_stats, void* hip_stream);
int32_t gr_rayset_endpoints(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays,
                            gr_point* points /* host, n */, gr_stats* stats);

/* ---- apply(pf, cache): evaluate a built-in point function on endpoint records
 * (point-functions.jl:98-101, rendering.jl:103-107) ---- */
int32_t gr_apply_pointfunction_device(gr_ctx* ctx, const gr_config* cfg, const gr_pointfunction* pf,
                                      const gr_point* d_points, int64_t n, double max_time,
                                      double* d_out, void* hip_stream);
int32_t gr_apply_pointfunction(gr_ctx* ctx, const gr_config* cfg, const gr_pointfunction* pf,
                               const gr_point* points, int64_t n, double max_time, double* out);

/* ---- corona -> disc on the device (ABI 7): emissivity_profile(m, d, model; n_samples, sampler) (src/corona/emissivity.jl:118-168,
 * src/corona/radial.jl:38-100) for a source at one position.  Two calls on ONE context, nothing else on that context in between:
 *   gr_corona_trace  traces the rays of a sky source (gr_rayset.sky_*) against the disc and keeps (g, ρ, t, status) per ray ON THE
 *                    DEVICE -- g = energy_ratio against pf->u_src and the disc's Keplerian / plunging velocity, ρ = r |sinθ| and
 *                    t = coordinate time of the hit -- and returns min / max ρ over the rays that hit and their number (what
 *                    the radial grid of radial.jl:60-66 is built from);
 *   gr_corona_bin    bins those rays by ρ into the caller's edges (bucket(Simple()): last edge <= ρ, clamped) and returns, per
 *                    bin, the count, Σ g and Σ t (radial.jl:70-84 takes the means).  out: 3 x n_edges doubles, n_edges <= 65536.
 *                    The sums are accumulated as integers on a fixed-point grid finer than an ulp of the values, so they do
 *                    not depend on the order of the additions: the same rays give the same bits on every run.             */
int32_t gr_corona_trace(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                        double* rho_min_max /* 2 */, int64_t* n_hits, gr_stats* stats);
int32_t gr_corona_bin(gr_ctx* ctx, const double* edges, int64_t n_edges, double* out /* 3 x n_edges */);
/* ABI 8 -- the same two calls over SEVERAL contexts from one host thread (see gr_render_multi for the shape every *_multi call
 * has): context k traces a contiguous share of the source's samples (gr_rayset.sky_first / sky_total) and keeps its rows; min / max ρ
 * and the hit count are those of all shares.  gr_corona_bin_multi bins every context's rows into the caller's edges on ONE common
 * fixed-point grid (from the largest |g|, |t| and the hit count of all shares) and adds the integer accumulators: the result is
 * bit for bit what one context gives for the whole source.  stats: n entries or NULL. */
int32_t gr_corona_trace_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                              double* rho_min_max /* 2 */, int64_t* n_hits, gr_stats* stats);
int32_t gr_corona_bin_multi(gr_ctx* const* ctxs, int32_t n, const double* edges, int64_t n_edges, double* out /* 3 x n_edges */);

/* ---- tabulated metrics (ABI 7; segments, axis terms: ABI 8; GR_METRIC_TABULATED): the AbstractMetric plugin interface on the device ----
 * Host-only functions (no context, no device): plan a grid, learn its nodes, fit, check.
 *
 *   gr_metric_grid g;  gr_metric_grid_plan(r_min, r_max, r0, m_r, n_theta, &g);
 *   gr_metric_grid_nodes(&g, r_nodes, theta_nodes);                 // g.n_r_nodes radii, g.n_theta_nodes angles
 *   samples[(a * g.n_theta_nodes + b) * 5 + k] = metric_components(m, (r_nodes[a], theta_nodes[b]))[k];   // the caller's metric
 *   gr_metric_table_fit(&g, samples, table, err);                   // table: g.table_doubles doubles
 *   cfg.metric_id = GR_METRIC_TABULATED; cfg.metric_table = table; cfg.metric_table_n = g.table_doubles;
 *
 * Radial patches are the m_r equal parts of every octave [2^e, 2^(e+1)) of r - r0 between r_min and r_max: choose r0 at (or just
 * inside) the event horizon -- inner_radius(m) -- and r_min = the chart's inner radius, so that the patches shrink geometrically
 * towards the pole of g_rr; r_max = the chart's outer radius (the entry points refuse a chart, an observer or a source outside
 * [r_min, r_max]: polynomials do not extrapolate).  Polar patches: n_theta equal parts of [0, π]; the components are
 * taken to be even about both poles (θ outside [0, π] is folded).  err[0..2] receive the fit's own estimates of the largest
 * truncation error of a component, of its ∂/∂ln(r - r0) and of its ∂/∂θ, each relative to the component's magnitude on the
 * patch: refine (m_r, n_theta) until they are below what the integration tolerance needs.
 *
 * ABI 8 -- metrics that are PIECEWISE in r (the reference's KerrDarkMatter, src/metrics/kerr-dark-matter.jl:12-20: kinks at rₛ and
 * rₛ + Δr; KerrRefractive, kerr-refractive-ad.jl:26 with utils.jl:158-168: jumps at corona_radius ± δx/2 and an arctangent step
 * of width δx/10⁴ between them) name their break radii: gr_metric_grid_plan_breaks starts a new radial SEGMENT at every break,
 * anchored there (octaves of r - break, behind a core of m_r equal parts of the first octave's width), so that no patch straddles
 * a break and the rows next to one are fitted on their own side only.  A break with scale > 0 is a feature of that width centred
 * there: the patches shrink geometrically towards it from BOTH sides, down to that scale.  Radii may be negative (a chart through
 * the throat of src/metrics/morris-thorne-ad.jl's wormhole: a break at 0 with scale b): a segment measures distances from its
 * anchor.  gr_metric_grid_plan is the call without breaks: one segment, anchored at r0. */
#define GR_METRIC_MAX_SEG 12
typedef struct gr_metric_break {
    double radius;            /* where metric_components changes form                                         */
    double scale;             /* 0: a kink or a jump AT radius; > 0: a smooth feature of this width centred at radius */
} gr_metric_break;
typedef struct gr_metric_segment {
    double r_lo, r_hi;        /* radii r_lo <= r < r_hi take this segment                                    */
    double anchor;            /* x = dir (r - anchor) > 0 inside the segment                                 */
    double xmin;              /* 2^e_lo                                                                       */
    double fit_lo, fit_hi;    /* the metric is smooth on (fit_lo, fit_hi) ⊇ [r_lo, r_hi): no node lies outside (±inf: no limit) */
    int32_t e_lo, e_hi;       /* octaves 2^e_lo .. 2^(e_hi + 1) of x                                          */
    int32_t first_row, n_rows; /* radial rows (core + e_hi - e_lo + 1) m_r of the table                       */
    int32_t dir;              /* +1 | -1                                                                      */
    int32_t core;             /* 1: m_r equal parts of [0, 2^e_lo) come first                                 */
} gr_metric_segment;
typedef struct gr_metric_grid {
    double r0;                /* origin of the radial octaves of segment 0                                   */
    double r_min, r_max;      /* radii the table covers (patch edges are rounded outwards)                   */
    int32_t e_min, n_oct;     /* segment 0: octaves 2^e_min .. 2^(e_min + n_oct) of r - r0                   */
    int32_t m_r, n_theta;     /* patches per octave, patches over [0, π]                                     */
    int32_t degree, fit_nodes; /* total degree of a patch polynomial (5), Chebyshev nodes per patch and direction (12) */
    int32_t pole_factor;      /* the form g_ϕϕ and g_tϕ are stored in (the caller may change it between plan and fit):
                                 1 [set by gr_metric_grid_plan]: divided by sin²θ -- both vanish like sin²θ on the axis of a regular
                                 axis-symmetric metric, and a polynomial with an ABSOLUTE error would leave g^ϕϕ with an unbounded
                                 RELATIVE one for rays that graze the axis; the kernels multiply the factor (and its derivative) back.
                                 2: g = K_m(r) + K_d(r) cos θ + sin²θ h(r, θ) -- a metric whose g_ϕϕ, g_tϕ do NOT vanish on the axis
                                 (an axion charge: the reference's DilatonAxion with β != 0): K_m ± K_d, the limits on the two poles,
                                 are taken from the samples nearest the poles and stored as polynomials per radial row.
                                 0: stored as sampled (the reference's MorrisThorneWormhole, g_ϕϕ ∝ sin θ)                        */
    int32_t n_seg;            /* radial segments (1 without breaks)                                          */
    int64_t n_r_nodes;        /* n_rows * fit_nodes                                                          */
    int64_t n_theta_nodes;    /* n_theta * fit_nodes                                                         */
    int64_t table_doubles;    /* length of the table gr_metric_table_fit writes                              */
    int32_t n_rows;           /* radial rows of all segments                                                 */
    int32_t reserved;
    gr_metric_segment seg[GR_METRIC_MAX_SEG];
} gr_metric_grid;
int32_t gr_metric_grid_plan(double r_min, double r_max, double r0, int32_t m_r, int32_t n_theta, gr_metric_grid* grid);
int32_t gr_metric_grid_plan_breaks(double r_min, double r_max, double r0, int32_t m_r, int32_t n_theta, int32_t n_breaks,
                                   const gr_metric_break* breaks /* n_breaks, any order */, gr_metric_grid* grid);
int32_t gr_metric_grid_nodes(const gr_metric_grid* grid, double* r_nodes, double* theta_nodes);
int32_t gr_metric_table_fit(const gr_metric_grid* grid, const double* samples /* n_r_nodes x n_theta_nodes x 5 */,
                            double* table /* table_doubles */, double* err /* 3, may be NULL */);
/* the table's value at one point, by the arithmetic the kernels use: (g, ∂r g, ∂θ g) of the five components -- for checking a
 * table against the caller's own Jacobian (Julia: Gradus.metric_jacobian) before tracing with it */
int32_t gr_metric_table_eval(const double* table, int64_t table_n, double r, double theta, double* g /* 5 */, double* dr /* 5 */,
                             double* dth /* 5 */);

/* ---- ONE host thread, SEVERAL devices, ray sets (see gr_render_multi for the shape every *_multi call has) ---- */
/* Ray sets with one output row per ray (gr_rayset_endpoints, gr_ray_summary, gr_ray_tangent, gr_redshift_radius): contiguous
 * shares as above.  A separable ray set must come whole and in ray order (sep_block = 0, sep_tiled = 0). */
int32_t gr_rayset_endpoints_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays,
                                  gr_point* points /* host, rays->n */, gr_stats* stats);
int32_t gr_ray_summary_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays,
                             const gr_pointfunction* pf, double* out /* host, rays->n x 4 */, gr_stats* stats);
int32_t gr_ray_tangent_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays,
                             const gr_pointfunction* pf, double* out /* host, rays->n x 8 */, gr_stats* stats);
int32_t gr_redshift_radius_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays,
                                 const gr_pointfunction* pf, double r_min, double r_max,
                                 double* pairs /* host, rays->n x 2 */, gr_stats* stats);
/* lineprofile(bins, ε, m, x, d, BinningMethod(); plane) (src/line-profiles.jl:152-198): the rays are dealt block-cyclically
 * (a separable plane: one block = one strip of 8 x 8 tiles, all radii of 8 neighbouring angles, so every device sees every
 * radius; ray arrays: contiguous shares), each device bins its own rays into its own histogram and the host adds the n
 * histograms in context order.  The order in which a bin receives its rays differs from the one-context call, so the sums
 * agree to rounding (1e-13 relative), not bit for bit.  A separable ray set must come whole (sep_block = 0). */
int32_t gr_lineprofile_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays,
                             const gr_pointfunction* pf, const gr_binning* b, double* flux /* host, n_bins */,
                             gr_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* GRADUS_MI355X_H */

# GradusMI355X.jl -- the reference-side binding a Gradus.jl maintainer would add.
#
# Selects the MI355X backend by dispatch on a new ensemble type at Gradus.jl's existing
# boundary, exactly as ext/GradusDiffEqGPUExt/GradusDiffEqGPUExt.jl:10-31 does for DiffEqGPU:
#
#     using Gradus, GradusMI355X
#     α, β, img = rendergeodesics(m, x, d, 2000.0; pf = pf, ensemble = EnsembleMI355X())
#
# Two methods are added, nothing in Gradus' src/ changes:
#
#   Gradus.ensemble_solve_tracing_problem(::EnsembleMI355X, problem, config; ...)   src/tracing/tracing.jl:151-196
#       the generic boundary: every caller that passes `ensemble = EnsembleMI355X()` (tracegeodesics,
#       prerendergeodesics, lineprofile(BinningMethod), tracecorona, ...) gets a Vector{GeodesicPoint}
#       traced on the device (gr_trace_endpoints / gr_render_endpoints).
#   Gradus.render_into_image!(image, trace, config::TracingConfiguration{...,<:EnsembleMI355X}; pf) rendering.jl:89-101
#       the fused path of rendergeodesics: when the velocity is the closure of _render_velocity_function
#       (rendering.jl:140-163) and `pf` is one of the built-in point functions, pixels are turned into rays on
#       the device and the point function is evaluated in the kernel (gr_render_multi: 8 B per ray leave the
#       GPU instead of 152 B).  Anything else falls through to the generic method above plus Gradus' own
#       apply_to_image!.
#
# What cannot cross the C ABI is detected and handed back to the CPU with a warning (SURVEY §8b): unknown metric or
# geometry types, solvers other than Tsit5, user callbacks other than domain_upper_hemisphere, save_on = true.
#
# All logic lives behind the C ABI (include/gradus_mi355x.h); this file only flattens Julia structs into the POD
# structs and ccall's.  It cannot be exercised in the build container (no Julia there); tests/test_julia_binding.py
# parses this file and checks every struct mirror and every ccall signature against the C header and against the
# ctypes binding that the GPU tests drive.
module GradusMI355X

using Gradus
using Gradus: TracingConfiguration, EnsembleProblem, GeodesicPoint, StatusCodes, AbstractTrace,
    KerrMetric, JohannsenMetric, ThinDisc, PolarChart, ConstPointFunctions, PointFunction
using StaticArrays
import SciMLBase

export EnsembleMI355X, SampledThickDisc, render_mi355x, lineprofile_mi355x, emissivity_profile_mi355x, winding_numbers, metric_table,
    selftest

const LIB = get(ENV, "GRADUS_MI355X_LIB", "libgradus_mi355x.so")
const ABI_VERSION = 8

# ---------------------------------------------------------------------------------------------------------------
# POD mirrors of include/gradus_mi355x.h (field order and types checked by tests/test_julia_binding.py)
# ---------------------------------------------------------------------------------------------------------------
struct GrDiscComponent               # == gr_disc_component: one geometry of a CompositeGeometry
    disc_id::Int32
    _pad::Int32
    disc_r_in::Float64
    disc_r_out::Float64
    disc_params::NTuple{4,Float64}
end
const _NO_COMPONENT = GrDiscComponent(Int32(0), Int32(0), 0.0, 0.0, (0.0, 0.0, 0.0, 0.0))
const _NO_COMPONENTS = (_NO_COMPONENT, _NO_COMPONENT, _NO_COMPONENT, _NO_COMPONENT)

struct GrConfig                      # == gr_config
    metric_id::Int32
    disc_id::Int32
    params::NTuple{8,Float64}
    r_inner::Float64
    r_outer::Float64
    disc_r_in::Float64
    disc_r_out::Float64
    gtol::Float64
    lambda0::Float64
    lambda1::Float64
    abstol::Float64
    reltol::Float64
    mu::Float64
    maxiters::Int64
    upper_hemisphere::Int32
    _pad::Int32
    hemi_delta::Float64
    disc_params::NTuple{4,Float64}
    disc_table::Ptr{Float64}
    disc_table_n::Int64
    chart_table::Ptr{Float64}
    chart_table_n::Int64
    chart_theta0::Float64
    chart_theta1::Float64
    q::Float64
    count_windings::Int32            # TraceWindings (tracing/photon-rings.jl): count in bits 16..31 of GeodesicPoint padding
    _pad2::Int32
    winding_plane::Float64
    comp_n::Int32                    # CompositeGeometry (geometry/composite.jl): its geometries, 2..4 of them
    _pad3::Int32
    comp::NTuple{4,GrDiscComponent}
    metric_table::Ptr{Float64}       # GR_METRIC_TABULATED (ABI 7): the table gr_metric_table_fit wrote, or C_NULL / 0
    metric_table_n::Int64
end

struct GrMetricBreak                 # == gr_metric_break: a radius where metric_components changes form (scale 0), or a feature that narrow
    radius::Float64
    scale::Float64
end

struct GrMetricSegment               # == gr_metric_segment: one radial segment of a tabulated metric's grid
    r_lo::Float64
    r_hi::Float64
    anchor::Float64
    xmin::Float64
    fit_lo::Float64
    fit_hi::Float64
    e_lo::Int32
    e_hi::Int32
    first_row::Int32
    n_rows::Int32
    dir::Int32
    core::Int32
end

const METRIC_MAX_SEG = 12            # GR_METRIC_MAX_SEG

struct GrMetricGrid                  # == gr_metric_grid: the patch grid of a tabulated metric
    r0::Float64
    r_min::Float64
    r_max::Float64
    e_min::Int32
    n_oct::Int32
    m_r::Int32
    n_theta::Int32
    degree::Int32
    fit_nodes::Int32
    pole_factor::Int32
    n_seg::Int32
    n_r_nodes::Int64
    n_theta_nodes::Int64
    table_doubles::Int64
    n_rows::Int32
    reserved::Int32
    seg::NTuple{12,GrMetricSegment}
end

struct GrStats                       # == gr_stats
    rays::Int64
    accepted_steps::Int64
    rejected_steps::Int64
    rhs_evals::Int64
    flagged_rays::Int64
    status_count::NTuple{4,Int64}
    kernel_ms::Float64               # start of the call's device work -> end of its last trace kernel
    call_ms::Float64                 # ... -> end of the last copy into the caller's buffer (ABI 5)
    enqueue_ms::Float64              # *_multi: host time spent enqueueing this context's share (ABI 6); 0 elsewhere
end

struct GrPlane                       # == gr_plane
    x_obs::NTuple{4,Float64}
    Mx::NTuple{16,Float64}           # ROW-major ginv * hcat(lnrbasis(g)...): Mx[4(i-1)+k] = M[i,k]
    alpha0::Float64
    alpha1::Float64
    beta0::Float64
    beta1::Float64
    width::Int64
    height::Int64
    offset::Float64
end

struct GrPointFunction               # == gr_pointfunction
    pf_id::Int32
    filter_id::Int32
    fill::Float64
    r_isco::Float64
    n_plunge::Int64
    plunge_r::Ptr{Float64}
    plunge_vt::Ptr{Float64}
    plunge_vr::Ptr{Float64}
    plunge_vphi::Ptr{Float64}
    has_u_src::Int32                 # 1 = the photon's starting energy is measured against u_src (energy_ratio, flux-calculations.jl:96-110)
    _pad_u::Int32
    u_src::NTuple{4,Float64}
end

struct GrRange                       # == gr_range
    first::Int64
    count::Int64
    block::Int64
    stride_blocks::Int64
end

struct GrRayset                      # == gr_rayset
    x_obs::NTuple{4,Float64}
    Mx::NTuple{16,Float64}
    alpha::Ptr{Float64}
    beta::Ptr{Float64}
    area::Ptr{Float64}
    n::Int64
    height::Ptr{Float64}
    sep_r::Ptr{Float64}              # a PolarPlane as three small tables: the device forms α = r cos θ, β = r sin θ, area = r²
    sep_cos::Ptr{Float64}
    sep_sin::Ptr{Float64}
    sep_nr::Int64
    sep_nt::Int64
    sep_tiled::Int32
    sep_reserved::Int32
    sep_first::Int64
    sep_block::Int64
    sep_stride::Int64
    sky_sampler::Int32               # rays from a source into its sky: 0 = off, 1 = EvenSampler, 2 = WeierstrassSampler
    sky_both::Int32                  # 0 = LowerHemisphere, 1 = BothHemispheres
    sky_generator::Int32             # 0 = GoldenSpiralGenerator, 1 = EvenGenerator, 2 = sky_i (RandomGenerator's numbers)
    sky_reserved::Int32
    sky_resolution::Float64
    sky_i::Ptr{Float64}
    sky_first::Int64                 # ABI 8: a share of a source's samples (0, 0 = the whole source)
    sky_total::Int64
    sky_rows::Ptr{Float64}           # ABI 8: a source without one position: 28 doubles per ray (x, Mx, lowered source velocity, g_tμ)
end

struct GrBinning                     # == gr_binning
    r_min::Float64
    r_max::Float64
    emissivity_index::Float64
    n_bins::Int64
    bin_edges::Ptr{Float64}
    eps_r::Ptr{Float64}              # a tabulated emissivity (RadialDiscProfile) instead of the power law, or C_NULL / 0
    eps_v::Ptr{Float64}
    eps_n::Int64
end

_check(rc) = rc == 0 || error(unsafe_string(ccall((:gr_last_error, LIB), Cstring, ())))

"What the device cannot do: caught at the boundary and handed back to a CPU ensemble."
struct UnsupportedOnDevice <: Exception
    msg::String
end

# This file reads values OUT OF CLOSURES of Gradus.jl (the geometry callback's `gtol`, the hemisphere callback's `δ`, the trace in
# the problem builder, the render closure's αs / βs): private layout that a refactor upstream may change without notice, and this
# file cannot be run where it was written.  Three guards make such a change loud instead of silently wrong:
#   * the Gradus.jl versions whose closure layouts were read are pinned (`GRADUS_TESTED`); any other version is an error unless
#     ENV["GRADUS_MI355X_ALLOW_UNTESTED_GRADUS"] = "1" (then a warning);
#   * every closure that is recognised must have EXACTLY the fields it had (`_expect_fields`);
#   * `GradusMI355X.selftest()` renders a small scene on the device and with `EnsembleEndpointThreads` and compares: the first
#     thing to run after installing or upgrading either side (INTEGRATION.md).
const GRADUS_TESTED = (v"0.4.30", v"0.5.0")      # [first, next): read against v0.4.30
function _check_gradus_version()
    v = pkgversion(Gradus)
    (!isnothing(v) && GRADUS_TESTED[1] <= v < GRADUS_TESTED[2]) && return
    msg = "GradusMI355X was written against Gradus.jl $(GRADUS_TESTED[1]) (closure layouts of bootstrap.jl, callbacks.jl, " *
          "geodesic-problem.jl, rendering.jl); this is $(something(v, "an unversioned checkout")). Run GradusMI355X.selftest() " *
          "and set ENV[\"GRADUS_MI355X_ALLOW_UNTESTED_GRADUS\"] = \"1\" if it passes."
    get(ENV, "GRADUS_MI355X_ALLOW_UNTESTED_GRADUS", "0") == "1" ? (@warn msg maxlog = 1) : error(msg)
end
"A closure this file reads from must have exactly the fields it had in the Gradus.jl versions of `GRADUS_TESTED`."
function _expect_fields(f, names::Symbol...)
    have = fieldnames(typeof(f))
    Set(have) == Set(names) ||
        error("GradusMI355X: the closure $(nameof(typeof(f))) has fields $have, this binding was written for $names -- " *
              "Gradus.jl changed underneath it; see GradusMI355X.selftest()")
    f
end

"""
    EnsembleMI355X(devices = [0])

One `gr_ctx` per listed HIP device, all driven from the calling Julia task.  With several devices every entry of the
boundary spreads its rays over them through the library's `*_multi` entry points: `rendergeodesics` and
`prerendergeodesics` deal the image's columns block-cyclically (`gr_render_multi`, `gr_render_endpoints_multi`),
`tracegeodesics` on arrays and image planes takes contiguous shares (`gr_trace_endpoints_multi`,
`gr_rayset_endpoints_multi`), `lineprofile_mi355x` adds one histogram per device (`gr_lineprofile_multi`).  The host
enqueues every device's share before it waits for any of them; there is no exchange between devices.
"""
mutable struct EnsembleMI355X
    devices::Vector{Int32}
    ctxs::Vector{Ptr{Cvoid}}
    function EnsembleMI355X(devices = [0])
        _check_gradus_version()
        abi = ccall((:gr_abi_version, LIB), Int32, ())
        abi == ABI_VERSION || error("GradusMI355X: libgradus_mi355x.so has ABI version $abi, this binding is written for $ABI_VERSION")
        ctxs = Ptr{Cvoid}[]
        for d in devices
            ref = Ref{Ptr{Cvoid}}(C_NULL)
            _check(ccall((:gr_ctx_create, LIB), Int32, (Int32, Ref{Ptr{Cvoid}}), d, ref))
            push!(ctxs, ref[])
        end
        ens = new(Int32.(collect(devices)), ctxs)
        finalizer(e -> foreach(c -> ccall((:gr_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), c), e.ctxs), ens)
        ens
    end
end
EnsembleMI355X(device::Integer) = EnsembleMI355X([device])

# ---------------------------------------------------------------------------------------------------------------
# Julia objects -> plain data
# ---------------------------------------------------------------------------------------------------------------
_metric(m::KerrMetric) = (Int32(0), (m.M, m.a, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::JohannsenMetric) = (Int32(1), (m.M, m.a, m.α13, m.α22, m.α52, m.ϵ3, 0.0, 0.0))
_metric(m::MorrisThorneWormhole) = (Int32(2), (m.b, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::BumblebeeMetric) = (Int32(3), (m.M, m.a, m.l, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::KerrNewmanMetric) = (Int32(4), (m.M, m.a, m.Q, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::JohannsenPsaltisMetric) = (Int32(5), (m.M, m.a, m.ϵ3, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::DilatonAxion) = (Int32(6), (m.M, m.a, m.β, m.b, 0.0, 0.0, 0.0, 0.0))
_metric(m::Gradus.SphericalMetric) = (Int32(7), (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::KerrDarkMatter) = (Int32(8), (m.M, m.a, m.M_dark_matter, m.Δr, m.rₛ, 0.0, 0.0, 0.0))
_metric(m::KerrRefractive) = (Int32(9), (m.M, m.a, m.n, m.corona_radius, 0.0, 0.0, 0.0, 0.0))
_metric(m::NoZMetric) = (Int32(10), (m.M, m.a, m.ϵ, 0.0, 0.0, 0.0, 0.0, 0.0))
# Any other static, axis-symmetric metric -- the reference's plugin contract: a struct `<: AbstractStaticAxisSymmetric` and ONE
# method, `metric_components(m, rθ)` (src/Gradus.jl:78-86, src/metrics/kerr-metric.jl:62-70) -- runs on the device from a table of
# that method's values (GR_METRIC_TABULATED, ABI 7; `metric_table` below).  Only what is not even that stays on the CPU.
_metric(m::Gradus.AbstractStaticAxisSymmetric) = (Int32(11), _tabulated_params(m))
function _tabulated_params(m)
    # The table carries `metric_components` and nothing else.  A type that ALSO brings its own `geodesic_equation` or
    # `metric_jacobian` (instead of the generic methods of src/tracing/method-implementations/auto-diff.jl:206-226), or an
    # electromagnetic potential for charged particles (kerr-newman-ad.jl:63), defines dynamics the table does not hold: those
    # stay with the reference's CPU ensemble rather than being traced as plain geodesics of the components.
    T = typeof(m)
    x4 = SVector{4,Float64}
    generic_ge = which(Gradus.geodesic_equation, Tuple{Gradus.AbstractStaticAxisSymmetric,x4,x4})
    generic_mj = which(Gradus.metric_jacobian, Tuple{Gradus.AbstractStaticAxisSymmetric,SVector{2,Float64}})
    (which(Gradus.geodesic_equation, Tuple{T,x4,x4}) === generic_ge && which(Gradus.metric_jacobian, Tuple{T,SVector{2,Float64}}) === generic_mj) ||
        throw(UnsupportedOnDevice("$T overrides geodesic_equation / metric_jacobian: its dynamics are not those of its metric_components alone"))
    hasmethod(Gradus.electromagnetic_potential, Tuple{T,SVector{2,Float64}}) &&
        throw(UnsupportedOnDevice("$T carries an electromagnetic potential: the table holds the metric only"))
    (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
end
_metric(m) = throw(UnsupportedOnDevice("metric $(typeof(m)) is not static and axis-symmetric: no device implementation"))
_is_tabulated(m) = _metric(m)[1] == Int32(11)

const _METRIC_TABLES = Dict{Any,Vector{Float64}}()      # (metric, r_min, r_max) -> fitted table; metrics are immutable structs
const METRIC_TABLE_TOL = (1e-10, 1e-7, 1e-7)             # the fit's own estimates a table must meet: value, ∂/∂ln(r - r0), ∂/∂θ

"""
    metric_breaks(m) -> Vector{Tuple{Float64,Float64}}

Radii where `metric_components(m, (r, θ))` changes form, as `(radius, scale)`: scale 0 for a kink or a jump AT the radius,
scale > 0 for a smooth feature of that width centred there.  A table's patches never straddle a break (ABI 8,
`gr_metric_grid_plan_breaks`); towards a break with a scale they shrink geometrically from both sides.  Extend it for a metric of
your own that is piecewise in r:  `GradusMI355X.metric_breaks(m::MyMetric) = [(m.r_shell, 0.0)]`.
"""
metric_breaks(m) = Tuple{Float64,Float64}[]
# src/metrics/kerr-dark-matter.jl:12-20: the enclosed mass is piecewise, C¹ at rₛ and rₛ + Δr
metric_breaks(m::KerrDarkMatter) = [(Float64(m.rₛ), 0.0), (Float64(m.rₛ + m.Δr), 0.0)]
# src/metrics/kerr-refractive-ad.jl:26 with src/utils.jl:158-168 (δx = 2.5, smoothing_offset = 1e4): jumps of 6e-5 in n at
# corona_radius ± δx/2 and an arctangent step of width δx / 1e4 at corona_radius
metric_breaks(m::KerrRefractive) = [(Float64(m.corona_radius) - 1.25, 0.0), (Float64(m.corona_radius), 2.5e-4), (Float64(m.corona_radius) + 1.25, 0.0)]

# a copy of an (immutable) grid with another storage form for g_ϕϕ, g_tϕ
_with_form(g::GrMetricGrid, form) = GrMetricGrid(ntuple(i -> fieldname(GrMetricGrid, i) === :pole_factor ? Int32(form) : getfield(g, i), fieldcount(GrMetricGrid))...)

"""
    metric_table(m, r_inner, r_outer; m_r = 24, n_theta = 96, refinements = 3, breaks = metric_breaks(m)) -> Vector{Float64}

The piecewise-polynomial table of `metric_components(m, (r, θ))` between the chart's radii that the kernels trace a
user-defined metric through.  The library names the sample nodes (`gr_metric_grid_plan[_breaks]`, `gr_metric_grid_nodes`), this
function evaluates `Gradus.metric_components` there -- the ONLY thing it asks of `m`, exactly the reference's contract --, the
library fits (`gr_metric_table_fit`) and reports its error estimates; the grid is refined until they meet
`METRIC_TABLE_TOL`.  Radial patches are geometric in `r - r0` with `r0` just inside `Gradus.inner_radius(m)`, and start anew
at every radius of `breaks`.  The azimuthal components are stored divided by sin²θ; for a metric whose g_ϕϕ, g_tϕ do not vanish
on the axis (DilatonAxion with β != 0) their limits on the two poles are taken out first (form 2), and where neither is smooth
(MorrisThorneWormhole: g_ϕϕ ∝ sin θ) they are stored as sampled (form 0).
Tables are cached per (metric, radii).  `gr_metric_table_eval` checks one against `Gradus.metric_jacobian` (see `selftest`).
"""
function metric_table(m, r_inner::Float64, r_outer::Float64; m_r = 24, n_theta = 96, refinements = 3, breaks = metric_breaks(m))
    get!(_METRIC_TABLES, (m, r_inner, r_outer)) do
        rh = Float64(Gradus.inner_radius(m))
        # the table starts a hair inside the chart; its octaves count from a tenth of that distance behind the horizon
        r_min = (rh > 0 && r_inner > rh) ? r_inner - 1e-3 * (r_inner - rh) : r_inner - 1e-9 * max(1.0, abs(r_inner))
        r0 = (rh > 0 && r_min > rh) ? rh - 0.1 * (r_min - rh) : r_min - max(1.0, abs(r_min))
        bs = [GrMetricBreak(Float64(b[1]), Float64(b[2])) for b in breaks if r_min < b[1] < r_outer]
        local table
        previous, last_err = Inf, [Inf, Inf, Inf]
        form = nothing                 # decided by the first fit: how g_ϕϕ, g_tϕ are stored (gr_metric_grid.pole_factor)
        for _ = 0:refinements
            grid = Ref{GrMetricGrid}()
            _check(ccall((:gr_metric_grid_plan_breaks, LIB), Int32, (Float64, Float64, Float64, Int32, Int32, Int32, Ptr{GrMetricBreak}, Ref{GrMetricGrid}),
                r_min, r_outer, r0, m_r, n_theta, length(bs), bs, grid))
            g = grid[]
            rn, tn = Vector{Float64}(undef, g.n_r_nodes), Vector{Float64}(undef, g.n_theta_nodes)
            _check(ccall((:gr_metric_grid_nodes, LIB), Int32, (Ref{GrMetricGrid}, Ptr{Float64}, Ptr{Float64}), grid, rn, tn))
            # samples[k, b, a] = component k at (r_a, θ_b): column-major (5, nθ, nr) is the C order [n_r][n_θ][5]
            samples = Array{Float64,3}(undef, 5, length(tn), length(rn))
            Threads.@threads for a in eachindex(rn)
                for b in eachindex(tn)
                    comps = Gradus.metric_components(m, SVector(rn[a], tn[b]))
                    for k = 1:5
                        samples[k, b, a] = comps[k]
                    end
                end
            end
            fit(f) = begin
                t, e = Vector{Float64}(undef, g.table_doubles), zeros(Float64, 3)
                _check(ccall((:gr_metric_table_fit, LIB), Int32, (Ref{GrMetricGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    Ref(_with_form(g, f)), samples, t, e))
                (t, e)
            end
            table, err = fit(form === nothing ? 1 : form)
            if form === nothing
                form = 1
                if maximum(err ./ METRIC_TABLE_TOL) > 1e4
                    # g_ϕϕ / sin²θ is not smooth on the axis of every metric: the same samples in the other two forms, the first that is
                    # two orders better is kept
                    for alt in (2, 0)
                        t2, e2 = fit(alt)
                        if maximum(e2 ./ METRIC_TABLE_TOL) < 1e-2 * maximum(err ./ METRIC_TABLE_TOL)
                            table, err, form = t2, e2, alt
                            break
                        end
                    end
                end
            end
            all(err .<= METRIC_TABLE_TOL) && return table
            # a degree-5 fit gains 1.5^6 = 11 when a smooth function's patches shrink by a third: one that gains less than 3 is looking at
            # a kink or a pole (a horizon inside [r_inner, r_outer], a piecewise-defined function), and more samples will not change that
            miss = maximum(err ./ METRIC_TABLE_TOL)
            stalled = miss > previous / 3 && miss > 1e3
            previous = miss
            last_err = err
            stalled && break
            (err[1] > METRIC_TABLE_TOL[1] || err[2] > METRIC_TABLE_TOL[2]) && (m_r = (3 * m_r + 1) ÷ 2)
            (err[1] > METRIC_TABLE_TOL[1] || err[3] > METRIC_TABLE_TOL[3]) && (n_theta = (3 * n_theta + 1) ÷ 2)
        end
        # Good enough to trace with (value 1e-7, derivatives 1e-4)?  Then warn.  Otherwise a sample was not finite or the metric has a
        # break nobody named: UnsupportedOnDevice sends the problem to the reference's own CPU ensemble, as for any configuration the
        # device does not take.
        if last_err[1] <= 1e-7 && last_err[2] <= 1e-4 && last_err[3] <= 1e-4
            @warn "EnsembleMI355X: the table of $(typeof(m)) misses the fit tolerances (estimates $last_err, asked $METRIC_TABLE_TOL); tracing with it"
            return table
        end
        throw(UnsupportedOnDevice("$(typeof(m)) is not smooth on r in [$r_min, $r_outer] (fit estimates $last_err): a horizon outside " *
                                  "inner_radius(m), or a piecewise-defined metric function whose break radii GradusMI355X.metric_breaks(m) does not name"))
    end
end
_metric_table(m, r_inner, r_outer) = _is_tabulated(m) ? metric_table(m, Float64(r_inner), Float64(r_outer)) : Float64[]

# (disc_id, disc_r_in, disc_r_out, disc_params)
_disc(::Nothing) = (Int32(0), 0.0, 0.0, (0.0, 0.0, 0.0, 0.0))
_disc(d::ThinDisc) = (Int32(1), Float64(d.inner_radius), Float64(d.outer_radius), (0.0, 0.0, 0.0, 0.0))
_disc(d::ShakuraSunyaev) = (Int32(2), Float64(d.inner_radius), Inf, (Float64(d.Ṁ_Ṁedd), Float64(d.inv_η), 0.0, 0.0))
_disc(d::DatumPlane) = (Int32(4), 0.0, 0.0, (Float64(d.height), 0.0, 0.0, 0.0))
_disc(d::EllipticalDisc) = (Int32(5), Float64(d.inner_radius), Inf, (Float64(d.semi_major), Float64(d.semi_minor), 0.0, 0.0))
_disc(d::PrecessingDisc{T,<:ThinDisc}) where {T} =
    (Int32(6), Float64(d.disc.inner_radius), Float64(d.disc.outer_radius), (Float64(d.β), Float64(d.γ), cos(d.β), sin(d.β)))
# CompositeGeometry(d1, d2, ...) = d1 ∘ d2 (geometry/composite.jl): the components cross as gr_config.comp[]
_disc(d::Gradus.CompositeGeometry) = (Int32(7), 0.0, 0.0, (0.0, 0.0, 0.0, 0.0))
# MeshAccretionGeometry (src/geometry/meshes.jl:1-22): the triangles and the bounding box travel in the disc table
_disc(d::Gradus.MeshAccretionGeometry) = (Int32(8), 0.0, 0.0, (0.0, 0.0, 0.0, 0.0))
_disc(d) = throw(UnsupportedOnDevice("geometry $(typeof(d)) has no device implementation"))

# (comp_n, comp[4]) of gr_config
_components(d) = (Int32(0), _NO_COMPONENTS)
function _components(d::Gradus.CompositeGeometry)
    n = length(d.geometry)
    2 <= n <= 4 || throw(UnsupportedOnDevice("a composite geometry of $n components (the device takes 2..4)"))
    comps = map(d.geometry) do g
        g isa Union{ThinDisc,Gradus.ShakuraSunyaev,Gradus.EllipticalDisc,Gradus.DatumPlane} ||
            throw(UnsupportedOnDevice("$(typeof(g)) as a component of a composite geometry"))
        did, rin, rout, dparams = _disc(g)
        GrDiscComponent(did, Int32(0), rin, rout, dparams)
    end
    (Int32(n), ntuple(k -> k <= n ? comps[k] : _NO_COMPONENT, 4))
end

"""
    SampledThickDisc(d::AbstractThickAccretionDisc, ρ_min, ρ_max; samples = 16384)

A thick disc whose `cross_section` closure cannot cross the ABI (`ThickDisc(f)`, `PolishDoughnut`, ...),
sampled on a uniform ρ grid for the device (GR_DISC_TABULATED: linear interpolation, height <= 0 = no
disc there).  Pass it as the geometry; keep it alive for the duration of the call (it owns the table).
"""
struct SampledThickDisc{D} <: Gradus.AbstractThickAccretionDisc{Float64}
    disc::D
    ρ_min::Float64
    ρ_max::Float64
    table::Vector{Float64}
end
# the same interpolation the device applies, so a CPU ensemble sees the identical surface
function Gradus.cross_section(d::SampledThickDisc, ρ)
    (d.ρ_min <= ρ <= d.ρ_max) || return -one(ρ)
    n = length(d.table)
    u = (ρ - d.ρ_min) * (n - 1) / (d.ρ_max - d.ρ_min)
    k = clamp(floor(Int, u), 0, n - 2)
    w = u - k
    (1 - w) * d.table[k+1] + w * d.table[k+2]
end
Gradus.inner_radius(d::SampledThickDisc) = d.ρ_min
function SampledThickDisc(d, ρ_min, ρ_max; samples = 16384)
    ρs = range(Float64(ρ_min), Float64(ρ_max), samples)
    SampledThickDisc(d, Float64(ρ_min), Float64(ρ_max), [Float64(Gradus.cross_section(d, ρ)) for ρ in ρs])
end
_disc(d::SampledThickDisc) = (Int32(3), 0.0, Inf, (d.ρ_min, d.ρ_max, maximum(d.table), 0.0))
_disc_table(d) = Float64[]
_disc_table(d::SampledThickDisc) = d.table
# GR_DISC_MESH: x_extent, y_extent, z_extent (meshes.jl:5-7), then V1 V2 V3 of every triangle (9 doubles)
function _disc_table(d::Gradus.MeshAccretionGeometry)
    tab = Float64[d.x_extent[1], d.x_extent[2], d.y_extent[1], d.y_extent[2], d.z_extent[1], d.z_extent[2]]
    sizehint!(tab, 6 + 9 * length(d.mesh))
    for tri in d.mesh, vert in tri, c in vert
        push!(tab, Float64(c))
    end
    tab
end
# gr_config.disc_table_n: samples of a tabulated profile, TRIANGLES of a mesh
_disc_table_n(d, dtab) = length(dtab)
_disc_table_n(d::Gradus.MeshAccretionGeometry, dtab) = length(d.mesh)

# chart -> (r_inner, r_outer, table, θ_first, θ_last).  A PoloidalShapeChart built by
# event_horizon_chart wraps LinearInterpolation(r_min(θ_k), θ_k) on a uniform θ range (charts.jl:61-70).
_chart(c::PolarChart) = (Float64(c.inner_radius), Float64(c.outer_radius), Float64[], 0.0, 0.0)
function _chart(c::Gradus.PoloidalShapeChart)
    θ = collect(Float64, c.shapefunc.t)
    all(isapprox.(diff(θ), θ[2] - θ[1]; rtol = 1e-9)) || throw(UnsupportedOnDevice("the chart's θ grid must be uniform"))
    tab = collect(Float64, c.shapefunc.u)
    (minimum(filter(!isnan, tab)), Float64(c.outer_radius), tab, θ[1], θ[end])
end
_chart(c) = throw(UnsupportedOnDevice("chart $(typeof(c)) has no device implementation"))

# ---------------------------------------------------------------------------------------------------------------
# What the reference leaves inside closures: gtol, the hemisphere callback, the trace
# ---------------------------------------------------------------------------------------------------------------
_closure_name(f) = String(nameof(typeof(f)))

"""
    _callbacks(config) -> (gtol, δ_or_nothing)

`tracing_configuration(trace, m, x, v, geometry, ...; gtol, callback)` (src/geometry/bootstrap.jl:1-22) CONSUMES `gtol`
and merges `geometry_collision_callback(geometry, trace; gtol)` -- a `ContinuousCallback` whose condition is the
closure `_distance_to_disc_wrapper` over `(g, gtol)` (bootstrap.jl:43-60) -- into `config.callback` together with any
user callback (`merge_callbacks`, src/tracing/callbacks.jl:13-23).  So for every trace with a disc `config.callback`
is NOT nothing.  The geometry callback is what the device implements itself (from `config.geometry`); it is recognised
here, `gtol` is read back out of its closure, and only what remains is a *user* callback:
`domain_upper_hemisphere(δ)` (callbacks.jl:31-40, closure `_domain_upper_hemisphere_check` over `δ`) runs on the device,
anything else is refused.
"""
function _callbacks(config::TracingConfiguration)
    gtol = 1e-2          # bootstrap.jl:8 default; overwritten by what the geometry callback carries
    δ = nothing
    cb = config.callback
    isnothing(cb) && return (gtol, δ)
    conts, discs = if cb isa SciMLBase.CallbackSet
        (cb.continuous_callbacks, cb.discrete_callbacks)
    elseif cb isa Union{SciMLBase.ContinuousCallback,SciMLBase.VectorContinuousCallback}
        ((cb,), ())
    elseif cb isa SciMLBase.DiscreteCallback
        ((), (cb,))
    elseif cb isa Tuple
        (filter(c -> c isa Union{SciMLBase.ContinuousCallback,SciMLBase.VectorContinuousCallback}, cb),
            filter(c -> c isa SciMLBase.DiscreteCallback, cb))
    else
        throw(UnsupportedOnDevice("callback of type $(typeof(cb))"))
    end
    seen_geometry = false
    for c in conts
        cond = c.condition
        if !seen_geometry && config.geometry isa Gradus.CompositeGeometry && c isa SciMLBase.VectorContinuousCallback &&
           hasproperty(cond, :callbacks)
            # geometry_collision_callback(::CompositeGeometry) (bootstrap.jl:76-110): `_composite_condition` closes over
            # `callbacks`, one (condition, affect) pair per geometry, every condition over its own (g, gtol)
            cbs = cond.callbacks
            ok = length(cbs) == length(config.geometry.geometry) && all(1:length(cbs)) do k
                ck = cbs[k][1]
                hasproperty(ck, :g) && hasproperty(ck, :gtol) && ck.g === config.geometry.geometry[k] &&
                    !isnothing(_expect_fields(ck, :g, :gtol))
            end
            ok || throw(UnsupportedOnDevice("the composite geometry's callback does not match config.geometry"))
            gtol = Float64(cbs[1][1].gtol)
            seen_geometry = true
        elseif !seen_geometry && !isnothing(config.geometry) && hasproperty(cond, :g) && hasproperty(cond, :gtol) &&
           cond.g === config.geometry
            _expect_fields(cond, :g, :gtol)                 # _distance_to_disc_wrapper's closure, bootstrap.jl:43-60
            gtol = Float64(cond.gtol)
            seen_geometry = true
        else
            throw(UnsupportedOnDevice("a user ContinuousCallback ($(_closure_name(cond)))"))
        end
    end
    for c in discs
        cond = c.condition
        if !seen_geometry && config.geometry isa Gradus.MeshAccretionGeometry && hasproperty(cond, :g) && cond.g === config.geometry
            # geometry_collision_callback(::MeshAccretionGeometry) (meshes.jl:67-78): a DiscreteCallback whose condition closes
            # over the mesh `g`; the device tests the same line element itself (GR_DISC_MESH)
            seen_geometry = true
        elseif occursin("_domain_upper_hemisphere_check", _closure_name(cond)) && hasproperty(cond, :δ)
            _expect_fields(cond, :δ)                         # callbacks.jl:31-40
            δ = Float64(cond.δ)
        else
            throw(UnsupportedOnDevice("a user DiscreteCallback ($(_closure_name(cond))); only domain_upper_hemisphere runs on the device"))
        end
    end
    !isnothing(config.geometry) && config.geometry isa Union{Gradus.AbstractAccretionDisc,Gradus.CompositeGeometry,Gradus.MeshAccretionGeometry} && !seen_geometry &&
        throw(UnsupportedOnDevice("the geometry's collision callback was not found in config.callback"))
    (gtol, δ)
end

"""
    _trace_of(problem) -> AbstractTrace

`solve_tracing_problem(problem, config; solver_opts...)` (tracing.jl:83-87) does not forward the trace; it lives in the
problem builder that `assemble_tracing_problem` closed over (geodesic-problem.jl:141-150: `_problem_builder` captures
`trace`, `config`, `cbs`; `wrap_arguments` :158-191 captures it as `_problem_func` in `prob_func`).  μ and q are read
from there, not from keyword arguments that never arrive.
"""
function _trace_of(problem::EnsembleProblem)
    pf = problem.prob_func
    if hasproperty(pf, :_problem_func) && hasproperty(getproperty(pf, :_problem_func), :trace)
        return getproperty(pf, :_problem_func).trace
    end
    throw(UnsupportedOnDevice("cannot recover the trace (μ, q) from the problem builder"))
end

# Returns the config and the arrays it points into (keep them alive for the duration of the call).
function _config(config::TracingConfiguration, trace::AbstractTrace; maxiters = 1_000_000)
    config.solver isa Gradus.Tsit5 || throw(UnsupportedOnDevice("solver $(typeof(config.solver)); the device integrates with Tsit5"))
    trace isa Union{Gradus.TraceGeodesic,Gradus.TraceWindings} || throw(UnsupportedOnDevice("trace $(typeof(trace))"))
    id, params = _metric(config.metric)
    did, rin, rout, dparams = _disc(config.geometry)
    r_in, r_out, tab, θ0, θ1 = _chart(config.chart)
    dtab = _disc_table(config.geometry)
    comp_n, comps = _components(config.geometry)
    gtol, δ = _callbacks(config)
    mtab = _metric_table(config.metric, r_in, r_out)
    windings = trace isa Gradus.TraceWindings
    q = hasproperty(trace, :q) ? Float64(trace.q) : 0.0
    (q == 0.0 || config.metric isa KerrNewmanMetric) || throw(UnsupportedOnDevice("charged test particles outside Kerr-Newman"))
    cfg = GrConfig(id, did, params, r_in, r_out, rin, rout, gtol,
        Float64(config.λ_domain[1]), Float64(config.λ_domain[2]), Float64(config.abstol), Float64(config.reltol),
        Float64(trace.μ), maxiters,
        isnothing(δ) ? Int32(0) : Int32(1), Int32(0), isnothing(δ) ? 1e-4 : δ, dparams,
        isempty(dtab) ? Ptr{Float64}(C_NULL) : pointer(dtab), _disc_table_n(config.geometry, dtab),
        isempty(tab) ? Ptr{Float64}(C_NULL) : pointer(tab), length(tab), θ0, θ1,
        q, windings ? Int32(1) : Int32(0), Int32(0), windings ? Float64(trace.plane_inc) : π / 2,
        comp_n, Int32(0), comps,
        isempty(mtab) ? Ptr{Float64}(C_NULL) : pointer(mtab), length(mtab))
    cfg, (tab, dtab, mtab)
end

"""
    winding_numbers(points)

`gp.aux.winding` of end points traced with `trace = TraceWindings(...)` on `EnsembleMI355X`: the library returns
`GeodesicPoint{Float64,Nothing}` records and carries the count in bits 16..31 of the four padding bytes after `status`.
"""
winding_numbers(points::Vector{<:GeodesicPoint}) = GC.@preserve points [
    Int(unsafe_load(Ptr{UInt32}(pointer(points, i)) + 4) >> 16) for i in eachindex(points)
]

# ---------------------------------------------------------------------------------------------------------------
# The render closure and the built-in point functions, recognised structurally
# ---------------------------------------------------------------------------------------------------------------
"""
    _render_plane(config) -> GrPlane or nothing

`_render_velocity_function` (rendering.jl:140-163) returns the closure `velfunc(i)` over `αs`, `βs`, `image_height`,
`xfm`, `position`.  Its ranges give back `αlims`, `βlims`, `W`, `H`; `Mx = inv(g) * hcat(lnrbasis(g)...)` is what
`lnr_momentum_to_global_velocity_transform` built (tracing/utility.jl:32-40).  Julia matrices are column-major, the ABI
wants `Mx` row-major: hence `permutedims`.
"""
function _render_plane(config::TracingConfiguration)
    v = config.velocity
    (v isa Function && hasproperty(v, :αs) && hasproperty(v, :βs) && hasproperty(v, :image_height)) || return nothing
    x = config.position
    x isa SVector{4} || return nothing
    αs, βs = v.αs, v.βs
    H = Int64(v.image_height)
    (length(βs) == H && length(αs) * H == config.trajectories) || return nothing
    g = Gradus.metric(config.metric, x)
    Mx = inv(g) * hcat(Gradus.lnrbasis(g)...)
    GrPlane(Tuple(Float64.(x)), Tuple(Float64.(permutedims(Mx))), Float64(first(αs)), Float64(last(αs)),
        Float64(first(βs)), Float64(last(βs)), Int64(length(αs)), H, 1e-6)
end

const _AFFINE_F = typeof(ConstPointFunctions.affine_time().f)
const _EARLY_F = typeof(ConstPointFunctions.filter_early_term().f)
const _INTERSECTED_F = typeof(ConstPointFunctions.filter_intersected().f)

# base function -> (pf_id, r_isco, plunging table or nothing)
function _builtin_base(f, m)
    f isa _AFFINE_F && return (Int32(0), 0.0, nothing)
    if f === Gradus._redshift_guard
        # redshift(::KerrMetric, _) = PointFunction(_redshift_guard) (const-point-functions.jl:76): analytic plunge
        m isa KerrMetric || return nothing
        return (Int32(1), Float64(Gradus.isco(m)), nothing)
    end
    if hasproperty(f, :plunging_interpolation) && hasproperty(f, :isco)
        # _interpolate_redshift_closure (redshift.jl:246-276): PlungingInterpolation holds three NaNLinearInterpolators
        # over the same radii (orbit-solving.jl:99-131, interpolations.jl:1-45)
        pintrp = f.plunging_interpolation
        pintrp.m == m || return nothing
        tab = (collect(Float64, pintrp.t.t), collect(Float64, pintrp.t.u), collect(Float64, pintrp.r.u), collect(Float64, pintrp.ϕ.u))
        return (Int32(1), Float64(f.isco), tab)
    end
    nothing
end

"""
    _builtin_pf(pf, m) -> (GrPointFunction, keepalive) or nothing

Recognises `affine_time`, `redshift(m, x)` (Kerr: analytic; other metrics: `interpolate_redshift`) on their own or
composed with `filter_early_term` / `filter_intersected` through `∘` (point-functions.jl:107-120: the composite is a
`PointFunction` over a closure with fields `f1`, `f2`, `pf2`).  `shadow()` is `affine_time() ∘ filter_early_term()`.
"""
const _U_STATIC = (1.0, 0.0, 0.0, 0.0)      # has_u_src = 0: every image-plane caller measures against the static observer
function _builtin_pf(pf, m)
    pf isa PointFunction || return nothing
    f = pf.f
    base, filter_id, fill = f, Int32(0), NaN
    if hasproperty(f, :f1) && hasproperty(f, :f2) && hasproperty(f, :pf2)
        filter_id = f.f2 isa _EARLY_F ? Int32(1) : f.f2 isa _INTERSECTED_F ? Int32(2) : return nothing
        fill = Float64(f.pf2.default)
        base = f.f1
    end
    b = _builtin_base(base, m)
    isnothing(b) && return nothing
    pf_id, r_isco, tab = b
    if isnothing(tab)
        return (GrPointFunction(pf_id, filter_id, fill, r_isco, 0, C_NULL, C_NULL, C_NULL, C_NULL, Int32(0), Int32(0), _U_STATIC), nothing)
    end
    (GrPointFunction(pf_id, filter_id, fill, r_isco, length(tab[1]), pointer(tab[1]), pointer(tab[2]), pointer(tab[3]),
        pointer(tab[4]), Int32(0), Int32(0), _U_STATIC), tab)
end

# ---------------------------------------------------------------------------------------------------------------
# The generic boundary: same signature as src/tracing/tracing.jl:151-158.  GeodesicPoint{Float64,Nothing} is isbits
# with the layout of gr_point (152 bytes), so the result vector is filled in place by the library.
# ---------------------------------------------------------------------------------------------------------------
"The (αs, βs) an image plane's velocity closure captured, or `nothing` for any other velocity."
function _plane_rays(config::TracingConfiguration)
    f = config.velocity
    f isa Function || return nothing
    names = fieldnames(typeof(f))
    (:αs in names && :βs in names) || return nothing
    αs, βs = getfield(f, :αs), getfield(f, :βs)
    (αs isa AbstractArray{Float64} && βs isa AbstractArray{Float64}) || return nothing
    (length(αs) == config.trajectories == length(βs)) || return nothing
    collect(vec(αs)), collect(vec(βs))
end

# The result of ensemble_solve_tracing_problem -- the reference allocates `Vector{GeodesicPoint{T}}(undef, n)` itself
# (tracing/tracing.jl:179-183) -- in memory the LIBRARY page-locked (gr_host_alloc, ABI 5): 152 B per ray come back by DMA
# at the link's rate, band by band under the trace of the later bands, instead of through the pageable path (page faults,
# then ~30 GB/s).  Julia owns the wrapper; its finalizer hands the block back (gr_host_free does not need the context, so
# the order in which the ensemble and the array are collected does not matter).  Small results and a refused allocation
# (RLIMIT_MEMLOCK) take an ordinary Vector.
# The library builds the block on transparent huge pages and registers it (13 ms for 608 MiB) and keeps freed blocks in a small pool:
# a render whose previous result has been finalized by then (`finalize(prev)` in a loop that keeps only the latest result) gets
# its block for nothing.
const PINNED_RESULT_MIN_BYTES = 64 << 20
const PINNED_IMAGE_MIN_BYTES = 8 << 20

# Julia's GC sees the ~100-byte wrapper of such an array, not the hundreds of MiB of page-locked memory behind it, and a render
# loop that drops its results without `finalize` could pile those up long before a collection runs.  So:
#   * ENV["GRADUS_MI355X_PINNED_RESULTS"] = "0" turns pinned results off altogether (ordinary Vectors / Matrices, as the
#     reference allocates them; the Python binding has the same switch);
#   * the LIBRARY counts the bytes it has handed out and refuses a request that would take them past its cap
#     (gr_ctx_set "pinned_max_mib", default 8 GiB).  On a refusal the shim runs an incremental collection -- which finalizes
#     the dropped results and returns their blocks -- and asks once more; a second refusal means the caller really HOLDS that
#     much, and the result is an ordinary pageable array (slower copy home, nothing else changes);
#   * freed blocks wait in a pool of at most 1 GiB ("pinned_pool_mib") that is emptied when the last context is destroyed.
_pinned_results_enabled() = get(ENV, "GRADUS_MI355X_PINNED_RESULTS", "1") != "0"

function _pinned_block(ensemble::EnsembleMI355X, bytes::Int64)
    _pinned_results_enabled() || return C_NULL
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    for attempt = 1:2
        rc = ccall((:gr_host_alloc, LIB), Int32, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), ensemble.ctxs[1], bytes, ref)
        (rc == 0 && ref[] != C_NULL) && return ref[]
        attempt == 1 && GC.gc(false)          # finalizers of dropped results hand their blocks back
    end
    C_NULL
end

function _result_vector(ensemble::EnsembleMI355X, N::Integer)
    bytes = Int64(N) * 152
    if bytes >= PINNED_RESULT_MIN_BYTES
        p = _pinned_block(ensemble, bytes)
        if p != C_NULL
            out = unsafe_wrap(Array, Ptr{GeodesicPoint{Float64,Nothing}}(p), Int(N); own = false)
            finalizer(a -> ccall((:gr_host_free, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), C_NULL, pointer(a)), out)
            return out
        end
    end
    Vector{GeodesicPoint{Float64,Nothing}}(undef, N)
end

# the H x W image of a fused render: page-locked by the library from 8 MiB up, an ordinary Matrix below that or when refused
function _result_matrix(ensemble::EnsembleMI355X, H::Integer, W::Integer)
    bytes = Int64(H) * Int64(W) * 8
    if bytes >= PINNED_IMAGE_MIN_BYTES
        p = _pinned_block(ensemble, bytes)
        if p != C_NULL
            out = unsafe_wrap(Array, Ptr{Float64}(p), (Int(H), Int(W)); own = false)
            finalizer(a -> ccall((:gr_host_free, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), C_NULL, pointer(a)), out)
            return out
        end
    end
    zeros(Float64, (H, W))
end

function _cpu_fallback(reason, problem, config; kwargs...)
    @warn "EnsembleMI355X: $reason -- tracing on the CPU with EnsembleEndpointThreads instead"
    Gradus.ensemble_solve_tracing_problem(Gradus.EnsembleEndpointThreads(), problem, config; kwargs...)
end

function Gradus.ensemble_solve_tracing_problem(
    ensemble::EnsembleMI355X,
    problem::EnsembleProblem,
    config::TracingConfiguration{T};
    progress_bar = nothing,
    save_on = false,
    maxiters = 1_000_000,               # OrdinaryDiffEq's default for adaptive solvers
    solver_opts...,
) where {T}
    save_on && error("Cannot use `EnsembleMI355X` with `save_on`")       # as tracing.jl:159-161 for EnsembleEndpointThreads
    isnothing(progress_bar) || @warn "Progress meter is not supported by EnsembleMI355X." maxlog = 1
    N = config.trajectories
    local cfg_val, keep
    try
        T === Float64 || throw(UnsupportedOnDevice("number type $T; the boundary carries Float64"))
        isempty(solver_opts) || throw(UnsupportedOnDevice("solver options $(keys(solver_opts))"))
        cfg_val, keep = _config(config, _trace_of(problem); maxiters = maxiters)
    catch e
        e isa UnsupportedOnDevice || rethrow()
        return _cpu_fallback(e.msg, problem, config; progress_bar, save_on, maxiters, solver_opts...)
    end
    cfg = Ref(cfg_val)
    @assert sizeof(GeodesicPoint{Float64,Nothing}) == 152
    out = _result_vector(ensemble, N)
    stats = Ref{GrStats}()
    nctx = length(ensemble.ctxs)
    mstats = Vector{GrStats}(undef, nctx)           # the *_multi entry points report per context
    plane = _render_plane(config)
    if !isnothing(plane)
        # the pixel -> velocity closure of rendergeodesics / prerendergeodesics: rays are made on the device
        pl = Ref(plane)
        if nctx > 1 && plane.width % nctx == 0
            # columns dealt over the devices; into a pinned `out` every device's kernel stores its records at their place
            _check(GC.@preserve keep out ccall((:gr_render_endpoints_multi, LIB), Int32,
                (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrPlane}, Int64, Ptr{Cvoid}, Ptr{GrStats}),
                ensemble.ctxs, nctx, cfg, pl, 0, out, mstats))
            return out
        end
        rg = Ref(GrRange(0, N, max(N, 1), 1))
        _check(GC.@preserve keep out ccall((:gr_render_endpoints, LIB), Int32,
            (Ptr{Cvoid}, Ref{GrConfig}, Ref{GrPlane}, Ref{GrRange}, Ptr{Cvoid}, Ref{GrStats}),
            ensemble.ctxs[1], cfg, pl, rg, out, stats))
        return out
    end
    # the velocity closure of an image plane (promote_velfunc, image-planes/planes.jl:180-184: it captures the plane's impact
    # parameters αs, βs): 16 B per ray go over and map_impact_parameters runs on the device -- what lineprofile(…,
    # BinningMethod()) and lagtransfer trace (line-profiles.jl:171-183)
    pr = _plane_rays(config)
    if !isnothing(pr) && config.position isa SVector
        αv, βv = pr
        g = Gradus.metric(config.metric, config.position)
        Mx = inv(g) * hcat(Gradus.lnrbasis(g)...)
        rays = Ref(GrRayset(Tuple(SVector{4,Float64}(config.position)), Tuple(permutedims(Mx)), pointer(αv), pointer(βv),
            Ptr{Float64}(C_NULL), N, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0, 0,
            Int32(0), Int32(0), 0, 0, 0, Int32(0), Int32(0), Int32(0), Int32(0), 0.0, Ptr{Float64}(C_NULL), 0, 0, Ptr{Float64}(C_NULL)))
        if nctx > 1
            _check(GC.@preserve keep αv βv out ccall((:gr_rayset_endpoints_multi, LIB), Int32,
                (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrRayset}, Ptr{Cvoid}, Ptr{GrStats}),
                ensemble.ctxs, nctx, cfg, rays, out, mstats))
            return out
        end
        _check(GC.@preserve keep αv βv out ccall((:gr_rayset_endpoints, LIB), Int32,
            (Ptr{Cvoid}, Ref{GrConfig}, Ref{GrRayset}, Ptr{Cvoid}, Ref{GrStats}),
            ensemble.ctxs[1], cfg, rays, out, stats))
        return out
    end
    # arbitrary velocity closures are evaluated on the host (UNCONSTRAINED: constrain_all runs on the device with
    # the trace's μ); (xs, vs) arrays are passed through
    xs = config.position isa SVector ? [SVector{4,Float64}(config.position)] : Vector{SVector{4,Float64}}(config.position)
    vs = config.velocity isa Function ? SVector{4,Float64}[config.velocity(i) for i = 1:N] :
         Vector{SVector{4,Float64}}(config.velocity)
    if nctx > 1
        _check(GC.@preserve keep xs vs out ccall((:gr_trace_endpoints_multi, LIB), Int32,
            (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Cvoid}, Ptr{GrStats}),
            ensemble.ctxs, nctx, cfg, reinterpret(Float64, xs), length(xs) == 1 ? 0 : 4, reinterpret(Float64, vs), N, out, mstats))
        return out
    end
    _check(GC.@preserve keep xs vs out ccall((:gr_trace_endpoints, LIB), Int32,
        (Ptr{Cvoid}, Ref{GrConfig}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Cvoid}, Ref{GrStats}),
        ensemble.ctxs[1], cfg, reinterpret(Float64, xs), length(xs) == 1 ? 0 : 4, reinterpret(Float64, vs), N, out, stats))
    out
end

# ---------------------------------------------------------------------------------------------------------------
# The fused path of rendergeodesics: render_into_image! (rendering.jl:89-101) specialised on the ensemble type, which
# TracingConfiguration carries as its 9th type parameter (configuration.jl:3-16,57).
# ---------------------------------------------------------------------------------------------------------------
function Gradus.render_into_image!(
    image,
    trace::AbstractTrace,
    config::TracingConfiguration{T,<:Any,<:Any,<:Any,<:Any,<:Any,<:Any,<:Any,<:EnsembleMI355X};
    pf = ConstPointFunctions.shadow(T),       # == the reference's default: affine_time ∘ (λ_max < max_time), NaN
    solver_opts...,
) where {T}
    ens = config.ensemble
    fused = nothing
    # what reaches here besides `pf`: `save_on = false` (render_configuration, rendering.jl:20) and `verbose`
    plain = all(k -> k === :verbose || (k === :save_on && solver_opts[k] == false), keys(solver_opts))
    if T === Float64 && image isa Matrix{Float64} && plain
        try
            plane = _render_plane(config)
            bpf = _builtin_pf(pf, config.metric)
            if !isnothing(plane) && !isnothing(bpf) && size(image) == (plane.height, plane.width)
                cfg_val, keep = _config(config, trace)
                fused = (cfg_val, keep, plane, bpf)
            end
        catch e
            e isa UnsupportedOnDevice || rethrow()
        end
    end
    if isnothing(fused)
        # user-defined point functions, Float32 configurations, ...: device-traced end points (the generic method
        # above) + Gradus' own apply_to_image!
        return invoke(Gradus.render_into_image!, Tuple{Any,AbstractTrace,TracingConfiguration}, image, trace, config;
            pf = pf, solver_opts...)
    end
    cfg_val, keep, plane, (gpf, keep_pf) = fused
    cfg, pl, pfr = Ref(cfg_val), Ref(plane), Ref(gpf)
    stats = Vector{GrStats}(undef, length(ens.ctxs))
    # image is the column-major H x W matrix of rendering.jl:50: ray i = (x-1) H + y is image[i], the order the
    # library writes; block_cols = 0 lets it pick the deal of columns over the ensemble's devices
    _check(GC.@preserve keep keep_pf image ccall((:gr_render_multi, LIB), Int32,
        (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrPlane}, Ref{GrPointFunction}, Int64, Ptr{Float64}, Ptr{GrStats}),
        ens.ctxs, length(ens.ctxs), cfg, pl, pfr, 0, image, stats))
    image
end

# ---------------------------------------------------------------------------------------------------------------
# The same fused call without going through Gradus' configuration machinery (scripts, benchmarks)
# ---------------------------------------------------------------------------------------------------------------
function render_mi355x(ensemble::EnsembleMI355X, m, x::SVector{4,Float64}, d, λmax; image_width, image_height,
        αlims, βlims, pf = ConstPointFunctions.shadow(), gtol = 1e-2, abstol = 1e-9, reltol = 1e-9,
        chart = Gradus.chart_for_metric(m), μ = 0.0, q = 0.0)
    id, params = _metric(m)
    did, rin, rout, dparams = _disc(d)
    r_in, r_out, tab, θ0, θ1 = _chart(chart)
    dtab = _disc_table(d)
    mtab = _metric_table(m, r_in, r_out)
    cfg = Ref(GrConfig(id, did, params, r_in, r_out, rin, rout, gtol, 0.0, Float64(λmax),
        abstol, reltol, Float64(μ), 1_000_000, Int32(0), Int32(0), 1e-4, dparams,
        isempty(dtab) ? Ptr{Float64}(C_NULL) : pointer(dtab), _disc_table_n(d, dtab),
        isempty(tab) ? Ptr{Float64}(C_NULL) : pointer(tab), length(tab), θ0, θ1, Float64(q), Int32(0), Int32(0), π / 2,
        _components(d)[1], Int32(0), _components(d)[2],
        isempty(mtab) ? Ptr{Float64}(C_NULL) : pointer(mtab), length(mtab)))
    g = Gradus.metric(m, x)
    Mx = inv(g) * hcat(Gradus.lnrbasis(g)...)                       # tracing/utility.jl:32-40
    plane = Ref(GrPlane(Tuple(x), Tuple(permutedims(Mx)), Float64(αlims[1]), Float64(αlims[2]), Float64(βlims[1]),
        Float64(βlims[2]), image_width, image_height, 1e-6))
    bpf = _builtin_pf(pf, m)
    isnothing(bpf) && error("render_mi355x: `pf` is not one of the built-in point functions the kernels evaluate")
    gpf, keep_pf = bpf
    pfs = Ref(gpf)
    stats = Vector{GrStats}(undef, length(ensemble.ctxs))
    if length(ensemble.ctxs) == 1
        # one device: the image lives in a block the library page-locked (from 8 MiB up; pooled, see _result_vector) and the
        # kernel stores the pixels across the link itself -- no staging image in HBM, no copy (2048²: 18.4 against 21.6 ms per call)
        image = _result_matrix(ensemble, image_height, image_width)
        N = image_height * image_width
        rg = Ref(GrRange(0, N, max(N, 1), 1))
        _check(GC.@preserve tab dtab mtab keep_pf image ccall((:gr_render, LIB), Int32,
            (Ptr{Cvoid}, Ref{GrConfig}, Ref{GrPlane}, Ref{GrPointFunction}, Ref{GrRange}, Ptr{Float64}, Ptr{GrStats}),
            ensemble.ctxs[1], cfg, plane, pfs, rg, image, stats))
    else
        # several devices: the same pinned block, every device's kernel stores its columns at their place (no copies at all)
        image = _result_matrix(ensemble, image_height, image_width)
        _check(GC.@preserve tab dtab mtab keep_pf image ccall((:gr_render_multi, LIB), Int32,
            (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrPlane}, Ref{GrPointFunction}, Int64, Ptr{Float64}, Ptr{GrStats}),
            ensemble.ctxs, length(ensemble.ctxs), cfg, plane, pfs, 0, image, stats))
    end
    α, β = Gradus.impact_axes(image_width, image_height, αlims, βlims)
    α, β, image
end

# ---------------------------------------------------------------------------------------------------------------
# lineprofile(bins, ε, m, u, d, BinningMethod(); plane) (src/line-profiles.jl:152-198) fused on the device for a power-law
# emissivity ε(r) = r^-q or an emissivity profile (RadialDiscProfile) on a PolarPlane: the plane crosses the boundary as its radii and the cosines / sines of its
# angles (src/image-planes/planes.jl:96-131), the device forms the rays, traces them, evaluates redshift and ε g³ area
# and bins -- BASELINE config 5 (4096² rays) in one call.  The reference's method fixes `ensemble =
# EnsembleEndpointThreads()` ahead of `solver_args...`, so a maintainer wiring this in adds one branch there:
#     ensemble isa EnsembleMI355X && ε isa PowerLaw... && plane isa PolarPlane && return lineprofile_mi355x(...)
# ---------------------------------------------------------------------------------------------------------------
function lineprofile_mi355x(ensemble::EnsembleMI355X, bins::AbstractVector{Float64}, ε::Union{Real,Gradus.RadialDiscProfile}, m,
        u::SVector{4,Float64}, d;
        plane::Gradus.PolarPlane = Gradus.PolarPlane(Gradus.GeometricGrid(); Nr = 450, Nθ = 1300, r_max = 250.0),
        λ_max = 2 * u[2], minrₑ = Gradus.isco(m), maxrₑ = 50.0, redshift_pf = ConstPointFunctions.redshift(m, u),
        gtol = 1e-2, abstol = 1e-9, reltol = 1e-9, chart = Gradus.chart_for_metric(m), upper_hemisphere = true)
    id, params = _metric(m)
    did, rin, rout, dparams = _disc(d)
    r_in, r_out, tab, θ0, θ1 = _chart(chart)
    dtab = _disc_table(d)
    mtab = _metric_table(m, r_in, r_out)
    cfg = Ref(GrConfig(id, did, params, r_in, r_out, rin, rout, gtol, 0.0, Float64(λ_max),
        abstol, reltol, 0.0, 1_000_000, Int32(upper_hemisphere), Int32(0), 1e-4, dparams,
        isempty(dtab) ? Ptr{Float64}(C_NULL) : pointer(dtab), _disc_table_n(d, dtab),
        isempty(tab) ? Ptr{Float64}(C_NULL) : pointer(tab), length(tab), θ0, θ1, 0.0, Int32(0), Int32(0), π / 2,
        _components(d)[1], Int32(0), _components(d)[2],
        isempty(mtab) ? Ptr{Float64}(C_NULL) : pointer(mtab), length(mtab)))
    g = Gradus.metric(m, u)
    Mx = inv(g) * hcat(Gradus.lnrbasis(g)...)
    rs = collect(Float64, plane.grid(plane.r_min, plane.r_max, plane.Nr))             # planes.jl:110-114
    δθ = (plane.θ_max - plane.θ_min) / plane.Nθ
    θs = collect(range(plane.θ_min, plane.θ_max - δθ, plane.Nθ))
    cs, sn = cos.(θs), sin.(θs)
    rays = Ref(GrRayset(Tuple(u), Tuple(permutedims(Mx)), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL),
        plane.Nr * plane.Nθ, Ptr{Float64}(C_NULL), pointer(rs), pointer(cs), pointer(sn), plane.Nr, plane.Nθ, Int32(1), Int32(0),
        0, 0, 0, Int32(0), Int32(0), Int32(0), Int32(0), 0.0, Ptr{Float64}(C_NULL), 0, 0, Ptr{Float64}(C_NULL)))
    bpf = _builtin_pf(redshift_pf, m)
    isnothing(bpf) && error("lineprofile_mi355x: `redshift_pf` is not a redshift point function the kernels evaluate")
    gpf, keep_pf = bpf
    pfs = Ref(gpf)
    edges = collect(Float64, bins)
    # ε: the exponent q of a power law r^-q, or an emissivity profile whose table the device interpolates like
    # emissivity_at(prof, r) does (src/corona/radial.jl:15-18)
    q = ε isa Real ? Float64(ε) : 0.0
    er = ε isa Real ? Float64[] : collect(Float64, ε.radii)
    ev = ε isa Real ? Float64[] : collect(Float64, ε.ε)
    binning = Ref(GrBinning(Float64(minrₑ), Float64(maxrₑ), q, length(edges), pointer(edges),
        isempty(er) ? Ptr{Float64}(C_NULL) : pointer(er), isempty(ev) ? Ptr{Float64}(C_NULL) : pointer(ev), length(er)))
    flux = zeros(Float64, length(edges))
    if length(ensemble.ctxs) > 1
        # the plane's rays dealt over the devices, one histogram per device, added by the library on the host
        mstats = Vector{GrStats}(undef, length(ensemble.ctxs))
        _check(GC.@preserve tab dtab mtab keep_pf rs cs sn edges er ev ccall((:gr_lineprofile_multi, LIB), Int32,
            (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrRayset}, Ref{GrPointFunction}, Ref{GrBinning}, Ptr{Float64}, Ptr{GrStats}),
            ensemble.ctxs, length(ensemble.ctxs), cfg, rays, pfs, binning, flux, mstats))
        return bins, flux ./ sum(flux)                                                     # line-profiles.jl:197
    end
    stats = Ref{GrStats}()
    _check(GC.@preserve tab dtab mtab keep_pf rs cs sn edges er ev ccall((:gr_lineprofile, LIB), Int32,
        (Ptr{Cvoid}, Ref{GrConfig}, Ref{GrRayset}, Ref{GrPointFunction}, Ref{GrBinning}, Ptr{Float64}, Ref{GrStats}),
        ensemble.ctxs[1], cfg, rays, pfs, binning, flux, stats))
    bins, flux ./ sum(flux)                                                            # line-profiles.jl:197
end

# ---------------------------------------------------------------------------------------------------------------
# emissivity_profile(m, d, model, spectrum; n_samples, sampler, N, grid) (src/corona/emissivity.jl:118-168) for a source at
# one position, with the per-ray half on the device: the sky directions of samplers.jl:30-99 are formed per lane from the
# sample number (the tetrad and the Cartesian -> spherical Jacobian cross as one 4 x 4 matrix), the rays are traced against
# the disc, energy_ratio (flux-calculations.jl:96-110) is taken against the source's and the disc's four-velocity, and
# bucket!(IndexBucket, Simple(), radii, bins) with the per-bin means of radial.jl:60-84 is a histogram in LDS.  Back come
# (min ρ, max ρ, hits) and 3 N doubles; radial.jl:86-100 (N evaluations) runs here.  A maintainer wiring this in adds one
# branch to emissivity_profile(setup, m, d, model): `ensemble isa EnsembleMI355X && return emissivity_profile_mi355x(...)`.
# ---------------------------------------------------------------------------------------------------------------
_sky_sampler(s::Gradus.EvenSampler) = (Int32(1), 0.0)
_sky_sampler(s::Gradus.WeierstrassSampler) = (Int32(2), Float64(s.resolution))
_sky_domain(::Gradus.AbstractDirectionSampler{Gradus.LowerHemisphere}) = Int32(0)
_sky_domain(::Gradus.AbstractDirectionSampler{Gradus.BothHemispheres}) = Int32(1)
_sky_generator(::Gradus.AbstractDirectionSampler{D,Gradus.GoldenSpiralGenerator}, N) where {D} = (Int32(0), Float64[])
_sky_generator(::Gradus.AbstractDirectionSampler{D,Gradus.EvenGenerator}, N) where {D} = (Int32(1), Float64[])
_sky_generator(s::Gradus.AbstractDirectionSampler, N) = (Int32(2), Float64[Gradus.geti(s, i, N) for i = 1:N])   # RandomGenerator: rand() N

function emissivity_profile_mi355x(ensemble::EnsembleMI355X, m::Gradus.AbstractStaticAxisSymmetric, d, model,
        spectrum = Gradus.PowerLawSpectrum(2); n_samples = 1000,
        sampler = Gradus.EvenSampler(Gradus.BothHemispheres(), Gradus.GoldenSpiralGenerator()), λmax = 10_000.0,
        grid = Gradus.GeometricGrid(), N = 100, gtol = 1e-2, abstol = 1e-9, reltol = 1e-9, chart = Gradus.chart_for_metric(m),
        upper_hemisphere = true)
    rmin = Gradus.inner_radius(m) * 1.9
    function matrix_at(x, v)                                                               # v = T (1, J k̂), samplers.jl:81-99
        B = zeros(Float64, 4, 4)
        B[1, 1] = 1.0
        B[2:4, 2:4] .= Gradus._cart_to_spher_jacobian(x[3], x[4])
        Matrix{Float64}(Gradus.tetradframe_matrix(m, x, v)) * B
    end
    one_position = model isa Union{Gradus.LampPostModel,Gradus.BeamedPointSource,Gradus.RingCorona}
    sky_rows = Float64[]
    if one_position
        x, v_src = Gradus.sample_position_velocity(m, model)
        x[2] < rmin && error("source position lies inside 1.9 inner radii")
        x = SVector{4,Float64}(x[1], x[2], clamp(x[3], 1e-3, π - 1e-3), x[4])              # corona-models.jl:18-24
        Mx = matrix_at(x, v_src)
    else
        # a source without one position (DiscCorona, extended.jl:165-183; any model whose sample_position_velocity draws): 28 doubles
        # per sample -- position, matrix, the source velocity with its index lowered, g_tμ -- in the order corona-models.jl:1-33 draws
        sky_rows = Vector{Float64}(undef, 28 * n_samples)
        x = SVector{4,Float64}(0.0, 0.0, 0.0, 0.0)
        for k = 1:n_samples
            xk, vk = Gradus.sample_position_velocity(m, model)
            while xk[2] < rmin
                xk, vk = Gradus.sample_position_velocity(m, model)
            end
            xk = SVector{4,Float64}(xk[1], xk[2], clamp(xk[3], 1e-3, π - 1e-3), xk[4])
            g = Gradus.metric_components(m, SVector(xk[2], xk[3]))
            o = 28 * (k - 1)
            sky_rows[o+1:o+4] .= xk
            sky_rows[o+5:o+20] .= vec(permutedims(matrix_at(xk, vk)))                      # row-major
            sky_rows[o+21:o+24] .= (g[1] * vk[1] + g[5] * vk[4], g[2] * vk[2], g[3] * vk[3], g[4] * vk[4] + g[5] * vk[1])
            sky_rows[o+25:o+28] .= (g[1], 0.0, 0.0, g[5])
            xk[2] > x[2] && (x = xk)
        end
        v_src = SVector{4,Float64}(1.0, 0.0, 0.0, 0.0)                                     # (not read: has_u_src = 0 below)
        Mx = Float64[i == j ? 1.0 : 0.0 for i = 1:4, j = 1:4]                               # (not read either: every row brings its own)
    end
    sid, res = _sky_sampler(sampler)
    gid, sky_i = _sky_generator(sampler, n_samples)
    rays = Ref(GrRayset(Tuple(x), Tuple(permutedims(Mx)), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL),
        n_samples, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0, 0, Int32(0), Int32(0),
        0, 0, 0, sid, _sky_domain(sampler), gid, Int32(0), res, isempty(sky_i) ? Ptr{Float64}(C_NULL) : pointer(sky_i), 0, 0, isempty(sky_rows) ? Ptr{Float64}(C_NULL) : pointer(sky_rows)))
    # the disc velocity of _keplerian_velocity_projector (circular-orbits.jl:155-170): Keplerian outside the ISCO, the traced
    # plunging table inside (three NaNLinearInterpolators over the same radii, orbit-solving.jl:99-131)
    pintrp = _expect_fields(Gradus.interpolate_plunging_velocities(m), :m, :t, :r, :ϕ)
    ptab = (collect(Float64, pintrp.t.t), collect(Float64, pintrp.t.u), collect(Float64, pintrp.r.u), collect(Float64, pintrp.ϕ.u))
    pfs = Ref(GrPointFunction(Int32(1), Int32(0), NaN, Float64(Gradus.isco(m)), length(ptab[1]), pointer(ptab[1]), pointer(ptab[2]),
        pointer(ptab[3]), pointer(ptab[4]), Int32(one_position ? 1 : 0), Int32(0), Tuple(SVector{4,Float64}(v_src))))
    id, params = _metric(m)
    did, rin, rout, dparams = _disc(d)
    r_in, r_out, tab, θ0, θ1 = _chart(chart)
    dtab = _disc_table(d)
    mtab = _metric_table(m, r_in, r_out)
    cfg = Ref(GrConfig(id, did, params, r_in, r_out, rin, rout, gtol, 0.0, Float64(λmax),
        abstol, reltol, 0.0, 1_000_000, Int32(upper_hemisphere), Int32(0), 1e-4, dparams,
        isempty(dtab) ? Ptr{Float64}(C_NULL) : pointer(dtab), _disc_table_n(d, dtab),
        isempty(tab) ? Ptr{Float64}(C_NULL) : pointer(tab), length(tab), θ0, θ1, 0.0, Int32(0), Int32(0), π / 2,
        _components(d)[1], Int32(0), _components(d)[2],
        isempty(mtab) ? Ptr{Float64}(C_NULL) : pointer(mtab), length(mtab)))
    lim = zeros(Float64, 2)
    hits = Ref{Int64}(0)
    stats = Ref{GrStats}()
    ctxs = ensemble.ctxs
    nctx = length(ctxs)
    if nctx > 1      # the samples shard like any ray set (ABI 8): every context traces and bins its share, the integer sums add up exactly
        cstats = Vector{GrStats}(undef, nctx)
        _check(GC.@preserve tab dtab mtab ptab sky_i sky_rows ccall((:gr_corona_trace_multi, LIB), Int32,
            (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrRayset}, Ref{GrPointFunction}, Ptr{Float64}, Ref{Int64}, Ptr{GrStats}),
            ctxs, nctx, cfg, rays, pfs, lim, hits, cstats))
    else
        _check(GC.@preserve tab dtab mtab ptab sky_i sky_rows ccall((:gr_corona_trace, LIB), Int32,
            (Ptr{Cvoid}, Ref{GrConfig}, Ref{GrRayset}, Ref{GrPointFunction}, Ptr{Float64}, Ref{Int64}, Ref{GrStats}),
            ctxs[1], cfg, rays, pfs, lim, hits, stats))
    end
    hits[] > 0 || error("no ray of the corona reached the disc")
    bins = collect(Float64, grid(lim[1], lim[2], N))                                       # radial.jl:60
    sums = zeros(Float64, length(bins), 3)                                                 # columns: count, Σ g, Σ t
    if nctx > 1
        _check(ccall((:gr_corona_bin_multi, LIB), Int32, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, Int64, Ptr{Float64}), ctxs, nctx, bins, length(bins), sums))
    else
        _check(ccall((:gr_corona_bin, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), ctxs[1], bins, length(bins), sums))
    end
    count, gs, ts = sums[:, 1], sums[:, 2] ./ sums[:, 1], sums[:, 3] ./ sums[:, 1]         # the means of radial.jl:70-84,97
    g_interp = Gradus._make_interpolation(bins, gs)
    disc_velocity = Gradus._keplerian_velocity_projector(m)
    ε = similar(bins)
    for i in eachindex(bins)                                                               # radial.jl:86-95
        R = bins[i]
        dr = R - (i == 1 ? 0.0 : bins[i-1])
        xb = SVector(0.0, R, π / 2, 0.0)
        ε[i] = Gradus.source_to_disc_emissivity(m, spectrum, count[i], dr * Gradus._proper_area(m, xb), xb, g_interp(R), disc_velocity(xb))
    end
    Gradus.RadialDiscProfile(bins, ts, ε)
end

# ---------------------------------------------------------------------------------------------------------------
# selftest: the device against the reference's own CPU ensemble, and a user-defined metric through the table
# ---------------------------------------------------------------------------------------------------------------
"Kerr in disguise: a metric this file has no `_metric` method for, defined the way a user defines one (kerr-metric.jl:62-72)."
struct _SelftestMetric{T} <: Gradus.AbstractStaticAxisSymmetric{T}
    M::T
    a::T
end
Gradus.metric_components(m::_SelftestMetric, rθ) = Gradus.metric_components(KerrMetric(m.M, m.a), rθ)
Gradus.inner_radius(m::_SelftestMetric) = Gradus.inner_radius(KerrMetric(m.M, m.a))

"""
    selftest(; device = 0, verbose = true) -> Bool

Run after installing or upgrading Gradus.jl or the library.  Renders the 20 x 20 scene of the reference's smoke tests
(test/smoke-tests/rendergeodesics.jl: observer at r = 100, θ = 85°, ThinDisc(0, 40), λ_max = 200) three ways and compares end
points ray by ray:
  1. `KerrMetric` on `EnsembleMI355X` against `EnsembleEndpointThreads` (status equal on all but disc-rim rays, positions to 1e-6);
  2. the same through `rendergeodesics` with the redshift point function (the fused path against Gradus' `apply_to_image!`);
  3. a user-defined metric (a struct with only `metric_components` and `inner_radius`) on the device -- through the table -- against
     `KerrMetric` on the CPU, and the table against `Gradus.metric_jacobian` at a point;
  4. a lamp-post emissivity profile through `emissivity_profile_mi355x` against `Gradus.emissivity_profile` on the same samples.
Throws with a description of the first disagreement; returns `true` otherwise.
"""
function selftest(; device = 0, verbose = true)
    say(msg) = verbose && @info "GradusMI355X.selftest: $msg"
    ens = EnsembleMI355X([device])
    m = KerrMetric(1.0, 0.998)
    x = SVector(0.0, 100.0, deg2rad(85), 0.0)
    d = ThinDisc(0.0, 40.0)
    kw = (; image_width = 20, image_height = 20, αlims = (-9.5, 9.5), βlims = (-9.5, 9.5), verbose = false)
    function endpoints(metric, ensemble)
        _, _, cache = Gradus.prerendergeodesics(metric, x, d, 200.0; ensemble = ensemble, kw...)
        vec(cache.points)
    end
    function compare(a, b, what; flips = 4)
        nf = count(i -> a[i].status != b[i].status, eachindex(a))
        nf <= flips || error("selftest ($what): $nf rays differ in status")
        for i in eachindex(a)
            (a[i].status == b[i].status && a[i].status != StatusCodes.WithinInnerBoundary) || continue
            err = maximum(abs.(a[i].x .- b[i].x) ./ max.(abs.(b[i].x), 1e-3 * maximum(abs.(b[i].x))))
            err < 1e-6 || error("selftest ($what): ray $i ends at $(a[i].x), the CPU ensemble at $(b[i].x)")
        end
        say("$what: $(length(a)) rays agree ($nf rim flips)")
    end
    cpu = endpoints(m, Gradus.EnsembleEndpointThreads())
    compare(endpoints(m, ens), cpu, "KerrMetric, end points")
    pf = ConstPointFunctions.redshift(m, x) ∘ ConstPointFunctions.filter_intersected()
    _, _, img_d = Gradus.rendergeodesics(m, x, d, 200.0; pf = pf, ensemble = ens, kw...)
    _, _, img_c = Gradus.rendergeodesics(m, x, d, 200.0; pf = pf, ensemble = Gradus.EnsembleEndpointThreads(), kw...)
    both = .!isnan.(img_d) .& .!isnan.(img_c)
    (count(isnan.(img_d) .!= isnan.(img_c)) <= 4 && maximum(abs.(img_d[both] ./ img_c[both] .- 1)) < 1e-6) ||
        error("selftest (fused redshift render): the images differ")
    say("fused redshift render agrees on $(count(both)) pixels")
    um = _SelftestMetric(1.0, 0.998)
    chart = Gradus.chart_for_metric(um)
    tab = metric_table(um, Float64(chart.inner_radius), Float64(chart.outer_radius))
    g, dr, dth = zeros(5), zeros(5), zeros(5)
    _check(ccall((:gr_metric_table_eval, LIB), Int32, (Ptr{Float64}, Int64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        tab, length(tab), 6.0, 1.2, g, dr, dth))
    gc, jac = Gradus.metric_jacobian(m, SVector(6.0, 1.2))            # auto-diff.jl:206-211: components and their (∂r, ∂θ)
    (maximum(abs.(g .- gc) ./ abs.(gc)) < 1e-10 && maximum(abs.(dr .- jac[:, 1])) < 1e-8 && maximum(abs.(dth .- jac[:, 2])) < 1e-8) ||
        error("selftest (tabulated metric): the table disagrees with Gradus.metric_jacobian at (6, 1.2)")
    compare(endpoints(um, ens), cpu, "user-defined metric through the table")
    # 4. the corona route: golden-spiral sampling is deterministic, so the device's profile and Gradus' own (CPU threads, the same
    #    samples) must agree bin by bin except where one photon sits on a bin edge
    model, dc = LampPostModel(h = 10.0), ThinDisc(0.0, 400.0)
    sampler = Gradus.EvenSampler(Gradus.BothHemispheres(), Gradus.GoldenSpiralGenerator())
    p_dev = emissivity_profile_mi355x(ens, m, dc, model; n_samples = 2000, sampler = sampler, N = 20)
    p_cpu = Gradus.emissivity_profile(m, dc, model; n_samples = 2000, sampler = sampler, N = 20)
    inner = 1:(length(p_cpu.radii)-2)                       # (the outermost edges hang on the last bit of the largest radius)
    okb = [i for i in inner if isfinite(p_cpu.ε[i]) && isfinite(p_dev.ε[i])]
    (maximum(abs.(p_dev.radii ./ p_cpu.radii .- 1)) < 1e-6 && length(okb) >= 10 &&
     count(i -> abs(p_dev.ε[i] / p_cpu.ε[i] - 1) > 1e-5, okb) <= 2) ||
        error("selftest (corona): the device's emissivity profile differs from Gradus.emissivity_profile")
    say("corona: emissivity profile agrees on $(length(okb)) bins")
    true
end

end # module

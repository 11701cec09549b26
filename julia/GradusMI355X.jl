# GradusMI355X.jl -- the reference-side binding a Gradus.jl maintainer would add.
#
# Selects the MI355X backend by dispatch on a new ensemble type at Gradus.jl's existing
# boundary, exactly as ext/GradusDiffEqGPUExt/GradusDiffEqGPUExt.jl:10-31 does for DiffEqGPU:
#
#     using Gradus, GradusMI355X
#     α, β, img = rendergeodesics(m, x, d, 2000.0; pf = pf, ensemble = EnsembleMI355X())
#
# All logic lives behind the C ABI (include/gradus_mi355x.h); this file only flattens Julia
# structs into the POD structs and ccall's.  It cannot be exercised in the build container
# (no Julia there); the same ABI is exercised from Python by the test-suite.
module GradusMI355X

using Gradus
using Gradus: TracingConfiguration, EnsembleProblem, GeodesicPoint, StatusCodes, AbstractTrace,
    KerrMetric, JohannsenMetric, ThinDisc, PolarChart, lnr_momentum_to_global_velocity_transform
using StaticArrays

export EnsembleMI355X, SampledThickDisc, render_mi355x, winding_numbers

const LIB = get(ENV, "GRADUS_MI355X_LIB", "libgradus_mi355x.so")

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
end

struct GrStats                       # == gr_stats
    rays::Int64
    accepted_steps::Int64
    rejected_steps::Int64
    rhs_evals::Int64
    flagged_rays::Int64
    status_count::NTuple{4,Int64}
    kernel_ms::Float64
end

struct GrPlane                       # == gr_plane
    x_obs::NTuple{4,Float64}
    Mx::NTuple{16,Float64}           # row-major ginv * hcat(lnrbasis(g)...)
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
end

_check(rc) = rc == 0 || error(unsafe_string(ccall((:gr_last_error, LIB), Cstring, ())))

"""
    EnsembleMI355X(devices = [0])

One `gr_ctx` per listed HIP device, all driven from the calling Julia task.  With several devices
`rendergeodesics` deals the image's columns to them (`gr_render_multi`).
"""
mutable struct EnsembleMI355X
    devices::Vector{Int32}
    ctxs::Vector{Ptr{Cvoid}}
    function EnsembleMI355X(devices = [0])
        abi = ccall((:gr_abi_version, LIB), Int32, ())
        abi == 3 || error("GradusMI355X: libgradus_mi355x.so has ABI version $abi, this binding is written for 3")
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
_metric(m) = error("GradusMI355X: metric $(typeof(m)) has no device implementation; use a CPU ensemble")

# (disc_id, disc_r_in, disc_r_out, disc_params)
_disc(::Nothing) = (Int32(0), 0.0, 0.0, (0.0, 0.0, 0.0, 0.0))
_disc(d::ThinDisc) = (Int32(1), Float64(d.inner_radius), Float64(d.outer_radius), (0.0, 0.0, 0.0, 0.0))
_disc(d::ShakuraSunyaev) = (Int32(2), Float64(d.inner_radius), Inf, (Float64(d.Ṁ_Ṁedd), Float64(d.inv_η), 0.0, 0.0))
_disc(d::DatumPlane) = (Int32(4), 0.0, 0.0, (Float64(d.height), 0.0, 0.0, 0.0))
_disc(d::EllipticalDisc) = (Int32(5), Float64(d.inner_radius), Inf, (Float64(d.semi_major), Float64(d.semi_minor), 0.0, 0.0))
_disc(d::PrecessingDisc{T,<:ThinDisc}) where {T} =
    (Int32(6), Float64(d.disc.inner_radius), Float64(d.disc.outer_radius), (Float64(d.β), Float64(d.γ), cos(d.β), sin(d.β)))
_disc(d) = error("GradusMI355X: geometry $(typeof(d)) has no device implementation; use a CPU ensemble")

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

# chart -> (r_inner, r_outer, table, θ_first, θ_last).  A PoloidalShapeChart built by
# event_horizon_chart wraps LinearInterpolation(r_min(θ_k), θ_k) on a uniform θ range (charts.jl:61-70).
_chart(c::PolarChart) = (Float64(c.inner_radius), Float64(c.outer_radius), Float64[], 0.0, 0.0)
function _chart(c::Gradus.PoloidalShapeChart)
    θ = collect(Float64, c.shapefunc.t)
    all(isapprox.(diff(θ), θ[2] - θ[1]; rtol = 1e-9)) || error("GradusMI355X: the chart's θ grid must be uniform")
    tab = collect(Float64, c.shapefunc.u)
    (minimum(filter(!isnan, tab)), Float64(c.outer_radius), tab, θ[1], θ[end])
end

# Returns the config and the arrays it points into (keep them alive for the duration of the call).
function _config(config::TracingConfiguration, trace::AbstractTrace; gtol = 1e-2, maxiters = 1_000_000,
        upper_hemisphere = nothing)
    id, params = _metric(config.metric)
    did, rin, rout, dparams = _disc(config.geometry)
    r_in, r_out, tab, θ0, θ1 = _chart(config.chart)
    dtab = _disc_table(config.geometry)
    # SciML callbacks are opaque closures and cannot cross the ABI.  `domain_upper_hemisphere(δ)`
    # (callbacks.jl:31-40) is implemented on the device: request it with the solver option
    # `upper_hemisphere = δ` instead of `callback = domain_upper_hemisphere(δ)`; any other callback is refused.
    isnothing(config.callback) || isnothing(upper_hemisphere) == false ||
        error("GradusMI355X: callbacks cannot run on the device (use `upper_hemisphere = δ` for domain_upper_hemisphere)")
    hemi = isnothing(upper_hemisphere) ? Int32(0) : Int32(1)
    δ = isnothing(upper_hemisphere) ? 1e-4 : Float64(upper_hemisphere)
    cfg = GrConfig(id, did, params, r_in, r_out, rin, rout, gtol,
        config.λ_domain[1], config.λ_domain[2], config.abstol, config.reltol, Float64(trace.μ),
        maxiters, hemi, Int32(0), δ, dparams,
        isempty(dtab) ? Ptr{Float64}(C_NULL) : pointer(dtab), length(dtab),
        isempty(tab) ? Ptr{Float64}(C_NULL) : pointer(tab), length(tab), θ0, θ1,
        trace isa Gradus.TraceWindings ? 0.0 : Float64(trace.q),
        trace isa Gradus.TraceWindings ? Int32(1) : Int32(0), Int32(0),
        trace isa Gradus.TraceWindings ? Float64(trace.plane_inc) : π / 2)
    cfg, (tab, dtab)
end

"""
    winding_numbers(points)

`gp.aux.winding` of end points traced with `trace = TraceWindings(...)` on `EnsembleMI355X`: the library returns
`GeodesicPoint{Float64,Nothing}` records and carries the count in bits 16..31 of the four padding bytes after `status`.
"""
winding_numbers(points::Vector{<:GeodesicPoint}) = GC.@preserve points [
    Int(unsafe_load(Ptr{UInt32}(pointer(points, i)) + 4) >> 16) for i in eachindex(points)
]

# The drop-in method: same signature as src/tracing/tracing.jl:151-158.  GeodesicPoint{Float64,
# Nothing} is isbits with the layout of gr_point (152 bytes), so the result vector is filled in
# place by the library.
function Gradus.ensemble_solve_tracing_problem(
    ensemble::EnsembleMI355X,
    problem::EnsembleProblem,
    config::TracingConfiguration{Float64};
    progress_bar = nothing,
    save_on = false,
    trace = Gradus.TraceGeodesic(),
    gtol = 1e-2,
    upper_hemisphere = nothing,
    solver_opts...,
)
    save_on && error("Cannot use `EnsembleMI355X` with `save_on`")
    isnothing(progress_bar) || @warn "Progress meter is not supported by EnsembleMI355X."
    N = config.trajectories
    # unconstrained initial velocities: evaluate the (arbitrary) Julia velocity closure on the
    # host; constrain_all is applied on the device
    xs = config.position isa SVector ? [config.position] : config.position
    vs = config.velocity isa Function ? [config.velocity(i) for i = 1:N] : config.velocity
    cfg_val, keep = _config(config, trace; gtol = gtol, upper_hemisphere = upper_hemisphere)
    cfg = Ref(cfg_val)
    out = Vector{GeodesicPoint{Float64,Nothing}}(undef, N)
    @assert sizeof(eltype(out)) == 152
    stats = Ref{GrStats}()
    rc = GC.@preserve keep xs vs ccall((:gr_trace_endpoints, LIB), Int32,
        (Ptr{Cvoid}, Ref{GrConfig}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Cvoid}, Ref{GrStats}),
        ensemble.ctxs[1], cfg, reinterpret(Float64, xs), length(xs) == 1 ? 0 : 4, reinterpret(Float64, vs), N, out, stats)
    _check(rc)
    out
end

# ---- fused fast path: rendergeodesics with a recognised built-in point function -------------------
# `render_into_image!` (src/rendering/rendering.jl:89-101) receives the configuration whose velocity
# is the closure of `_render_velocity_function` (rendering.jl:140-163); its captured variables give
# αlims/βlims/W/H back, and `Mx` is what `lnr_momentum_to_global_velocity_transform` builds.
"""
    BuiltinPF(pf_id, filter_id)

Tag for the point functions the kernels evaluate themselves: `pf_id` 0 = affine_time, 1 = redshift;
`filter_id` 0 = none, 1 = filter_early_term, 2 = filter_intersected.  `render_mi355x` below is what
`rendergeodesics(...; ensemble = EnsembleMI355X(...), pf = BuiltinPF(1, 2))` dispatches to.
"""
struct BuiltinPF
    pf_id::Int32
    filter_id::Int32
end

function render_mi355x(ensemble::EnsembleMI355X, m, x::SVector{4,Float64}, d, λmax; image_width, image_height,
        αlims, βlims, pf::BuiltinPF = BuiltinPF(0, 1), gtol = 1e-2, abstol = 1e-9, reltol = 1e-9,
        chart = Gradus.chart_for_metric(m), q = 0.0)
    id, params = _metric(m)
    did, rin, rout, dparams = _disc(d)
    r_in, r_out, tab, θ0, θ1 = _chart(chart)
    dtab = _disc_table(d)
    cfg = Ref(GrConfig(id, did, params, r_in, r_out, rin, rout, gtol, 0.0, Float64(λmax),
        abstol, reltol, 0.0, 1_000_000, Int32(0), Int32(0), 1e-4, dparams,
        isempty(dtab) ? Ptr{Float64}(C_NULL) : pointer(dtab), length(dtab),
        isempty(tab) ? Ptr{Float64}(C_NULL) : pointer(tab), length(tab), θ0, θ1, Float64(q), Int32(0), Int32(0), π / 2))
    g = Gradus.metric(m, x)
    Mx = inv(g) * hcat(Gradus.lnrbasis(g)...)                       # tracing/utility.jl:32-40
    plane = Ref(GrPlane(Tuple(x), Tuple(permutedims(Mx)), αlims[1], αlims[2], βlims[1], βlims[2],
        image_width, image_height, 1e-6))
    r_isco = pf.pf_id == 1 ? Float64(Gradus.isco(m)) : 0.0
    pfs = Ref(GrPointFunction(pf.pf_id, pf.filter_id, NaN, r_isco, 0, C_NULL, C_NULL, C_NULL, C_NULL))
    image = zeros(Float64, (image_height, image_width))             # rendering.jl:50, column-major H x W
    stats = Vector{GrStats}(undef, length(ensemble.ctxs))
    _check(GC.@preserve tab dtab ccall((:gr_render_multi, LIB), Int32,
        (Ptr{Ptr{Cvoid}}, Int32, Ref{GrConfig}, Ref{GrPlane}, Ref{GrPointFunction}, Int64, Ptr{Float64}, Ptr{GrStats}),
        ensemble.ctxs, length(ensemble.ctxs), cfg, plane, pfs, 0, image, stats))
    α, β = Gradus.impact_axes(image_width, image_height, αlims, βlims)
    α, β, image
end

end # module

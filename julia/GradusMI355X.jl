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

export EnsembleMI355X

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

mutable struct EnsembleMI355X
    device::Int32
    ctx::Ptr{Cvoid}
    function EnsembleMI355X(device::Integer = 0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gr_ctx_create, LIB), Int32, (Int32, Ref{Ptr{Cvoid}}), device, ref)
        rc == 0 || error(unsafe_string(ccall((:gr_last_error, LIB), Cstring, ())))
        ens = new(device, ref[])
        finalizer(e -> ccall((:gr_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), e.ctx), ens)
        ens
    end
end

_metric(m::KerrMetric) = (Int32(0), (m.M, m.a, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
_metric(m::JohannsenMetric) = (Int32(1), (m.M, m.a, m.α13, m.α22, m.α52, m.ϵ3, 0.0, 0.0))
_metric(m) = error("GradusMI355X: metric $(typeof(m)) has no device implementation; use a CPU ensemble")

_disc(::Nothing) = (Int32(0), 0.0, 0.0)
_disc(d::ThinDisc) = (Int32(1), Float64(d.inner_radius), Float64(d.outer_radius))
_disc(d) = error("GradusMI355X: geometry $(typeof(d)) has no device implementation; use a CPU ensemble")

function _config(config::TracingConfiguration, trace::AbstractTrace; gtol = 1e-2, maxiters = 1_000_000)
    id, params = _metric(config.metric)
    did, rin, rout = _disc(config.geometry)
    chart = config.chart::PolarChart
    GrConfig(id, did, params, chart.inner_radius, chart.outer_radius, rin, rout, gtol,
        config.λ_domain[1], config.λ_domain[2], config.abstol, config.reltol, Float64(trace.μ),
        maxiters, Int32(0), Int32(0), 1e-4, (0.0, 0.0, 0.0, 0.0), Ptr{Float64}(C_NULL), 0)
end

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
    solver_opts...,
)
    save_on && error("Cannot use `EnsembleMI355X` with `save_on`")
    isnothing(progress_bar) || @warn "Progress meter is not supported by EnsembleMI355X."
    N = config.trajectories
    # unconstrained initial velocities: evaluate the (arbitrary) Julia velocity closure on the
    # host; constrain_all is applied on the device
    xs = config.position isa SVector ? [config.position] : config.position
    vs = config.velocity isa Function ? [config.velocity(i) for i = 1:N] : config.velocity
    cfg = Ref(_config(config, trace; gtol = gtol))
    out = Vector{GeodesicPoint{Float64,Nothing}}(undef, N)
    @assert sizeof(eltype(out)) == 152
    stats = Ref{GrStats}()
    rc = ccall((:gr_trace_endpoints, LIB), Int32,
        (Ptr{Cvoid}, Ref{GrConfig}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Cvoid}, Ref{GrStats}),
        ensemble.ctx, cfg, reinterpret(Float64, xs), length(xs) == 1 ? 0 : 4, reinterpret(Float64, vs), N, out, stats)
    rc == 0 || error(unsafe_string(ccall((:gr_last_error, LIB), Cstring, ())))
    out
end

end # module

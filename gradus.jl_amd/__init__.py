"""gradus.jl_amd -- MI355X-native backend for Gradus.jl's image-plane render path.

Host-side mirror of the reference API (rendergeodesics / tracegeodesics / PointFunction /
AbstractMetric) over the C ABI of libgradus_mi355x.so (include/gradus_mi355x.h).  All per-ray
work runs in hand-written HIP kernels for gfx950; there is no CPU fallback.
"""
from . import _lib
from ._lib import Context, GradusMI355XError, POINT_DTYPE
from . import device, distributed
from . import corona, reverberation, transfer_functions
from .corona import (BeamedPointSource, BothHemispheres, CoronaGeodesics, EvenGenerator, EvenSampler,
                     GoldenSpiralGenerator, LampPostModel, LowerHemisphere, PowerLawSpectrum, RadialDiscProfile, RingCorona,
                     SourceVelocities,
                     RandomGenerator, WeierstrassSampler, coordtime_at, emissivity_at, emissivity_profile,
                     energy_ratio, lorentz_factor, sky_angles_to_velocity, tetradframe_matrix, tracecorona)
from .distributed import gather_buffers, gather_image, gather_image_async, shard_plan
from .geometry import DatumPlane, EllipticalDisc, PrecessingDisc, ShakuraSunyaev, ThickDisc, ThinDisc
from .polish_doughnut import PolishDoughnut
from .lineprofiles import BinningMethod, PowerLawEmissivity, TransferFunctionMethod, bucket_simple, lineprofile
from .metrics import (BumblebeeMetric, DilatonAxion, JohannsenMetric, JohannsenPsaltisMetric, KerrDarkMatter, KerrMetric,
                      KerrNewmanMetric, KerrRefractive, MorrisThorneWormhole, NoZMetric, SphericalMetric, inner_radius, isco)
from .orthonormalization import lnrbasis, lnrbasis_matrix, lnrframe, lnrframe_matrix
from .planes import (CartesianPlane, GeometricGrid, InverseGrid, LinearGrid, PolarPlane, image_plane,
                     impact_parameters, trajectory_count, unnormalized_areas)
from .pointfunctions import ConstPointFunctions, FilterPointFunction, FilterStatusCode, PointFunction
from .rendering import (EndpointRenderCache, apply, impact_axes, prerendergeodesics, render_configuration,
                        render_into_image, rendergeodesics)
from .status import StatusCodes
from .tracing import (EnsembleMI355X, PolarChart, PoloidalShapeChart, TracingConfiguration, chart_for_metric,
                      domain_upper_hemisphere, event_horizon, event_horizon_chart, is_naked_singularity,
                      ensemble_solve_tracing_problem, local_momentum, lnr_momentum_to_global_velocity_transform,
                      map_impact_parameters, tracegeodesic_path, tracegeodesic_paths, tracegeodesics, tracing_configuration)
from .transfer_functions import (CunninghamTransferData, InterpolatingTransferBranches, TransferBranches,
                                cunningham_transfer_function, cunningham_transfer_functions, integrate_lagtransfer,
                                integrate_lineprofile,
                                interpolate_branches, splitbranches, transferfunctions)
from .reverberation import (AnalyticRadialDiscProfile, LagTransferFunction, bin_transfer_function, binflux, continuum_time,
                            lag_frequency,
                            lagtransfer, observer_to_disc)
from .precision_solvers import (find_offset_for_radius, impact_parameters_for_radius, impact_parameters_for_radius_obscured,
                                impact_parameters_for_target, jacobian_αβ_gr, optimize_for_target)
from .special_radii import (CircularOrbits, PlungingInterpolation, generic_isco, interpolate_plunging_velocities,
                            plunging_fourvelocity)

__all__ = [n for n in dir() if not n.startswith("_")]

interpolate_redshift = ConstPointFunctions.interpolate_redshift

"""ctypes binding of libgradus_mi355x.so (the C ABI declared in include/gradus_mi355x.h).

There is deliberately no fallback: if the shared library is missing, or no gfx950 device is
present when a context is created, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRADUS_MI355X_LIB selects another build of the same library (A/B timing of compiler flags)
LIB_PATH = os.environ.get("GRADUS_MI355X_LIB") or os.path.join(_HERE, "csrc", "libgradus_mi355x.so")



def kernel_source_sha16() -> str:
    """sha256[:16] over the kernel sources of the library in this tree with comments and white space removed (a
    comment edit is not a different kernel).  rocprofv3 summaries under profiles/ record it, and bench.py refuses to
    price a launch with counters taken from a different build of the kernels."""
    import hashlib
    import re

    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    for rel in ("gradus.jl_amd/csrc/gr_device.hpp", "gradus.jl_amd/csrc/gr_kernels.hpp", "gradus.jl_amd/csrc/gr_tangent.hpp",
                "gradus.jl_amd/csrc/gr_tabmetric.hpp", "gradus.jl_amd/csrc/metric_table.hip",
                "gradus.jl_amd/csrc/kernels_tu.hip",
                "gradus.jl_amd/csrc/gradus_mi355x.hip",
                "include/gradus_mi355x.h"):
        with open(os.path.join(root, rel), "r", encoding="utf-8") as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", " ", text)                    # line comments (no string literal here holds "//")
        h.update(" ".join(text.split()).encode("utf-8"))
    return h.hexdigest()[:16]


GR_OK = 0
ABI_VERSION = 8      # GR_ABI_VERSION of include/gradus_mi355x.h this module mirrors (checked in load() and tests/test_host_api.py)
ERROR_NAMES = {
    -1: "GR_ERR_INVALID_ARGUMENT",
    -2: "GR_ERR_UNSUPPORTED",
    -3: "GR_ERR_NO_DEVICE",
    -4: "GR_ERR_HIP",
    -5: "GR_ERR_OUT_OF_MEMORY",
}


class GradusMI355XError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class gr_disc_component(C.Structure):
    _fields_ = [("disc_id", C.c_int32), ("_pad", C.c_int32), ("disc_r_in", C.c_double), ("disc_r_out", C.c_double),
                ("disc_params", C.c_double * 4)]


GR_COMP_MAX = 4


class gr_config(C.Structure):
    _fields_ = [
        ("metric_id", C.c_int32),
        ("disc_id", C.c_int32),
        ("params", C.c_double * 8),
        ("r_inner", C.c_double),
        ("r_outer", C.c_double),
        ("disc_r_in", C.c_double),
        ("disc_r_out", C.c_double),
        ("gtol", C.c_double),
        ("lambda0", C.c_double),
        ("lambda1", C.c_double),
        ("abstol", C.c_double),
        ("reltol", C.c_double),
        ("mu", C.c_double),
        ("maxiters", C.c_int64),
        ("upper_hemisphere", C.c_int32),
        ("_pad", C.c_int32),
        ("hemi_delta", C.c_double),
        ("disc_params", C.c_double * 4),
        ("disc_table", C.c_void_p),
        ("disc_table_n", C.c_int64),
        ("chart_table", C.c_void_p),
        ("chart_table_n", C.c_int64),
        ("chart_theta0", C.c_double),
        ("chart_theta1", C.c_double),
        ("q", C.c_double),
        ("count_windings", C.c_int32),
        ("_pad2", C.c_int32),
        ("winding_plane", C.c_double),
        ("comp_n", C.c_int32),
        ("_pad3", C.c_int32),
        ("comp", gr_disc_component * GR_COMP_MAX),
        ("metric_table", C.c_void_p),      # GR_METRIC_TABULATED (ABI 7): the table gr_metric_table_fit wrote, host pointer
        ("metric_table_n", C.c_int64),
    ]


GR_METRIC_MAX_SEG = 12


class gr_metric_break(C.Structure):
    """A radius where metric_components changes form (scale 0), or the centre of a feature that narrow (scale > 0)."""

    _fields_ = [("radius", C.c_double), ("scale", C.c_double)]


class gr_metric_segment(C.Structure):
    _fields_ = [
        ("r_lo", C.c_double),
        ("r_hi", C.c_double),
        ("anchor", C.c_double),
        ("xmin", C.c_double),
        ("fit_lo", C.c_double),
        ("fit_hi", C.c_double),
        ("e_lo", C.c_int32),
        ("e_hi", C.c_int32),
        ("first_row", C.c_int32),
        ("n_rows", C.c_int32),
        ("dir", C.c_int32),
        ("core", C.c_int32),
    ]


class gr_metric_grid(C.Structure):
    """Patch grid of a tabulated metric (include/gradus_mi355x.h, "tabulated metrics")."""

    _fields_ = [
        ("r0", C.c_double),
        ("r_min", C.c_double),
        ("r_max", C.c_double),
        ("e_min", C.c_int32),
        ("n_oct", C.c_int32),
        ("m_r", C.c_int32),
        ("n_theta", C.c_int32),
        ("degree", C.c_int32),
        ("fit_nodes", C.c_int32),
        ("pole_factor", C.c_int32),
        ("n_seg", C.c_int32),
        ("n_r_nodes", C.c_int64),
        ("n_theta_nodes", C.c_int64),
        ("table_doubles", C.c_int64),
        ("n_rows", C.c_int32),
        ("reserved", C.c_int32),
        ("seg", gr_metric_segment * GR_METRIC_MAX_SEG),
    ]


class gr_plane(C.Structure):
    _fields_ = [
        ("x_obs", C.c_double * 4),
        ("Mx", C.c_double * 16),
        ("alpha0", C.c_double),
        ("alpha1", C.c_double),
        ("beta0", C.c_double),
        ("beta1", C.c_double),
        ("width", C.c_int64),
        ("height", C.c_int64),
        ("offset", C.c_double),
    ]


class gr_range(C.Structure):
    _fields_ = [("first", C.c_int64), ("count", C.c_int64), ("block", C.c_int64), ("stride_blocks", C.c_int64)]


class gr_pointfunction(C.Structure):
    _fields_ = [
        ("pf_id", C.c_int32),
        ("filter_id", C.c_int32),
        ("fill", C.c_double),
        ("r_isco", C.c_double),
        ("n_plunge", C.c_int64),
        ("plunge_r", C.POINTER(C.c_double)),
        ("plunge_vt", C.POINTER(C.c_double)),
        ("plunge_vr", C.POINTER(C.c_double)),
        ("plunge_vphi", C.POINTER(C.c_double)),
        ("has_u_src", C.c_int32),
        ("_pad_u", C.c_int32),
        ("u_src", C.c_double * 4),
    ]


class gr_rayset(C.Structure):
    _fields_ = [
        ("x_obs", C.c_double * 4),
        ("Mx", C.c_double * 16),
        ("alpha", C.c_void_p),
        ("beta", C.c_void_p),
        ("area", C.c_void_p),
        ("n", C.c_int64),
        ("height", C.c_void_p),
        ("sep_r", C.c_void_p),
        ("sep_cos", C.c_void_p),
        ("sep_sin", C.c_void_p),
        ("sep_nr", C.c_int64),
        ("sep_nt", C.c_int64),
        ("sep_tiled", C.c_int32),
        ("sep_reserved", C.c_int32),
        ("sep_first", C.c_int64),
        ("sep_block", C.c_int64),
        ("sep_stride", C.c_int64),
        ("sky_sampler", C.c_int32),
        ("sky_both", C.c_int32),
        ("sky_generator", C.c_int32),
        ("sky_reserved", C.c_int32),
        ("sky_resolution", C.c_double),
        ("sky_i", C.c_void_p),
        ("sky_first", C.c_int64),          # ABI 8: a share of a source's samples (the *_multi entry points set them per context)
        ("sky_total", C.c_int64),
        ("sky_rows", C.c_void_p),          # ABI 8: a source without one position: 28 doubles per ray (x, Mx, lowered source velocity, g_tμ)
    ]


class gr_binning(C.Structure):
    _fields_ = [
        ("r_min", C.c_double),
        ("r_max", C.c_double),
        ("emissivity_index", C.c_double),
        ("n_bins", C.c_int64),
        ("bin_edges", C.c_void_p),
        ("eps_r", C.c_void_p),
        ("eps_v", C.c_void_p),
        ("eps_n", C.c_int64),
    ]


class gr_stats(C.Structure):
    _fields_ = [
        ("rays", C.c_int64),
        ("accepted_steps", C.c_int64),
        ("rejected_steps", C.c_int64),
        ("rhs_evals", C.c_int64),
        ("flagged_rays", C.c_int64),
        ("status_count", C.c_int64 * 4),
        ("kernel_ms", C.c_double),     # host variants: start of the call's device work -> end of its last trace kernel
        ("call_ms", C.c_double),       # ... -> end of the last copy into the caller's buffer (ABI 5)
        ("enqueue_ms", C.c_double),    # *_multi: host time spent enqueueing this context's share (ABI 6); 0 elsewhere
    ]

    def asdict(self):
        return {
            "rays": self.rays,
            "accepted_steps": self.accepted_steps,
            "rejected_steps": self.rejected_steps,
            "rhs_evals": self.rhs_evals,
            "flagged_rays": self.flagged_rays,
            "status_count": list(self.status_count),
            "kernel_ms": self.kernel_ms,
            "call_ms": self.call_ms,
            "enqueue_ms": self.enqueue_ms,
        }


# GeodesicPoint{Float64,Nothing}: 152 bytes (src/solution-processing.jl:15-32)
POINT_DTYPE = np.dtype(
    [
        ("status", np.int32),
        ("flags", np.int32),
        ("lambda_min", np.float64),
        ("lambda_max", np.float64),
        ("x_init", np.float64, (4,)),
        ("x", np.float64, (4,)),
        ("v_init", np.float64, (4,)),
        ("v", np.float64, (4,)),
    ],
    align=True,
)
assert POINT_DTYPE.itemsize == 152

# every symbol include/gradus_mi355x.h declares
EXPORTS = [
    "gr_abi_version",
    "gr_last_error",
    "gr_ctx_create",
    "gr_ctx_destroy",
    "gr_ctx_set",
    "gr_host_alloc",
    "gr_host_free",
    "gr_render_device",
    "gr_render",
    "gr_render_multi",
    "gr_render_endpoints_device",
    "gr_render_endpoints",
    "gr_trace_endpoints_device",
    "gr_trace_endpoints",
    "gr_trace_path",
    "gr_trace_paths",
    "gr_lineprofile_device",
    "gr_lineprofile",
    "gr_redshift_radius_device",
    "gr_redshift_radius",
    "gr_ray_summary_device",
    "gr_ray_summary",
    "gr_ray_tangent_device",
    "gr_ray_tangent",
    "gr_rayset_endpoints_device",
    "gr_rayset_endpoints",
    "gr_apply_pointfunction_device",
    "gr_apply_pointfunction",
    "gr_corona_trace",
    "gr_corona_bin",
    "gr_corona_trace_multi",
    "gr_corona_bin_multi",
    "gr_render_endpoints_multi",
    "gr_trace_endpoints_multi",
    "gr_rayset_endpoints_multi",
    "gr_ray_summary_multi",
    "gr_ray_tangent_multi",
    "gr_redshift_radius_multi",
    "gr_lineprofile_multi",
    "gr_metric_grid_plan",
    "gr_metric_grid_plan_breaks",
    "gr_metric_grid_nodes",
    "gr_metric_table_fit",
    "gr_metric_table_eval",
]

_lib = None


def load():
    """Load the shared library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GradusMI355XError(
            -3, f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
        )
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.gr_abi_version.restype = i32
    # the structs below are laid out for ONE version of include/gradus_mi355x.h: a library of another version (an older
    # build named by GRADUS_MI355X_LIB, a variant of scripts/build_variant.sh) would read them wrongly without any error
    if L.gr_abi_version() != ABI_VERSION:
        raise GradusMI355XError(-1, f"{LIB_PATH} has ABI version {L.gr_abi_version()}, this binding is written for {ABI_VERSION}")
    L.gr_last_error.restype = C.c_char_p
    L.gr_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.gr_ctx_destroy.argtypes = [vp]
    L.gr_ctx_set.argtypes = [vp, C.c_char_p, i64]
    L.gr_host_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    L.gr_host_free.argtypes = [vp, vp]
    cfgp, plp, pfp, rgp, stp = (C.POINTER(t) for t in (gr_config, gr_plane, gr_pointfunction, gr_range, gr_stats))
    L.gr_render_device.argtypes = [vp, cfgp, plp, pfp, rgp, vp, vp, vp]
    L.gr_render.argtypes = [vp, cfgp, plp, pfp, rgp, vp, stp]
    L.gr_render_multi.argtypes = [C.POINTER(vp), i32, cfgp, plp, pfp, i64, vp, vp]
    L.gr_render_endpoints_device.argtypes = [vp, cfgp, plp, rgp, vp, vp, vp]
    L.gr_render_endpoints.argtypes = [vp, cfgp, plp, rgp, vp, stp]
    L.gr_trace_endpoints_device.argtypes = [vp, cfgp, vp, i64, vp, i64, vp, vp, vp]
    L.gr_trace_endpoints.argtypes = [vp, cfgp, vp, i64, vp, i64, vp, stp]
    L.gr_trace_path.argtypes = [vp, cfgp, vp, vp, i64, vp, C.POINTER(i64), vp]
    L.gr_trace_paths.argtypes = [vp, cfgp, vp, i64, vp, i64, i64, vp, vp, vp]
    rsp, bnp = C.POINTER(gr_rayset), C.POINTER(gr_binning)
    L.gr_lineprofile_device.argtypes = [vp, cfgp, rsp, pfp, bnp, vp, vp, vp]
    L.gr_lineprofile.argtypes = [vp, cfgp, rsp, pfp, bnp, vp, stp]
    L.gr_redshift_radius_device.argtypes = [vp, cfgp, rsp, pfp, C.c_double, C.c_double, vp, vp, vp]
    L.gr_redshift_radius.argtypes = [vp, cfgp, rsp, pfp, C.c_double, C.c_double, vp, stp]
    L.gr_ray_summary_device.argtypes = [vp, cfgp, rsp, pfp, vp, vp, vp]
    L.gr_ray_summary.argtypes = [vp, cfgp, rsp, pfp, vp, stp]
    L.gr_ray_tangent_device.argtypes = [vp, cfgp, rsp, pfp, vp, vp, vp]
    L.gr_ray_tangent.argtypes = [vp, cfgp, rsp, pfp, vp, stp]
    L.gr_rayset_endpoints_device.argtypes = [vp, cfgp, rsp, vp, vp, vp]
    L.gr_rayset_endpoints.argtypes = [vp, cfgp, rsp, vp, stp]
    L.gr_apply_pointfunction_device.argtypes = [vp, cfgp, pfp, vp, i64, C.c_double, vp, vp]
    L.gr_apply_pointfunction.argtypes = [vp, cfgp, pfp, vp, i64, C.c_double, vp]
    L.gr_corona_trace.argtypes = [vp, cfgp, rsp, pfp, vp, C.POINTER(C.c_int64), stp]
    L.gr_corona_bin.argtypes = [vp, vp, i64, vp]
    L.gr_corona_trace_multi.argtypes = [vp, i32, cfgp, rsp, pfp, vp, C.POINTER(C.c_int64), stp]
    L.gr_corona_bin_multi.argtypes = [vp, i32, vp, i64, vp]
    ctxa = C.POINTER(vp)
    L.gr_render_endpoints_multi.argtypes = [ctxa, i32, cfgp, plp, i64, vp, vp]
    L.gr_trace_endpoints_multi.argtypes = [ctxa, i32, cfgp, vp, i64, vp, i64, vp, vp]
    L.gr_rayset_endpoints_multi.argtypes = [ctxa, i32, cfgp, rsp, vp, vp]
    L.gr_ray_summary_multi.argtypes = [ctxa, i32, cfgp, rsp, pfp, vp, vp]
    L.gr_ray_tangent_multi.argtypes = [ctxa, i32, cfgp, rsp, pfp, vp, vp]
    L.gr_redshift_radius_multi.argtypes = [ctxa, i32, cfgp, rsp, pfp, C.c_double, C.c_double, vp, vp]
    L.gr_lineprofile_multi.argtypes = [ctxa, i32, cfgp, rsp, pfp, bnp, vp, vp]
    gdp, dp = C.POINTER(gr_metric_grid), C.POINTER(C.c_double)
    L.gr_metric_grid_plan.argtypes = [C.c_double, C.c_double, C.c_double, i32, i32, gdp]
    L.gr_metric_grid_plan_breaks.argtypes = [C.c_double, C.c_double, C.c_double, i32, i32, i32, C.POINTER(gr_metric_break), gdp]
    L.gr_metric_grid_nodes.argtypes = [gdp, vp, vp]
    L.gr_metric_table_fit.argtypes = [gdp, vp, vp, dp]
    L.gr_metric_table_eval.argtypes = [vp, i64, C.c_double, C.c_double, dp, dp, dp]
    for name in EXPORTS:
        if name not in ("gr_last_error",):
            getattr(L, name).restype = i32
    L.gr_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(code):
    if code != GR_OK:
        raise GradusMI355XError(code, load().gr_last_error().decode("utf-8", "replace"))


def ctx_array(ctxs):
    """(ctypes array of the contexts' handles, array of gr_stats, one per context) for a *_multi call."""
    arr = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
    return arr, (gr_stats * len(ctxs))()


def merge_stats(sts) -> gr_stats:
    """One gr_stats for a *_multi call: counters summed over the contexts, times = the slowest context
    (they run side by side), enqueue_ms = the sum (the host enqueues them one after another)."""
    st = gr_stats()
    for f in ("rays", "accepted_steps", "rejected_steps", "rhs_evals", "flagged_rays"):
        setattr(st, f, sum(getattr(x, f) for x in sts))
    for q in range(4):
        st.status_count[q] = sum(x.status_count[q] for x in sts)
    st.kernel_ms = max(x.kernel_ms for x in sts)
    st.call_ms = max(x.call_ms for x in sts)
    st.enqueue_ms = sum(x.enqueue_ms for x in sts)
    return st


class PinnedBlock:
    """A result buffer the LIBRARY page-locked (gr_host_alloc, ABI 5): what the Julia shim wraps as its
    Vector{GeodesicPoint} so that 152 B per ray come back by DMA under the trace instead of through the pageable path.
    `array(dtype, count)` views it as a numpy array that keeps the block alive; the block is returned when the last view
    is gone (gr_host_free needs no context)."""

    def __init__(self, ctx, nbytes: int):
        p = C.c_void_p()
        check(load().gr_host_alloc(ctx.handle, int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def array(self, dtype, count: int):
        buf = (C.c_char * self.nbytes).from_address(self.ptr)
        buf._pinned_block = self                      # the view keeps the block alive
        return np.frombuffer(buf, dtype=dtype, count=count)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                load().gr_host_free(None, C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


PINNED_RESULT_MIN_BYTES = 64 << 20


def result_points(ctx, n: int):
    """The array an end-point call fills: pinned by the library from 64 MiB up (GRADUS_MI355X_PINNED_RESULTS=0: never),
    an ordinary numpy array below that or when page-locking is refused."""
    nbytes = int(n) * POINT_DTYPE.itemsize
    if nbytes >= PINNED_RESULT_MIN_BYTES and os.environ.get("GRADUS_MI355X_PINNED_RESULTS", "1") != "0":
        try:
            return PinnedBlock(ctx, nbytes).array(POINT_DTYPE, int(n))
        except GradusMI355XError:
            pass
    return np.zeros(int(n), dtype=POINT_DTYPE)


PINNED_IMAGE_MIN_BYTES = 8 << 20


def result_image(ctx, n: int):
    """The image a fused render fills: from 8 MiB up a block the library page-locked (registered huge pages, taken from the
    pool of freed blocks when one fits), which the kernel writes across the link itself -- no staging image in HBM, no copy;
    below that, or when page-locking is refused (or GRADUS_MI355X_PINNED_RESULTS=0), an ordinary numpy array."""
    nbytes = int(n) * 8
    if nbytes >= PINNED_IMAGE_MIN_BYTES and os.environ.get("GRADUS_MI355X_PINNED_RESULTS", "1") != "0":
        try:
            return PinnedBlock(ctx, nbytes).array(np.float64, int(n))
        except GradusMI355XError:
            pass
    return np.zeros(int(n))


class Context:
    """Owns one gr_ctx (one HIP device)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        check(self._lib.gr_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def set(self, key: str, value: int):
        check(self._lib.gr_ctx_set(self.handle, key.encode(), int(value)))
        return self

    def close(self):
        if getattr(self, "handle", None):
            self._lib.gr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

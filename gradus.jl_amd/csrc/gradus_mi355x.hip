// gradus_mi355x.hip -- host side of libgradus_mi355x.so: the C ABI of include/gradus_mi355x.h, contexts, staging and the
// dispatch to the per-metric kernel objects (kernels_tu.hip, one translation unit per metric id and precision).
//
// Two launch shapes for the same per-lane integrator (gr_device.hpp):
//   kernel 0  "lane"        one ray per work-item, 8x8-pixel tiles per wave; a wave lives as
//                           long as its slowest ray.
//   kernel 1  "persistent"  a resident grid pulls rays from a global counter; when enough lanes
//                           of a wave have finished (wave ballot), their results are written and
//                           they are refilled with new rays, so lanes stay busy although step
//                           counts differ ~4x between rays.
// Output is one double per ray (fused PointFunction) or one 152-byte GeodesicPoint per ray.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define GR_NS gr
#include "gr_device.hpp"
#include "gr_mesh_grid.hpp"

using namespace gr;

// ---- the per-metric kernel objects (kernels_tu.hip) ----
#define GR_DECLARE_METRIC(ID)                                                                                          \
    hipError_t gr64_launch_trace_m##ID(int, int, int, int, unsigned long long*, const void*, hipStream_t);             \
    hipError_t gr32_launch_trace_m##ID(int, int, int, int, unsigned long long*, const void*, hipStream_t);             \
    hipError_t grt_launch_trace_m##ID(int, int, int, int, unsigned long long*, const void*, hipStream_t);              \
    hipError_t grt1_launch_trace_m##ID(int, int, int, int, unsigned long long*, const void*, hipStream_t);             \
    hipError_t gr64_launch_path_m##ID(const void*, double*, int64_t, unsigned long long*, hipStream_t);                \
    hipError_t gr64_launch_apply_m##ID(const void*, const gr_point*, double, double*, hipStream_t);
GR_DECLARE_METRIC(0) GR_DECLARE_METRIC(1) GR_DECLARE_METRIC(2) GR_DECLARE_METRIC(3) GR_DECLARE_METRIC(4) GR_DECLARE_METRIC(5)
GR_DECLARE_METRIC(6) GR_DECLARE_METRIC(7) GR_DECLARE_METRIC(8) GR_DECLARE_METRIC(9) GR_DECLARE_METRIC(10)
#undef GR_DECLARE_METRIC
// GR_METRIC_TABULATED (11): every flavour of the trace kernels (the fp32 ones evaluate the fp64 table in double, gr_device.hpp)
hipError_t gr64_launch_trace_m11(int, int, int, int, unsigned long long*, const void*, hipStream_t);
hipError_t gr32_launch_trace_m11(int, int, int, int, unsigned long long*, const void*, hipStream_t);
hipError_t grt_launch_trace_m11(int, int, int, int, unsigned long long*, const void*, hipStream_t);
hipError_t grt1_launch_trace_m11(int, int, int, int, unsigned long long*, const void*, hipStream_t);
hipError_t gr64_launch_path_m11(const void*, double*, int64_t, unsigned long long*, hipStream_t);
hipError_t gr64_launch_apply_m11(const void*, const gr_point*, double, double*, hipStream_t);
static_assert(GR_METRIC_NOZ == 10 && GR_METRIC_TABULATED == 11, "one kernel object per metric id 0..11: extend the tables below with the catalogue");

namespace {
struct LaunchKnobs {
    int kernel;              // 0 = lane, 1 = persistent
    int block;
    int n_cu;
    int waves_per_simd;      // 0 = from the occupancy query
    unsigned long long* queue;   // work counter of the persistent kernel (zeroed by the launcher)
};
typedef hipError_t (*trace_fn)(int, int, int, int, unsigned long long*, const void*, hipStream_t);
typedef hipError_t (*path_fn)(const void*, double*, int64_t, unsigned long long*, hipStream_t);
typedef hipError_t (*apply_fn)(const void*, const gr_point*, double, double*, hipStream_t);
#define GR_ROW(F, LAST) { F##0, F##1, F##2, F##3, F##4, F##5, F##6, F##7, F##8, F##9, F##10, LAST }
const trace_fn kTrace64[12] = GR_ROW(gr64_launch_trace_m, gr64_launch_trace_m11);
const trace_fn kTrace32[12] = GR_ROW(gr32_launch_trace_m, gr32_launch_trace_m11);
const trace_fn kTraceTan[12] = GR_ROW(grt_launch_trace_m, grt_launch_trace_m11);      // value + ∂/∂α + ∂/∂β (out_mode 5): one lane per ray
const trace_fn kTraceTan1[12] = GR_ROW(grt1_launch_trace_m, grt1_launch_trace_m11);   // the same with a PAIR of lanes per ray (kernels_tu.hip)
const path_fn kPath64[12] = GR_ROW(gr64_launch_path_m, gr64_launch_path_m11);
const apply_fn kApply64[12] = GR_ROW(gr64_launch_apply_m, gr64_launch_apply_m11);
#undef GR_ROW
}  // namespace

namespace {

thread_local std::string g_last_error;
std::atomic<int> g_live_ctx{ 0 };     // contexts alive: the pool of page-locked blocks is emptied when the last one goes

int32_t fail(int32_t code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

}  // namespace
// for the library's other host units (metric_table.hip)
int32_t gr_set_last_error(int32_t code, const char* msg) { return fail(code, msg); }
int32_t gr_metric_table_check(const double* table, int64_t table_n);      // metric_table.hip
namespace {

#define GR_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? GR_ERR_OUT_OF_MEMORY : GR_ERR_HIP,            \
                        std::string(#call) + ": " + hipGetErrorString(e_));                       \
    } while (0)

}  // namespace

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
struct gr_ctx {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;          // used by the host-buffer entry points
    unsigned long long* d_queue = nullptr; // ring of work counters (one per in-flight launch)
    int queue_slots = 64, queue_next = 0;
    unsigned long long* d_stats = nullptr; // for host-buffer entry points
    double* d_disc_table = nullptr;        // device copy of a tabulated disc profile
    size_t disc_table_bytes = 0;
    double* d_mesh = nullptr;              // GR_DISC_MESH: the grid-sorted triangle table (gr_mesh_grid.hpp), kept while the
    size_t mesh_bytes = 0;                 //   caller passes the same mesh (fingerprint of its table)
    uint64_t mesh_fp = 0;
    int64_t mesh_n = -1;
    std::vector<double> mesh_host;
    double* d_sky = nullptr;               // a sky source's (x, v) arrays, written by k_sky_velocities for the trace kernels
    unsigned* d_sky_table = nullptr;       // ... dealt by cost: counts, then offsets, per (class, chunk) (k_sky_velocities_dealt)
    size_t sky_table_bytes = 0;
    size_t sky_bytes = 0;
    int64_t sky_first = 0, sky_total = 0;  // the share of a sky source the launch being prepared traces (rays_params -> sky_prepare)
    bool sky_any_order = false;            // ... whose rows may come in any order (gr_corona_trace): the rays are dealt by predicted cost
    const double* sky_rows = nullptr;      // ... and its per-sample rows (gr_rayset.sky_rows, device) or null
    double* d_corona = nullptr;            // gr_corona_trace: (g, ρ, t, status) per ray, kept for gr_corona_bin
    size_t corona_bytes = 0;
    int64_t corona_n = -1, corona_hits = 0;
    double corona_gmax = 0.0, corona_tmax = 0.0;
    double* d_metric_table = nullptr;      // GR_METRIC_TABULATED: device copy of the caller's table, kept while its build id stays
    size_t metric_table_bytes = 0;
    double metric_table_id = 0.0;
    double* d_chart_table = nullptr;       // device copy of a PoloidalShapeChart table
    size_t chart_table_bytes = 0;
    Cold* d_cold = nullptr;                // ring of per-launch cold blocks
    int cold_next = 0;
    double* d_plunge = nullptr;            // 4 x n_plunge
    int64_t plunge_cap = 0;
    void* d_scratch = nullptr;             // staging for host-buffer entry points
    size_t scratch_bytes = 0;
    void* d_in = nullptr;
    size_t in_bytes = 0;
    // knobs
    int64_t kernel = 2;                    // 0 lane, 1 persistent, 2 auto (by launch depth)
    int64_t lpt_lane = 0;                  // also order tiles longest-first for the lane kernel
    int64_t block = 0;                    // 0 = auto: 64 for the one-ray-per-lane kernel, 256 for the persistent one
    int64_t refill_threshold = 0;          // idle lanes that trigger a refill; 0 = auto: 16 for the fp64 kernels, 32 for the fp32 ones
    int64_t waves_per_simd = 0;            // 0 = from occupancy query
    int64_t swizzle = 1;
    int64_t tile_rows = 8;                 // rows of the pixel tile of a wave: 8 (8 x 8) or 16 (16 x 4: whole 128-B lines per store)
    int64_t precision = 64;                // 64 = fp64 kernels, 32 = fp32 kernels (tolerance sweeps)
    int64_t lds = 1;                       // stage the plunging table / line-profile histogram in LDS
    int64_t lpt = 1;                       // longest-first tile order learned from the previous render
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_k = nullptr;             // end of the last trace kernel of a host call (gr_stats.kernel_ms vs call_ms)
    int64_t hugepages = 1;                 // madvise(MADV_HUGEPAGE) on large caller-owned result buffers before pre-faulting
    int64_t lds_points = 1;                // one-ray-per-lane kernel: a wave's end-point records leave through LDS as whole runs
    int64_t direct_host = 1;               // gr_render_endpoints into a gr_host_alloc block: the kernel stores across the link itself
    int64_t tangent_pairs = 2;             // tangent kernels: 0 = one lane per ray, 1 = a pair of lanes per ray, 2 = by launch size
    int64_t sky_deal = 1;                  // gr_corona_trace: the sky rays of a source dealt to the waves by predicted cost (k_sky_velocities_dealt)
    int64_t xcd_spread = 1;                // one-ray-per-lane kernel, rays in caller order: chunks dealt over the XCDs by digit sum
    int64_t tangent_norm = 1;              // tangent kernels: the tangents are part of the error norm (DiffEqBase on Dual state); 0 = values only
    // LPT state for one (config, plane, range) key
    std::vector<unsigned char> lpt_key;
    uint32_t* d_tile_cost = nullptr;
    uint32_t* d_tile_perm = nullptr;
    int64_t lpt_tiles = 0, lpt_cap = 0;
    bool lpt_have_perm = false, lpt_cost_pending = false;
    hipEvent_t ev_cost = nullptr;
    // The staged tables (plunging table, disc profile, chart) are single ctx-owned buffers.  A launch that reads
    // them records `ev_tables` on its stream; the next staging on a DIFFERENT stream first waits for that event,
    // so a table is never overwritten under a kernel that is still reading it (launches without tables -- Kerr
    // with a thin disc, the bench workload -- never wait on each other).
    hipEvent_t ev_tables = nullptr;
    hipStream_t tables_stream = nullptr;
    bool tables_busy = false;
    // host-buffer entry points that return 152 B per ray: the result goes back in bands on a second stream while later
    // bands are still being traced (see copy_back_in_bands)
    hipStream_t copy_stream = nullptr;
    hipStream_t band_stream = nullptr;     // odd bands are traced here, even ones on `stream`: the tail of one band's launch
                                           // (SIMDs draining) overlaps the head of the next
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_band[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    bool lpt_suspend = false;              // banded launches do not learn / use a tile order (their ranges differ)
    int64_t pipeline = 4;                  // bands of the end-point return (0 / 1: one launch + one copy)
    bool counted = false;                  // this context is in g_live_ctx
    bool multi_direct = false;             // *_multi: this context's kernel stores into the caller's pinned block itself (no copy in phase 2)
};

namespace {

int32_t ensure(void** buf, size_t* cap, size_t need)
{
    if (*cap >= need) return GR_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    GR_HIP(hipMalloc(buf, need));
    *cap = need;
    return GR_OK;
}

// make `stream` wait for the last launch that read the ctx-owned staged tables (if it ran on another stream)
int32_t tables_acquire(gr_ctx* ctx, hipStream_t stream)
{
    if (ctx->tables_busy && ctx->tables_stream != stream) GR_HIP(hipStreamWaitEvent(stream, ctx->ev_tables, 0));
    return GR_OK;
}
// called after a launch that reads staged tables
int32_t tables_release(gr_ctx* ctx, hipStream_t stream)
{
    GR_HIP(hipEventRecord(ctx->ev_tables, stream));
    ctx->tables_stream = stream;
    ctx->tables_busy = true;
    return GR_OK;
}

int32_t validate_cfg(const gr_config* cfg)
{
    if (!cfg) return fail(GR_ERR_INVALID_ARGUMENT, "config is null");
    if (cfg->metric_id < GR_METRIC_KERR || cfg->metric_id > GR_METRIC_TABULATED)
        return fail(GR_ERR_UNSUPPORTED, "unknown metric_id " + std::to_string(cfg->metric_id));
    if (cfg->metric_id == GR_METRIC_TABULATED) {
        const int32_t trc = gr_metric_table_check(cfg->metric_table, cfg->metric_table_n);
        if (trc != GR_OK) return trc;
        // The table is polynomials: outside [r_min, r_max] they extrapolate to garbage (g_tt = -3e4 at ten times r_max).  A ray
        // lives between the chart's boundaries, so the chart has to lie inside the table (a hair of slack: the Python and Julia
        // hosts plan r_min a part in a thousand inside the chart's inner radius).
        const double t_min = cfg->metric_table[gr_tab::H_RMIN], t_max = cfg->metric_table[gr_tab::H_RMAX];
        double c_in = cfg->r_inner;
        if (cfg->chart_table_n > 1 && cfg->chart_table)
            for (int64_t k = 0; k < cfg->chart_table_n; ++k) c_in = k == 0 ? cfg->chart_table[k] : std::fmin(c_in, cfg->chart_table[k]);
        if (!(c_in >= t_min - 1e-9 * std::fabs(t_min) - 1e-12) || !(cfg->r_outer <= t_max + 1e-9 * std::fabs(t_max)))
            return fail(GR_ERR_INVALID_ARGUMENT, "GR_METRIC_TABULATED: the chart [" + std::to_string(c_in) + ", " + std::to_string(cfg->r_outer)
                                                 + "] leaves the radial range of the metric table [" + std::to_string(t_min) + ", "
                                                 + std::to_string(t_max) + "]: fit the table on a range that contains the chart");
    }
    if (cfg->disc_id < GR_DISC_NONE || cfg->disc_id > GR_DISC_MESH)
        return fail(GR_ERR_UNSUPPORTED, "unknown disc_id " + std::to_string(cfg->disc_id));
    if (cfg->disc_id == GR_DISC_COMPOSITE) {
        if (cfg->comp_n < 2 || cfg->comp_n > GR_COMP_MAX)
            return fail(GR_ERR_INVALID_ARGUMENT, "a composite geometry has 2.." + std::to_string(GR_COMP_MAX) + " components");
        for (int k = 0; k < cfg->comp_n; ++k) {
            const int id = cfg->comp[k].disc_id;
            if (id != GR_DISC_THIN && id != GR_DISC_SHAKURA_SUNYAEV && id != GR_DISC_ELLIPTICAL && id != GR_DISC_DATUM)
                return fail(GR_ERR_UNSUPPORTED, "composite geometry: component " + std::to_string(k) + " must be a thin disc, a Shakura-Sunyaev "
                                                "disc, an elliptical disc or a datum plane");
            if (id == GR_DISC_THIN && !(cfg->comp[k].disc_r_out >= cfg->comp[k].disc_r_in))
                return fail(GR_ERR_INVALID_ARGUMENT, "composite geometry: disc outer radius below inner radius");
        }
    }
    if (!(cfg->abstol > 0.0) || !(cfg->reltol > 0.0))
        return fail(GR_ERR_INVALID_ARGUMENT, "abstol and reltol must be positive");
    if (!(cfg->lambda1 > cfg->lambda0))
        return fail(GR_ERR_INVALID_ARGUMENT, "λ domain must be increasing");
    if (cfg->maxiters <= 0) return fail(GR_ERR_INVALID_ARGUMENT, "maxiters must be positive");
    if (cfg->disc_id == GR_DISC_TABULATED && (!cfg->disc_table || cfg->disc_table_n < 2 || !(cfg->disc_params[1] > cfg->disc_params[0])))
        return fail(GR_ERR_INVALID_ARGUMENT, "tabulated disc needs >= 2 samples on an increasing ρ grid");
    if (cfg->disc_id == GR_DISC_MESH && (!cfg->disc_table || cfg->disc_table_n < 1 || cfg->disc_table_n > (int64_t)1 << 24))
        return fail(GR_ERR_INVALID_ARGUMENT, "a mesh geometry needs its bounding box and 1 .. 2^24 triangles in disc_table");
    if (cfg->chart_table_n < 0 || cfg->chart_table_n == 1 || (cfg->chart_table_n > 1 && (!cfg->chart_table || !(cfg->chart_theta1 > cfg->chart_theta0))))
        return fail(GR_ERR_INVALID_ARGUMENT, "chart table needs >= 2 samples on an increasing θ grid");
    if (cfg->disc_id == GR_DISC_THIN && !(cfg->disc_r_out >= cfg->disc_r_in))
        return fail(GR_ERR_INVALID_ARGUMENT, "disc outer radius below inner radius");
    return GR_OK;
}

int32_t validate_plane(const gr_plane* pl, const gr_range* rg)
{
    if (!pl || !rg) return fail(GR_ERR_INVALID_ARGUMENT, "plane/range is null");
    if (pl->width <= 0 || pl->height <= 0) return fail(GR_ERR_INVALID_ARGUMENT, "image dimensions must be positive");
    // @assert issorted(αlims), rendering.jl:148-149
    if (pl->alpha0 > pl->alpha1) return fail(GR_ERR_INVALID_ARGUMENT, "α limits must be sorted");
    if (pl->beta0 > pl->beta1) return fail(GR_ERR_INVALID_ARGUMENT, "β limits must be sorted");
    if (rg->count < 0 || rg->first < 0 || rg->block <= 0 || rg->stride_blocks <= 0)
        return fail(GR_ERR_INVALID_ARGUMENT, "bad ray range");
    if (rg->count > 0) {
        const int64_t last = rg->count - 1;
        const int64_t b = last / rg->block;
        const int64_t i = rg->first + b * rg->stride_blocks * rg->block + (last - b * rg->block);
        if (i >= pl->width * pl->height) return fail(GR_ERR_INVALID_ARGUMENT, "ray range exceeds the image");
    }
    return GR_OK;
}

// device copies of the tabulated chart and disc profile (cfg.chart_table / cfg.disc_table are host
// pointers); records "PoloidalShapeChart active" in bit 1 and "count windings" in bit 2 of the private copy of
// cfg.upper_hemisphere
// GR_METRIC_TABULATED: the device copy of the caller's table (uploaded when its build id changes), its header into cfg.params
// and its device address into the private copy of cfg.metric_table -- what TabulatedMetric::load reads
int32_t stage_metric_table(gr_ctx* ctx, Params& p, hipStream_t stream)
{
    if (p.cfg.metric_id != GR_METRIC_TABULATED) return GR_OK;
    const double* t = p.cfg.metric_table;
    const size_t bytes = sizeof(double) * (size_t)p.cfg.metric_table_n;
    const double id = t[gr_tab::H_BUILD_ID];
    if (!ctx->d_metric_table || ctx->metric_table_id != id || ctx->metric_table_bytes < bytes) {
        const int32_t arc = tables_acquire(ctx, stream);
        if (arc != GR_OK) return arc;
        ctx->metric_table_id = 0.0;
        const int32_t rc = ensure((void**)&ctx->d_metric_table, &ctx->metric_table_bytes, bytes);
        if (rc != GR_OK) return rc;
        GR_HIP(hipMemcpyAsync(ctx->d_metric_table, t, bytes, hipMemcpyHostToDevice, stream));
        GR_HIP(hipStreamSynchronize(stream));      // the caller's table may be pageable and gone after the call; once per table
        ctx->metric_table_id = id;
    }
    gr_tab::stage_params(t, p.cfg.params);
    p.cfg.metric_table = ctx->d_metric_table;
    return GR_OK;
}

int32_t stage_disc_table(gr_ctx* ctx, Params& p, hipStream_t stream)
{
    {
        const int32_t mrc = stage_metric_table(ctx, p, stream);
        if (mrc != GR_OK) return mrc;
    }
    p.chart_table = nullptr;
    p.cfg.upper_hemisphere = p.cfg.upper_hemisphere ? 1 : 0;
    if (p.cfg.chart_table_n > 1 || p.cfg.disc_id == GR_DISC_TABULATED || p.cfg.disc_id == GR_DISC_MESH) {
        const int32_t arc = tables_acquire(ctx, stream);
        if (arc != GR_OK) return arc;
    }
    if (p.cfg.chart_table_n > 1) {
        const size_t cb = sizeof(double) * (size_t)p.cfg.chart_table_n;
        int32_t crc = ensure((void**)&ctx->d_chart_table, &ctx->chart_table_bytes, cb);
        if (crc != GR_OK) return crc;
        GR_HIP(hipMemcpyAsync(ctx->d_chart_table, p.cfg.chart_table, cb, hipMemcpyHostToDevice, stream));
        p.chart_table = ctx->d_chart_table;
        p.cfg.upper_hemisphere |= 2;
    }
    if (p.cfg.count_windings) p.cfg.upper_hemisphere |= 4;     // TraceWindings: bit 2
    p.disc_table = nullptr;
    if (p.cfg.disc_id == GR_DISC_MESH) {
        // the caller's triangles -> the grid-sorted table the kernels walk; rebuilt only when the mesh changes
        const int64_t n = p.cfg.disc_table_n;
        const uint64_t fp = gr_mesh::fingerprint(p.cfg.disc_table, n);
        if (ctx->mesh_n != n || ctx->mesh_fp != fp || !ctx->d_mesh) {
            if (!gr_mesh::vertices_finite(p.cfg.disc_table, n)) return fail(GR_ERR_INVALID_ARGUMENT, "the mesh has a vertex that is not finite");
            gr_mesh::build_table(p.cfg.disc_table, n, ctx->mesh_host);
            const size_t mb = sizeof(double) * ctx->mesh_host.size();
            ctx->mesh_n = -1;
            const int32_t mrc = ensure((void**)&ctx->d_mesh, &ctx->mesh_bytes, mb);
            if (mrc != GR_OK) return mrc;
            GR_HIP(hipMemcpyAsync(ctx->d_mesh, ctx->mesh_host.data(), mb, hipMemcpyHostToDevice, stream));
            // mesh_host is rebuilt in place by the next mesh: the copy out of it must have finished by then (the device entry
            // points run on a caller's stream that nothing else would wait for).  Once per mesh.
            GR_HIP(hipStreamSynchronize(stream));
            ctx->mesh_n = n;
            ctx->mesh_fp = fp;
        }
        p.disc_table = ctx->d_mesh;
        return GR_OK;
    }
    if (p.cfg.disc_id != GR_DISC_TABULATED) return GR_OK;
    const size_t tb = sizeof(double) * (size_t)p.cfg.disc_table_n;
    int32_t rc = ensure((void**)&ctx->d_disc_table, &ctx->disc_table_bytes, tb);
    if (rc != GR_OK) return rc;
    GR_HIP(hipMemcpyAsync(ctx->d_disc_table, p.cfg.disc_table, tb, hipMemcpyHostToDevice, stream));
    p.disc_table = ctx->d_disc_table;
    return GR_OK;
}

// Longest-processing-time-first order of the 8x8 tiles.  The first render of a plane records the
// step count of one ray per tile; the next render of the SAME (config, plane, range) sorts the
// tiles by that cost, longest first, so the rays started last are the short ones and the tail of
// the launch (queue empty, waves draining) shrinks.  Only the ORDER of the work queue is learned:
// every ray is traced in full every time and results are bit-identical with or without it.
// Which launch shape a launch of n rays gets under kernel = 2 (auto).  Measured on MI355X with
// Launch shape when the caller leaves it to the library (kernel = 2, block = 0).  Measured on MI355X
// (scripts/kernel_shapes.py, bench.py --emulate-shard):
//  * image planes (8x8 pixel tiles per wave): one ray per lane in ONE-WAVE workgroups.  The hardware
//    dispatcher then refills a SIMD the moment a single wave retires, which beats the persistent
//    kernel's wave-ballot refill at every depth (2048²: 23.3 vs 23.8 ms; 1/8 shard: 3.3 vs 3.7 ms;
//    Johannsen 1024²: 8.5 vs 9.6 ms), and 256-thread workgroups of the same kernel by 5-10 %.
//  * line profiles: persistent (every workgroup flushes its LDS histogram at exit; a quarter of a
//    million one-wave workgroups would turn that into 5e7 global atomics: 31 vs 21 ms at 2048² rays).
//  * ray arrays in caller order (no tiles, neighbours may differ wildly in length): persistent with
//    wave-ballot refill, except for launches only a few rays per resident lane deep.
//  * a tabulated metric: the lane kernel whatever the source of the rays.  Refilled lanes take rays from anywhere in the set, a
//    wave's rays then sit in as many patches as it has lanes and most evaluations leave the patch cache for global memory
//    (10⁶ sky rays of a corona: 170 ms against 84; 2²⁰ rays in random order: 179 against 138).
int resolve_kernel(const gr_ctx* ctx, int64_t n, const Cold& cold, int metric_id = -1)
{
    if (ctx->kernel != 2) return (int)ctx->kernel;
    // a tabulated metric: one ray per lane whatever the launch -- a refilled wave's rays sit in more patches than its cache has
    // slots (BinningMethod line profile, 4096² rays at tolerance 1e-5: 69 ms against 158 through the persistent kernel; 203 either
    // way at 1e-9; scripts/sibling_workloads.py tabc5lo)
    if (metric_id == GR_METRIC_TABULATED) return 0;
    if (cold.out_mode == 2) return 1;
    if (cold.src_mode == 0 && cold.swizzle) return 0;
    const int64_t resident_lanes = (int64_t)ctx->n_cu * 8 * 64;
    return n < 6 * resident_lanes ? 0 : 1;
}
int resolve_block(const gr_ctx* ctx, int kernel)
{
    if (ctx->block) return (int)ctx->block;
    return kernel == 0 ? 64 : 256;
}

int32_t lpt_prepare(gr_ctx* ctx, const Params& p, Cold& cold, hipStream_t stream, bool* record)
{
    *record = false;
    cold.tile_perm = nullptr;
    cold.tile_cost = nullptr;
    const int kern = resolve_kernel(ctx, p.n, cold, p.cfg.metric_id);
    // (a tabulated metric's lane kernel orders its tiles longest-first by default: its waves' lifetimes spread 9x around their
    // mean -- the fused kernels' 2.7x -- and the cost it learns is the wave's lifetime, not a step count)
    const bool tab = p.cfg.metric_id == GR_METRIC_TABULATED;
    if (!ctx->lpt || ctx->lpt_suspend || (kern != 1 && !ctx->lpt_lane && !tab) || !cold.swizzle || cold.src_mode != 0) return GR_OK;
    const int64_t tiles = p.n >> 6;
    // Measured on MI355X: longest-first pays when a launch is only a few tiles per
    // resident wave deep (the 1/8 shard of a 2048² image: 4.0 -> 3.7 ms on the rank holding the
    // α≈0 columns, whose rays take up to 430 steps) and costs 2-4 % on deep launches.  lpt = 2 forces it.
    const int64_t resident_waves = (int64_t)ctx->n_cu * 8;
    if (tiles < resident_waves) return GR_OK;
    if (ctx->lpt == 1 && tiles >= (tab ? 24 : 6) * resident_waves) return GR_OK;
    // (a tabulated metric: the table's build id is part of the key -- the config itself holds only a pointer, and two metrics
    // fitted into the same buffer on the same grid would otherwise share one learned order)
    std::vector<unsigned char> key(sizeof(gr_config) + sizeof(gr_plane) + sizeof(gr_range) + sizeof(double));
    const double table_id = tab && p.cfg.metric_table ? p.cfg.metric_table[gr_tab::H_BUILD_ID] : 0.0;
    std::memcpy(key.data() + sizeof(gr_config) + sizeof(gr_plane) + sizeof(gr_range), &table_id, sizeof(double));
    std::memcpy(key.data(), &p.cfg, sizeof(gr_config));
    std::memcpy(key.data() + sizeof(gr_config), &cold.plane, sizeof(gr_plane));
    std::memcpy(key.data() + sizeof(gr_config) + sizeof(gr_plane), &cold.range, sizeof(gr_range));
    if (key != ctx->lpt_key) {
        ctx->lpt_key = key;
        ctx->lpt_have_perm = false;
        ctx->lpt_cost_pending = false;
        if (ctx->lpt_cap < tiles) {
            if (ctx->d_tile_cost) (void)hipFree(ctx->d_tile_cost);
            if (ctx->d_tile_perm) (void)hipFree(ctx->d_tile_perm);
            ctx->d_tile_cost = ctx->d_tile_perm = nullptr;
            ctx->lpt_cap = 0;
            GR_HIP(hipMalloc((void**)&ctx->d_tile_cost, sizeof(uint32_t) * tiles));
            GR_HIP(hipMalloc((void**)&ctx->d_tile_perm, sizeof(uint32_t) * tiles));
            ctx->lpt_cap = tiles;
        }
        ctx->lpt_tiles = tiles;
    }
    if (!ctx->lpt_have_perm && ctx->lpt_cost_pending) {
        // the recording launch has to be complete before its costs can be sorted (one-time)
        GR_HIP(hipEventSynchronize(ctx->ev_cost));
        std::vector<uint32_t> cost((size_t)tiles), perm((size_t)tiles);
        GR_HIP(hipMemcpy(cost.data(), ctx->d_tile_cost, sizeof(uint32_t) * tiles, hipMemcpyDeviceToHost));
        // counting sort, descending cost, stable in tile index
        uint32_t cmax = 0;
        for (uint32_t c : cost) cmax = c > cmax ? c : cmax;
        if (cmax < (1u << 24)) {
            std::vector<uint32_t> start((size_t)cmax + 2, 0);
            for (uint32_t c : cost) start[(size_t)(cmax - c) + 1]++;
            for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
            for (int64_t t = 0; t < tiles; ++t) perm[start[(size_t)(cmax - cost[(size_t)t])]++] = (uint32_t)t;
            GR_HIP(hipMemcpyAsync(ctx->d_tile_perm, perm.data(), sizeof(uint32_t) * tiles, hipMemcpyHostToDevice, stream));
            GR_HIP(hipStreamSynchronize(stream));   // perm is a local vector
            ctx->lpt_have_perm = true;
        }
        ctx->lpt_cost_pending = false;
    }
    if (ctx->lpt_have_perm) {
        cold.tile_perm = ctx->d_tile_perm;
    } else {
        GR_HIP(hipMemsetAsync(ctx->d_tile_cost, 0, sizeof(uint32_t) * tiles, stream));
        cold.tile_cost = ctx->d_tile_cost;
        *record = true;
    }
    return GR_OK;
}

#ifdef GR_WAVE_TIMELINE
unsigned long long* g_debug_timeline = nullptr;
#endif

extern "C++" {
namespace {
// Rays from a source into its sky (gr_rayset.sky_*): sample_position_direction_velocity for a source at one position
// (corona-models.jl:1-33).  Sample number j + 1 -> (θ, ϕ) on the source's sky (samplers.jl:30-44) -> k̂ -> v = Mx (1, k̂)
// (sky_angles_to_velocity, samplers.jl:81-99, the tetrad and the Jacobian folded into Mx on the host).  A kernel of its own
// that writes x (once) and v (32 B per ray) for the trace kernels to read as ray arrays: inside Ray::initial_conditions the
// branch -- libm's double-precision sin / cos / acos / atan / fmod -- cost every persistent kernel 6 % whether taken or not
// (C5 through the persistent kernel 58.2 -> 61.5 ms: SGPR spills in the refill path); 32 MB through HBM at 10⁶ rays cost 10 µs.
struct SkyParams {
    double x_obs[4], Mx[16];
    int64_t n;
    int64_t first, total;      // this launch's rays are samples first + 1 .. first + n of `total` (gr_rayset.sky_first / sky_total)
    const double* rows;        // gr_rayset.sky_rows (device) or null: per-sample position, matrix and lowered source velocity
    int32_t sampler, both, generator, reserved;
    double resolution;
    const double* sky_i;
};
// sample jl of the launch -> its direction on the source's sky (samplers.jl:30-44) and the four-velocity v = Mx (1, k̂)
__device__ __forceinline__ void sky_sample(const SkyParams& p, int64_t jl, double& el, double& az, double v[4], double x[4], double& f)
{
    const double n = (double)p.total;
    const double idx = (double)(p.first + jl + 1);
    const double i = p.generator == 0 ? idx : p.generator == 1 ? idx / n : p.sky_i[jl];
    if (p.sampler == 2) {
        const double ph = 2.0 * ::atan(::sqrt(p.resolution / i));
        const bool even = (::floor(i) == i) && (::fmod(i, 2.0) == 0.0);
        el = (!p.both || even) ? ph : 3.14159265358979323846 - ph;
    } else {
        const double u = i / n;
        el = p.both ? ::acos(1.0 - 2.0 * u) : ::acos(1.0 - u);
    }
    const double az_raw = (p.generator == 0 ? 3.14159265358979323846 * (1.0 + 2.2360679774997896964) : 6.28318530717958647692) * i;
    az = ::fmod(az_raw, 6.28318530717958647692);
    if (az < 0.0) az += 6.28318530717958647692;
    const double se = ::sin(el), ce = ::cos(el), sa = ::sin(az), ca = ::cos(az);
    const double pb[4] = { 1.0, -(se * ca), -(se * sa), -ce };
    f = 1.0;
    if (p.rows) {
        // a sample with a position of its own: its matrix, and the factor that turns the ratio against the static observer
        // (g_tμ v^μ) into the ratio against the source's own velocity (u_μ v^μ)
        const double* row = p.rows + GR_SKY_ROW * jl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q] = row[q];
            v[q] = row[4 + q * 4 + 0] * pb[0] + row[4 + q * 4 + 1] * pb[1] + row[4 + q * 4 + 2] * pb[2] + row[4 + q * 4 + 3] * pb[3];
        }
        double eu = 0.0, et = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { eu += row[20 + q] * v[q]; et += row[24 + q] * v[q]; }
        f = eu / et;
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        x[q] = p.x_obs[q];
        v[q] = p.Mx[q * 4 + 0] * pb[0] + p.Mx[q * 4 + 1] * pb[1] + p.Mx[q * 4 + 2] * pb[2] + p.Mx[q * 4 + 3] * pb[3];
    }
}
// out: x_obs[4], v[n][4] -- and for a source of many positions (SkyParams.rows) x[n][4] and f[n] behind them
__device__ __forceinline__ void sky_store(const SkyParams& p, double* out, int64_t slot, const double v[4], const double x[4], double f)
{
    double* o = out + 4 + 4 * slot;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = v[q];
    if (p.rows) {
        double* ox = out + 4 + 4 * p.n + 4 * slot;
#pragma unroll
        for (int q = 0; q < 4; ++q) ox[q] = x[q];
        out[4 + 8 * p.n + slot] = f;
    }
}
__global__ void __launch_bounds__(256) k_sky_velocities(const SkyParams p, double* out)
{
    const int64_t jl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (jl == 0)
        for (int q = 0; q < 4; ++q) out[q] = p.x_obs[q];
    if (jl >= p.n) return;
    double el, az, v[4], x[4], f;
    sky_sample(p, jl, el, az, v, x, f);
    sky_store(p, out, jl, v, x, f);
}
// rows (g, ρ, t, status) of a source of many positions: g against the sample's own source velocity
__global__ void __launch_bounds__(256) k_sky_scale_g(double* rows, const double* f, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) rows[4 * j] *= f[j];
}
// The same rays DEALT BY WHAT THEY WILL COST (gr_corona_trace: the order of its rows is free -- min / max and integer bins do not
// know it).  How many steps a ray takes is known, to a constant per polar angle of emission, BEFORE it is traced: the integrator
// resolves the ray's passage round the polar axis of the coordinates, and along the flat-space straight line from the source the
// azimuth sweeps Δϕ and ln sin θ runs down to the line's closest angular approach to the axis and back --
//     steps ≈ base(polar angle of emission) + 16 Δϕ + 36 ln(sin θ₀ / sin θ_min)
// follows the oracle's step counts of the lamp-post scene (h = 10, 0.01 rad off the axis: 53 ... 519 steps within one polar angle)
// with a residual of 4 steps rms (scripts/corona_lanes.py, DESIGN_measurements.md §M19).  The rays are counting-sorted by that
// number in classes of 12 steps, the most expensive class first, inside a class by chunk of kSkyChunk consecutive samples (one polar
// angle to 1 %: one base): the 64 rays of a wave then take the same number of steps to a few per cent AND the launch starts with its
// longest waves -- a wave of 500-step rays takes 2 ms alone whenever it starts, and consecutive samples put one into every chunk.
// Three launches: count per (class, chunk), one-workgroup exclusive scan, scatter.
// A source with a position per sample (SkyParams.rows: DiscCorona) has no "one base per chunk": its base depends on where the sample
// sits -- oracle step counts of a disc corona (r = 10, h = 5): +90 steps for positions next to the axis, -20 far from it.  Such rays
// are sorted by (class, bucket of |sin θ| of the POSITION, chunk): eight buckets, the one next to the axis first.  Lane utilisation of
// 64-ray waves by the oracle's counts, four chunks: 0.17 ... 0.56 in sample order, 0.21 ... 0.71 by class alone, 0.40 ... 0.83 with
// the position bucket (DESIGN_measurements.md §M19).
constexpr int kSkyChunk = 4096;
constexpr int kSkyClasses = 32;
constexpr int kSkyPosBuckets = 8;
__device__ __forceinline__ int sky_cost_class(const double x[4], const double v[4], int nb, int& pos_bucket)
{
    const double r = x[1], s = ::sin(x[2]), c = ::cos(x[2]);
    {
        const int b = (int)(::fabs(s) * (double)nb);
        pos_bucket = b < 0 ? 0 : b >= nb ? nb - 1 : b;
    }
    double er = v[1], et = r * v[2], ep = r * s * v[3];
    const double nrm = ::sqrt(er * er + et * et + ep * ep);
    if (!(nrm > 0.0) || !(nrm < 1e300)) return 0;
    er /= nrm; et /= nrm; ep /= nrm;
    // the source in the x-z plane, the ray's direction in Cartesian components
    const double Px = r * s, Pz = r * c;
    const double dx = er * s + et * c, dy = ep, dz = er * c - et * s;
    const double dphi = ::atan2(::fabs(dy), dx);
    // the directions origin -> points of the line run along a great circle from P̂ to d̂ (normal n = P x d): sin θ on it has an
    // extremum |n_z| / |n|, reached on the way iff cos θ moves towards that pole at the start and away from it at the end
    const double Pd = Px * dx + Pz * dz;
    const double nx = -Pz * dy, ny = Pz * dx - Px * dz, nz = Px * dy;
    const double nn = ::sqrt(nx * nx + ny * ny + nz * nz);
    const double a0 = dz * r * r - Pz * Pd, a1 = Pz - Pd * dz;
    double tv = 0.0;
    if (((a0 > 0.0 && a1 > 0.0) || (a0 < 0.0 && a1 < 0.0)) && nn > 0.0) {
        const double s_ext = ::fabs(nz) / nn, s0 = ::fabs(s);
        if (s_ext < s0) tv = ::log(s0 / (s_ext > 1e-12 ? s_ext : 1e-12));
    }
    const double cost = 16.0 * dphi + 36.0 * tv;
    const int cls = (int)(cost * (1.0 / 12.0));
    return cls < 0 ? 0 : cls >= kSkyClasses ? kSkyClasses - 1 : cls;      // (NaN -> 0)
}
// counts / offsets: [kSkyClasses - 1 - class][position bucket][chunks - 1 - chunk] -- the order of the dealt array (the last samples
// of a sky leave upwards, away from the disc, and have the larger base: first within their class and bucket)
template <bool COUNT>
__global__ void __launch_bounds__(1024) k_sky_velocities_dealt(const SkyParams p, double* out, unsigned* table)
{
    __shared__ int cnt[kSkyClasses * kSkyPosBuckets];
    const int nb = p.rows ? kSkyPosBuckets : 1;
    const int64_t base = (int64_t)blockIdx.x * kSkyChunk;
    const int64_t left = p.n - base;
    const int nloc = left < kSkyChunk ? (int)left : kSkyChunk;
    if (threadIdx.x < kSkyClasses * kSkyPosBuckets) cnt[threadIdx.x] = 0;
    if (!COUNT && blockIdx.x == 0 && threadIdx.x == 0)
        for (int q = 0; q < 4; ++q) out[q] = p.x_obs[q];
    __syncthreads();
    double v[kSkyChunk / 1024][4], x[kSkyChunk / 1024][4], f[kSkyChunk / 1024];
    int cls[kSkyChunk / 1024], pos[kSkyChunk / 1024];
#pragma unroll
    for (int k = 0; k < kSkyChunk / 1024; ++k) {
        const int j = (int)threadIdx.x + 1024 * k;
        cls[k] = -1;
        if (j < nloc) {
            double el, az;
            sky_sample(p, base + j, el, az, v[k], x[k], f[k]);
            int pb;
            const int c = sky_cost_class(x[k], v[k], nb, pb);
            cls[k] = (kSkyClasses - 1 - c) * nb + pb;          // the cell's row in the table: expensive classes first
            pos[k] = atomicAdd(&cnt[cls[k]], 1);
        }
    }
    const unsigned chunks = gridDim.x;
    const unsigned col = chunks - 1u - blockIdx.x;
    if (COUNT) {
        __syncthreads();
        if ((int)threadIdx.x < kSkyClasses * nb) table[(size_t)threadIdx.x * chunks + col] = (unsigned)cnt[threadIdx.x];
        return;
    }
#pragma unroll
    for (int k = 0; k < kSkyChunk / 1024; ++k) {
        if (cls[k] < 0) continue;
        sky_store(p, out, (int64_t)table[(size_t)cls[k] * chunks + col] + pos[k], v[k], x[k], f[k]);
    }
}
// in-place exclusive scan of `n` counts by one workgroup (n = 32 x chunks: 7808 for 10⁶ samples)
__global__ void __launch_bounds__(1024) k_sky_scan(unsigned* table, int64_t n)
{
    __shared__ unsigned part[1024];
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    unsigned sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += table[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
    for (int64_t i = lo; i < hi; ++i) { const unsigned c = table[i]; table[i] = run; run += c; }
}
}  // namespace
}  // extern "C++"

// are the rays of this launch's sky source dealt by cost? (offsets are 32-bit)
static bool sky_dealt(const gr_ctx* ctx, int64_t n)
{
    return ctx->sky_any_order && ctx->sky_deal && n >= 4 * kSkyChunk && n < ((int64_t)1 << 32);
}
// a sky source -> ray arrays in the context's sky buffer (ordered against earlier launches that read it like the staged tables)
static int32_t sky_prepare(gr_ctx* ctx, Params& p, Cold& cold, hipStream_t stream)
{
    int32_t rc;
    if ((rc = tables_acquire(ctx, stream)) != GR_OK) return rc;
    if ((rc = ensure((void**)&ctx->d_sky, &ctx->sky_bytes, sizeof(double) * (4 + (ctx->sky_rows ? 9 : 4) * (size_t)p.n))) != GR_OK) return rc;
    SkyParams sp;
    std::memcpy(sp.x_obs, cold.plane.x_obs, sizeof sp.x_obs);
    std::memcpy(sp.Mx, cold.plane.Mx, sizeof sp.Mx);
    sp.n = p.n;
    sp.first = ctx->sky_first;
    sp.total = ctx->sky_total > 0 ? ctx->sky_total : p.n;
    sp.sampler = cold.sky_sampler; sp.both = cold.sky_both; sp.generator = cold.sky_generator; sp.reserved = 0;
    sp.resolution = cold.sky_resolution;
    sp.sky_i = cold.sky_i;
    sp.rows = ctx->sky_rows;
    if (sky_dealt(ctx, p.n)) {
        const unsigned chunks = (unsigned)((p.n + kSkyChunk - 1) / kSkyChunk);
        const int64_t cells = (int64_t)kSkyClasses * (sp.rows ? kSkyPosBuckets : 1) * chunks;
        if ((rc = ensure((void**)&ctx->d_sky_table, &ctx->sky_table_bytes, sizeof(unsigned) * (size_t)cells)) != GR_OK) return rc;
        hipLaunchKernelGGL(k_sky_velocities_dealt<true>, dim3(chunks), dim3(1024), 0, stream, sp, ctx->d_sky, ctx->d_sky_table);
        hipLaunchKernelGGL(k_sky_scan, dim3(1), dim3(1024), 0, stream, ctx->d_sky_table, cells);
        hipLaunchKernelGGL(k_sky_velocities_dealt<false>, dim3(chunks), dim3(1024), 0, stream, sp, ctx->d_sky, ctx->d_sky_table);
    } else
        hipLaunchKernelGGL(k_sky_velocities, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, stream, sp, ctx->d_sky);
    GR_HIP(hipGetLastError());
    cold.src_mode = 1;
    cold.x = ctx->sky_rows ? ctx->d_sky + 4 + 4 * p.n : ctx->d_sky;
    cold.x_stride = ctx->sky_rows ? 4 : 0;
    cold.v = ctx->d_sky + 4;
    cold.sky_i = nullptr;
    return GR_OK;
}

int32_t launch_trace(gr_ctx* ctx, Params& p, const Cold& cold_in, hipStream_t stream)
{
    Cold cold = cold_in;
    cold.winding_plane = p.cfg.winding_plane;
    if (p.cfg.metric_id == GR_METRIC_TABULATED && cold_in.src_mode != 1) {
        // (validate_cfg has compared the chart with the table's range; p.cfg.metric_table is still the caller's host table here.)
        // The observer / source position the rays start from must lie inside the table as well.
        const double r_obs = cold_in.plane.x_obs[1], t_min = p.cfg.metric_table[gr_tab::H_RMIN], t_max = p.cfg.metric_table[gr_tab::H_RMAX];
        if (!(r_obs >= t_min - 1e-9 * std::fabs(t_min) - 1e-12) || !(r_obs <= t_max + 1e-9 * std::fabs(t_max)))
            return fail(GR_ERR_INVALID_ARGUMENT, "GR_METRIC_TABULATED: the rays start at r = " + std::to_string(r_obs) + ", outside the radial range of the "
                                                 "metric table [" + std::to_string(t_min) + ", " + std::to_string(t_max) + "]");
    }
    const bool sky = cold.src_mode == 3;
    if (sky && p.n > 0) {
        const int32_t src = sky_prepare(ctx, p, cold, stream);
        if (src != GR_OK) return src;
    }
    bool lpt_record = false;
    if (p.n > 0) {
        const int32_t lrc = lpt_prepare(ctx, p, cold, stream, &lpt_record);
        if (lrc != GR_OK) return lrc;
    }
    if (p.n == 0) return GR_OK;
    {
        const int32_t trc = stage_disc_table(ctx, p, stream);
        if (trc != GR_OK) return trc;
    }
    // stage the cold block into the next ring slot (stream-ordered before the kernel)
    Cold* slot = ctx->d_cold + ctx->cold_next;
    ctx->cold_next = (ctx->cold_next + 1) % ctx->queue_slots;
    GR_HIP(hipMemcpyAsync(slot, &cold, sizeof(Cold), hipMemcpyHostToDevice, stream));
    p.cold = slot;
    // (fp32 line profile, 4096² rays: 21.2 ms at 32 against 22.3 at 16 and 24.2 at 8; fp64: 50.8 either way -- profiles/r5n_c5f32_knobs.log)
    p.refill_threshold = (int32_t)(ctx->refill_threshold ? ctx->refill_threshold : (ctx->precision == 32 ? 32 : 16));
    // the tangent objects carry the one-ray-per-lane kernel only: settle kernel and block BEFORE anything is sized by them
    const bool tangent = cold.out_mode == 5;
    if (p.cfg.disc_id == GR_DISC_MESH && (tangent || ctx->precision == 32))
        return fail(GR_ERR_UNSUPPORTED, "a mesh geometry is traced by the fp64 kernels only (not with \"precision\" 32, not by the tangent entry points)");
    // A sky source whose rays were dealt by predicted cost (sky_prepare): a wave's 64 rays take nearly the same number of steps
    // (lane utilisation 0.93 against 0.5 for consecutive samples) and the array begins with the longest, which the persistent
    // kernel's refill would mix again -- one ray per lane, four waves to a workgroup (10⁶ lamp-post samples: 5.9 ms against 7.6
    // persistent, 6.6 with one-wave workgroups, 8.4 in sample order; profiles/r6_corona_cost_ab.log)
    const bool dealt = sky && sky_dealt(ctx, p.n) && !tangent;
    const int kern_sel = tangent ? 0 : (dealt && ctx->kernel == 2) ? 0 : resolve_kernel(ctx, p.n, cold, p.cfg.metric_id);
    const int block_sel = tangent ? (ctx->block ? (int)ctx->block : 64) : (dealt && ctx->kernel == 2 && !ctx->block) ? 256 : resolve_block(ctx, kern_sel);
    // LDS staging: the plunging table (<= 2048 rows = 64 KB) and the line-profile histogram (<= 4096 bins).
    // A table is staged per workgroup: with one-wave workgroups a CU holds 8 copies, so it is staged
    // only while those fit the 160 KB of LDS without capping the occupancy (<= 640 rows of 32 B).
    // The table is read at finalize only: on the Johannsen 1024² render (605 rows) staged and
    // L2-served lookups are within run-to-run noise of each other (8.5-8.7 ms).
    const int64_t lds_rows_max = block_sel >= 256 ? 2048 : 640;
    p.lds_plunge_rows = (ctx->lds && cold_in.pf.pf_id == GR_PF_REDSHIFT && cold_in.out_mode != 1
                         && cold_in.pf.n_plunge > 0
                         && cold_in.pf.n_plunge <= lds_rows_max) ? (int32_t)cold_in.pf.n_plunge : 0;
    p.lds_bins = (ctx->lds && cold_in.out_mode == 2 && cold_in.lp_nbins <= 4096) ? (int32_t)cold_in.lp_nbins : 0;
    p.lds_points = (ctx->lds_points && cold.out_mode == 1 && kern_sel == 0) ? 1 : 0;
    // rays in caller order on the one-ray-per-lane kernel: deal the chunks of 64 rays over the XCDs (gr_kernels.hpp, xcd_chunk);
    // image planes are tiled and their tile order already mixes (gr_ctx_set "xcd_spread" 0 switches it off)
    p.xcd_spread = (ctx->xcd_spread && kern_sel == 0 && cold.src_mode != 0) ? 1 : 0;
    derive_params(p);
    LaunchKnobs knobs{ kern_sel, block_sel, ctx->n_cu, (int)ctx->waves_per_simd,
                       ctx->d_queue + ctx->queue_next };
    ctx->queue_next = (ctx->queue_next + 1) % ctx->queue_slots;
    // validate_cfg() has pinned metric_id to [GR_METRIC_KERR, GR_METRIC_NOZ]
    // Tangent launches come in two shapes (kernels_tu.hip): a pair of lanes per ray has the shorter step (latency), one lane
    // per ray does less work in all (throughput).  A launch whose pairs fit the machine at one wave per SIMD -- 2 n / 64 waves
    // on 4 SIMDs per CU -- cannot keep the SIMDs busy either way and is as long as its longest ray: it takes the pairs.
    const bool pairs = tangent && (ctx->tangent_pairs == 1 || (ctx->tangent_pairs == 2 && 2 * p.n <= (int64_t)ctx->n_cu * 4 * 64));
    const trace_fn fn = (tangent ? (pairs ? kTraceTan1 : kTraceTan) : ctx->precision == 32 ? kTrace32 : kTrace64)[p.cfg.metric_id];
    p.tangent_norm = (tangent && ctx->tangent_norm) ? 1 : 0;
#ifdef GR_WAVE_TIMELINE
    p.queue = g_debug_timeline;      // debug builds: 4 x u64 per wave of a one-ray-per-lane launch (gr_kernels.hpp)
#endif
    const hipError_t le = fn(knobs.kernel, knobs.block, knobs.n_cu, knobs.waves_per_simd, knobs.queue, &p, stream);
    if (le != hipSuccess) return fail(GR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
    if (stream == ctx->stream) GR_HIP(hipEventRecord(ctx->ev_k, stream));      // host variants: where the kernel ends
    if (p.disc_table || p.chart_table || cold.pf.n_plunge > 0 || p.cfg.metric_id == GR_METRIC_TABULATED || sky) {
        const int32_t trc = tables_release(ctx, stream);
        if (trc != GR_OK) return trc;
    }
    if (lpt_record) {
        GR_HIP(hipEventRecord(ctx->ev_cost, stream));
        ctx->lpt_cost_pending = true;
    }
    return GR_OK;
}

// copy the plunging table (host pointers) into the context and fill the device-side pf
int32_t stage_pf(gr_ctx* ctx, const gr_config* cfg, const gr_pointfunction* pf, PfDev& out, hipStream_t stream)
{
    if (!pf) return fail(GR_ERR_INVALID_ARGUMENT, "point function is null");
    if (pf->pf_id < GR_PF_AFFINE_TIME || pf->pf_id > GR_PF_WINDING)
        return fail(GR_ERR_UNSUPPORTED, "unknown pf_id " + std::to_string(pf->pf_id));
    if (pf->filter_id < GR_FILTER_NONE || pf->filter_id > GR_FILTER_INTERSECTED)
        return fail(GR_ERR_UNSUPPORTED, "unknown filter_id " + std::to_string(pf->filter_id));
    out.pf_id = pf->pf_id;
    out.filter_id = pf->filter_id;
    out.fill = pf->fill;
    out.r_isco = pf->r_isco;
    out.n_plunge = 0;
    out.plunge_r = out.plunge_vt = out.plunge_vr = out.plunge_vphi = nullptr;
    out.has_u_src = pf->has_u_src ? 1 : 0;
    out.pad_u = 0;
    for (int q = 0; q < 4; ++q) out.u_src[q] = pf->has_u_src ? pf->u_src[q] : (q == 0 ? 1.0 : 0.0);
    if (pf->pf_id == GR_PF_REDSHIFT && pf->n_plunge > 0) {
        if (pf->n_plunge < 2 || !pf->plunge_r || !pf->plunge_vt || !pf->plunge_vr || !pf->plunge_vphi)
            return fail(GR_ERR_INVALID_ARGUMENT, "plunging table needs >= 2 rows and four arrays");
        const int64_t n = pf->n_plunge;
        {
            const int32_t arc = tables_acquire(ctx, stream);
            if (arc != GR_OK) return arc;
        }
        if (ctx->plunge_cap < n) {
            // a smaller table may still be read by a launch in flight on this very stream: hipFree waits for the device
            if (ctx->d_plunge) (void)hipFree(ctx->d_plunge);
            ctx->d_plunge = nullptr;
            ctx->plunge_cap = 0;
            GR_HIP(hipMalloc((void**)&ctx->d_plunge, sizeof(double) * 4 * n));
            ctx->plunge_cap = n;
        }
        const double* src[4] = { pf->plunge_r, pf->plunge_vt, pf->plunge_vr, pf->plunge_vphi };
        for (int q = 0; q < 4; ++q)
            GR_HIP(hipMemcpyAsync(ctx->d_plunge + q * ctx->plunge_cap, src[q], sizeof(double) * n, hipMemcpyHostToDevice, stream));
        out.n_plunge = n;
        out.plunge_r = ctx->d_plunge;
        out.plunge_vt = ctx->d_plunge + ctx->plunge_cap;
        out.plunge_vr = ctx->d_plunge + 2 * ctx->plunge_cap;
        out.plunge_vphi = ctx->d_plunge + 3 * ctx->plunge_cap;
    }
    if (pf->pf_id == GR_PF_REDSHIFT && !(pf->r_isco > 0.0))
        return fail(GR_ERR_INVALID_ARGUMENT, "redshift needs r_isco > 0");
    // n_plunge = 0 selects the analytic Cunningham plunge, which exists for Kerr only (redshift.jl:93-164); every
    // other metric interpolates a tabulated plunge inside its ISCO (redshift.jl:246-276) and must bring the table
    if (pf->pf_id == GR_PF_REDSHIFT && cfg->metric_id != GR_METRIC_KERR && pf->n_plunge < 2)
        return fail(GR_ERR_INVALID_ARGUMENT, "redshift of a non-Kerr metric needs a plunging table (n_plunge >= 2)");
    return GR_OK;
}

void plane_params(gr_ctx* ctx, Params& pp, Cold& p, const gr_config* cfg, const gr_plane* plane, const gr_range* range)
{
    std::memset(&pp, 0, sizeof pp);
    std::memset(&p, 0, sizeof p);
    pp.cfg = *cfg;
    pp.n = range->count;
    p.src_mode = 0;
    p.plane = *plane;
    p.range = *range;
    // tiles of R rows x 64/R columns need whole, column-aligned groups of 64/R columns in the local index space
    const int64_t H = plane->height;
    auto tiles_ok = [&](int64_t rows) {
        const int64_t cols = 64 / rows;
        return (H % rows == 0) && (range->first % H == 0) && (range->block % (cols * H) == 0) && (range->count % (cols * H) == 0);
    };
    p.swizzle = 0;
    if (ctx->swizzle) {
        if (ctx->tile_rows == 16 && tiles_ok(16)) p.swizzle = 4;
        else if (tiles_ok(8)) p.swizzle = 3;
    }
    const int64_t lim = (int64_t)1 << 31;
    p.idx32 = (plane->width * plane->height < lim && range->count < lim && range->block < lim) ? 1 : 0;
}

void stats_to_host(const unsigned long long* h, gr_stats* s)
{
    s->rays = (int64_t)h[0];
    s->accepted_steps = (int64_t)h[1];
    s->rejected_steps = (int64_t)h[2];
    s->rhs_evals = (int64_t)h[3];
    s->flagged_rays = (int64_t)h[4];
    for (int i = 0; i < 4; ++i) s->status_count[i] = (int64_t)h[5 + i];
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int32_t gr_abi_version(void) { return GR_ABI_VERSION; }

#ifdef GR_WAVE_TIMELINE
// debug builds only (not in the header): device buffer of 4 x u64 per wave for the next launches, or NULL
void gr_debug_set_timeline(void* device_buffer) { g_debug_timeline = (unsigned long long*)device_buffer; }
#endif

const char* gr_last_error(void) { return g_last_error.c_str(); }

int32_t gr_ctx_create(int32_t device, gr_ctx** out)
{
    if (!out) return fail(GR_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(GR_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= count)
        return fail(GR_ERR_INVALID_ARGUMENT, "device index out of range");
    GR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GR_HIP(hipGetDeviceProperties(&prop, device));
    gr_ctx* c = new gr_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    int32_t rc = GR_OK;
    do {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(GR_ERR_HIP, "hipStreamCreate failed"); break; }
        if (hipMalloc((void**)&c->d_queue, sizeof(unsigned long long) * c->queue_slots) != hipSuccess) { rc = fail(GR_ERR_OUT_OF_MEMORY, "hipMalloc(queue) failed"); break; }
        if (hipMalloc((void**)&c->d_stats, sizeof(unsigned long long) * N_STAT) != hipSuccess) { rc = fail(GR_ERR_OUT_OF_MEMORY, "hipMalloc(stats) failed"); break; }
        if (hipMalloc((void**)&c->d_cold, sizeof(Cold) * c->queue_slots) != hipSuccess) { rc = fail(GR_ERR_OUT_OF_MEMORY, "hipMalloc(cold) failed"); break; }
        if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev_k) != hipSuccess
            || hipEventCreateWithFlags(&c->ev_cost, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&c->ev_tables, hipEventDisableTiming) != hipSuccess) { rc = fail(GR_ERR_HIP, "hipEventCreate failed"); break; }
    } while (0);
    if (rc != GR_OK) {
        gr_ctx_destroy(c);
        return rc;
    }
    c->counted = true;
    g_live_ctx.fetch_add(1);
    *out = c;
    return GR_OK;
}

static void pool_drop_all();

int32_t gr_ctx_destroy(gr_ctx* c)
{
    if (!c) return GR_OK;
    // the last context takes the pool of page-locked blocks with it (blocks still HELD by the caller stay: their finalizers
    // may run later, gr_host_free needs no context)
    if (c->counted && g_live_ctx.fetch_sub(1) == 1) pool_drop_all();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_cold) (void)hipFree(c->d_cold);
    if (c->d_disc_table) (void)hipFree(c->d_disc_table);
    if (c->d_mesh) (void)hipFree(c->d_mesh);
    if (c->d_chart_table) (void)hipFree(c->d_chart_table);
    if (c->d_metric_table) (void)hipFree(c->d_metric_table);
    if (c->d_corona) (void)hipFree(c->d_corona);
    if (c->d_sky) (void)hipFree(c->d_sky);
    if (c->d_sky_table) (void)hipFree(c->d_sky_table);
    if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
    if (c->d_tile_perm) (void)hipFree(c->d_tile_perm);
    if (c->ev_cost) (void)hipEventDestroy(c->ev_cost);
    if (c->ev_tables) (void)hipEventDestroy(c->ev_tables);
    if (c->d_plunge) (void)hipFree(c->d_plunge);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_k) (void)hipEventDestroy(c->ev_k);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (hipEvent_t e : c->ev_band)
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->band_stream) (void)hipStreamDestroy(c->band_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    delete c;
    return GR_OK;
}

// Pinned result blocks are registered process-wide, not per context: the caller's array (a Julia Vector with a finalizer)
// may outlive the context that allocated it, and finalizers run in no particular order.
//
// How a block is made (round 3, scripts/microbench/host_register_thp.hip on the GPU box, 608 MiB = the end points of a 2048² plane):
//   hipHostMalloc 122-365 ms, hipHostFree 82-97 ms -- page-locking 155 000 4-KiB pages, ten times the call the block serves;
//   mmap + MADV_HUGEPAGE + first touch from 8 threads 11-12 ms, hipHostRegister (Portable | Mapped) of those 304 huge pages
//   1.3 ms, hipHostUnregister 0.0 ms, munmap 28-42 ms; device-to-host copies and the kernel's own stores reach the same
//   56 GB/s either way.
// So blocks of 8 MiB and more are anonymous mappings on transparent huge pages that the library registers (13 ms instead of
// 122-365); smaller ones, and any block for which one of those steps fails, come from hipHostMalloc as before.
// Freed blocks still go to a small process-wide POOL first and the next gr_host_alloc of a similar size takes one from there
// (no cost at all).  Bounded (4 blocks, 4 GiB in all by default: gr_ctx_set(ctx, "pinned_pool_mib", MiB); 0 empties and
// disables it); a pooled block is handed out for requests between half its size and its size.
namespace {
struct PinnedMem {
    void* p;          // what the caller holds
    size_t size;      // usable bytes
    void* map;        // base of the anonymous mapping (null: the block came from hipHostMalloc)
    size_t map_len;
};
std::mutex g_pinned_mutex;
std::vector<PinnedMem> g_pinned;      // blocks handed out
std::vector<PinnedMem> g_pool;        // page-locked blocks waiting for the next gr_host_alloc
size_t g_pool_cap = (size_t)1 << 30;  // freed blocks kept page-locked: 1 GiB (one 2048² end-point block and change); "pinned_pool_mib"
constexpr size_t kPoolBlocks = 4;
constexpr size_t kHugeMin = (size_t)8 << 20;
std::atomic<bool> g_pinned_huge{ true };   // gr_ctx_set(ctx, "pinned_huge", 0): every block from hipHostMalloc (process-wide)
// A garbage-collected caller (Julia, Python) sees a 100-byte wrapper, not the block behind it, and may pile up page-locked
// memory long before a collection runs: the bytes handed out and not yet freed are counted, and gr_host_alloc REFUSES
// (GR_ERR_OUT_OF_MEMORY) a request that would take them past this cap -- the bindings then collect and retry, or fall back to
// an ordinary pageable array.  gr_ctx_set(ctx, "pinned_max_mib", MiB), default 8 GiB.
size_t g_pinned_max = (size_t)8 << 30;
size_t g_pinned_out = 0;              // bytes of g_pinned (under g_pinned_mutex)

void prefault_threads(char* base, size_t bytes);     // below: first touch from up to 8 threads

bool pinned_make(size_t want, PinnedMem& out)
{
    if (g_pinned_huge.load(std::memory_order_relaxed) && want >= kHugeMin) {
        const size_t two = (size_t)2 << 20, len = (want + two - 1) / two * two;
        void* m = mmap(nullptr, len + two, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            char* al = (char*)(((size_t)m + two - 1) / two * two);
            (void)madvise(al, len, MADV_HUGEPAGE);          // refused or unavailable: 4-KiB pages, a slower registration, same result
            prefault_threads(al, len);
            if (hipHostRegister(al, len, hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess) {
                out = PinnedMem{ al, len, m, len + two };
                return true;
            }
            (void)hipGetLastError();
            (void)munmap(m, len + two);
        }
    }
    void* p = nullptr;
    // page-locked and mapped for every device (hipHostMallocPortable): a multi-device render may write into one block
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) return false;
    out = PinnedMem{ p, want, nullptr, 0 };
    return true;
}
hipError_t pinned_release(const PinnedMem& b)
{
    if (!b.map) return hipHostFree(b.p);      // waits for work that still targets the block
    const hipError_t e = hipHostUnregister(b.p);
    (void)munmap(b.map, b.map_len);
    return e;
}

size_t pool_bytes_locked()
{
    size_t t = 0;
    for (const auto& q : g_pool) t += q.size;
    return t;
}
// release pooled blocks until the pool fits `cap` (call with g_pinned_mutex held; trims are rare)
void pool_trim_locked(size_t cap)
{
    while (!g_pool.empty() && (pool_bytes_locked() > cap || g_pool.size() > kPoolBlocks)) {
        (void)pinned_release(g_pool.front());
        g_pool.erase(g_pool.begin());
    }
}
}

static void pool_drop_all()
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    pool_trim_locked(0);
}

int32_t gr_ctx_set(gr_ctx* c, const char* key, int64_t value)
{
    if (!c || !key) return fail(GR_ERR_INVALID_ARGUMENT, "ctx/key is null");
    const std::string k(key);
    if (k == "pipeline") {
        if (value < 0 || value > 8) return fail(GR_ERR_INVALID_ARGUMENT, "pipeline must be 0..8 bands");
        c->pipeline = value;
        return GR_OK;
    }
    if (k == "kernel") {
        if (value < 0 || value > 2) return fail(GR_ERR_INVALID_ARGUMENT, "kernel must be 0 (lane), 1 (persistent) or 2 (auto)");
        c->kernel = value;
    } else if (k == "block") {
        if (value != 0 && (value < 64 || value > 256 || value % 64)) return fail(GR_ERR_INVALID_ARGUMENT, "block must be 0 (auto) or a multiple of 64 in [64, 256]");
        c->block = value;
    } else if (k == "refill_threshold") {
        if (value < 0 || value > 64) return fail(GR_ERR_INVALID_ARGUMENT, "refill_threshold must be in [0, 64] (0 = auto)");
        c->refill_threshold = value;
    } else if (k == "waves_per_simd") {
        if (value < 0 || value > 8) return fail(GR_ERR_INVALID_ARGUMENT, "waves_per_simd must be in [0, 8]");
        c->waves_per_simd = value;
    } else if (k == "swizzle") {
        c->swizzle = value ? 1 : 0;
    } else if (k == "tile_rows") {
        if (value != 8 && value != 16) return fail(GR_ERR_INVALID_ARGUMENT, "tile_rows must be 8 or 16");
        c->tile_rows = value;
        c->lpt_key.clear();
    } else if (k == "lpt_lane") {
        c->lpt_lane = value ? 1 : 0;
        c->lpt_key.clear();
    } else if (k == "lds") {
        c->lds = value ? 1 : 0;
    } else if (k == "precision") {
        if (value != 32 && value != 64) return fail(GR_ERR_INVALID_ARGUMENT, "precision must be 32 or 64");
        c->precision = value;
    } else if (k == "lpt") {
        if (value < 0 || value > 2) return fail(GR_ERR_INVALID_ARGUMENT, "lpt must be 0 (off), 1 (auto) or 2 (always)");
        c->lpt = value;
        c->lpt_key.clear();
    } else if (k == "hugepages") {
        c->hugepages = value ? 1 : 0;
    } else if (k == "tangent_norm") {
        c->tangent_norm = value ? 1 : 0;
    } else if (k == "tangent_pairs") {
        if (value < 0 || value > 2) return fail(GR_ERR_INVALID_ARGUMENT, "tangent_pairs must be 0 (one lane per ray), 1 (a pair of lanes per ray) or 2 (by launch size)");
        c->tangent_pairs = value;
    } else if (k == "xcd_spread") {
        c->xcd_spread = value ? 1 : 0;
    } else if (k == "sky_deal") {
        c->sky_deal = value ? 1 : 0;
    } else if (k == "lds_points") {
        c->lds_points = value ? 1 : 0;
    } else if (k == "direct_host") {
        c->direct_host = value ? 1 : 0;
    } else if (k == "pinned_huge") {
        g_pinned_huge.store(value != 0);
    } else if (k == "pinned_max_mib") {
        if (value < 0) return fail(GR_ERR_INVALID_ARGUMENT, "pinned_max_mib must be non-negative");
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        g_pinned_max = (size_t)value << 20;        // process-wide: bytes of gr_host_alloc blocks that may be outstanding at once
    } else if (k == "pinned_pool_mib") {
        if (value < 0) return fail(GR_ERR_INVALID_ARGUMENT, "pinned_pool_mib must be non-negative");
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        g_pool_cap = (size_t)value << 20;          // process-wide, like the blocks themselves
        pool_trim_locked(g_pool_cap);
    } else {
        return fail(GR_ERR_INVALID_ARGUMENT, "unknown knob '" + k + "'");
    }
    return GR_OK;
}

int32_t gr_host_alloc(gr_ctx* ctx, int64_t bytes, void** out)
{
    if (!ctx || !out) return fail(GR_ERR_INVALID_ARGUMENT, "ctx/out is null");
    *out = nullptr;
    if (bytes < 0) return fail(GR_ERR_INVALID_ARGUMENT, "bytes must be non-negative");
    const size_t want = (size_t)(bytes > 0 ? bytes : 1);
    try {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        if (g_pinned_out + want > g_pinned_max)
            return fail(GR_ERR_OUT_OF_MEMORY, "gr_host_alloc: " + std::to_string((g_pinned_out + want) >> 20) + " MiB of page-locked results would be "
                        "outstanding (cap " + std::to_string(g_pinned_max >> 20) + " MiB, gr_ctx_set \"pinned_max_mib\"): free or finalize earlier results");
        // best fit among the pooled blocks that are large enough and not more than twice as large
        auto best = g_pool.end();
        for (auto it = g_pool.begin(); it != g_pool.end(); ++it)
            if (it->size >= want && it->size / 2 <= want && (best == g_pool.end() || it->size < best->size)) best = it;
        if (best != g_pool.end()) {
            g_pinned.push_back(*best);
            g_pinned_out += best->size;
            *out = best->p;
            g_pool.erase(best);
            return GR_OK;
        }
    } catch (...) {
        return fail(GR_ERR_OUT_OF_MEMORY, "gr_host_alloc: registry");
    }
    GR_HIP(hipSetDevice(ctx->device));
    PinnedMem b{};
    if (!pinned_make(want, b)) {
        const hipError_t e = hipGetLastError();
        return fail(GR_ERR_OUT_OF_MEMORY, std::string("gr_host_alloc: ") + hipGetErrorString(e));
    }
    try {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        g_pinned.push_back(b);
        g_pinned_out += b.size;
    } catch (...) {
        (void)pinned_release(b);
        return fail(GR_ERR_OUT_OF_MEMORY, "gr_host_alloc: registry");
    }
    *out = b.p;
    return GR_OK;
}

int32_t gr_host_free(gr_ctx* /* may be NULL or already destroyed: not dereferenced */, void* p)
{
    if (!p) return GR_OK;
    PinnedMem b{};
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        auto it = std::find_if(g_pinned.begin(), g_pinned.end(), [p](const PinnedMem& q) { return q.p == p; });
        if (it == g_pinned.end()) return fail(GR_ERR_INVALID_ARGUMENT, "pointer was not allocated by gr_host_alloc");
        b = *it;
        g_pinned.erase(it);
        g_pinned_out -= b.size <= g_pinned_out ? b.size : g_pinned_out;
        // every entry point that writes a block is blocking, so nothing targets it any more: it can wait for the next request
        if (b.size <= g_pool_cap && g_live_ctx.load() > 0) {      // no context left: nobody to hand the block to
            // the newest block is the likeliest to be asked for again: older ones make room (least recently freed first)
            while (!g_pool.empty() && (g_pool.size() >= kPoolBlocks || pool_bytes_locked() + b.size > g_pool_cap)) {
                (void)pinned_release(g_pool.front());
                g_pool.erase(g_pool.begin());
            }
            try {
                g_pool.push_back(b);
                return GR_OK;
            } catch (...) {
            }
        }
    }
    GR_HIP(pinned_release(b));
    return GR_OK;
}

// Do the `bytes` bytes at p lie inside ONE block the library pinned itself?  (No page faults to prepare, no huge-page advice to
// give -- and the only memory a kernel may store into across the link: the whole extent is checked, a pointer near the end of a
// block or a pooled block handed out for a smaller request must not send the GPU past the registered mapping.)
static bool is_pinned(const void* p, size_t bytes)
{
    if (!p) return false;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (const auto& q : g_pinned)
        if ((const char*)p >= (const char*)q.p && (const char*)p + (bytes ? bytes : 1) <= (const char*)q.p + q.size) return true;
    return false;
}

// out_global: pixel / record of local ray j goes to index range_map(j) of d_image / d_points (the whole plane's buffer, which
// several devices fill together) instead of to index j of a buffer holding this range only
static int32_t render_device_impl(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_pointfunction* pf,
                                  const gr_range* range, double* d_image, gr_stats* d_stats, void* hip_stream, bool out_global)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if ((rc = validate_plane(plane, range)) != GR_OK) return rc;
    if (!d_image && range->count > 0) return fail(GR_ERR_INVALID_ARGUMENT, "image is null");
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    plane_params(ctx, p, cd, cfg, plane, range);
    if ((rc = stage_pf(ctx, cfg, pf, cd.pf, stream)) != GR_OK) return rc;
    cd.out_mode = 0;
    cd.image = d_image;
    cd.out_global = out_global ? 1 : 0;
    p.stats = (unsigned long long*)d_stats;   // same layout: 9 x 64-bit counters then kernel_ms
    return launch_trace(ctx, p, cd, stream);
}

int32_t gr_render_device(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_pointfunction* pf,
                         const gr_range* range, double* d_image, gr_stats* d_stats, void* hip_stream)
{
    return render_device_impl(ctx, cfg, plane, pf, range, d_image, d_stats, hip_stream, false);
}

static int32_t render_endpoints_device_impl(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_range* range,
                                            gr_point* d_points, gr_stats* d_stats, void* hip_stream, bool out_global)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if ((rc = validate_plane(plane, range)) != GR_OK) return rc;
    if (!d_points && range->count > 0) return fail(GR_ERR_INVALID_ARGUMENT, "points is null");
    GR_HIP(hipSetDevice(ctx->device));
    Params p;
    Cold cd;
    plane_params(ctx, p, cd, cfg, plane, range);
    cd.out_mode = 1;
    cd.points = d_points;
    cd.out_global = out_global ? 1 : 0;
    p.stats = (unsigned long long*)d_stats;
    return launch_trace(ctx, p, cd, (hipStream_t)hip_stream);
}

int32_t gr_render_endpoints_device(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_range* range,
                                   gr_point* d_points, gr_stats* d_stats, void* hip_stream)
{
    return render_endpoints_device_impl(ctx, cfg, plane, range, d_points, d_stats, hip_stream, false);
}

int32_t gr_trace_endpoints_device(gr_ctx* ctx, const gr_config* cfg, const double* d_x, int64_t x_stride,
                                  const double* d_v, int64_t n, gr_point* d_points, gr_stats* d_stats, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (x_stride != 0 && x_stride != 4) return fail(GR_ERR_INVALID_ARGUMENT, "x_stride must be 0 or 4");
    if (n > 0 && (!d_x || !d_v || !d_points)) return fail(GR_ERR_INVALID_ARGUMENT, "x/v/points is null");
    GR_HIP(hipSetDevice(ctx->device));
    Params p;
    Cold cd;
    std::memset(&p, 0, sizeof p);
    std::memset(&cd, 0, sizeof cd);
    p.cfg = *cfg;
    p.n = n;
    p.stats = (unsigned long long*)d_stats;
    cd.src_mode = 1;
    cd.out_mode = 1;
    cd.x = d_x;
    cd.x_stride = x_stride;
    cd.v = d_v;
    cd.points = d_points;
    cd.range = gr_range{ 0, n, n > 0 ? n : 1, 1 };
    cd.swizzle = 0;
    return launch_trace(ctx, p, cd, (hipStream_t)hip_stream);
}

int32_t gr_apply_pointfunction_device(gr_ctx* ctx, const gr_config* cfg, const gr_pointfunction* pf,
                                      const gr_point* d_points, int64_t n, double max_time, double* d_out, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (n > 0 && (!d_points || !d_out)) return fail(GR_ERR_INVALID_ARGUMENT, "points/out is null");
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    std::memset(&p, 0, sizeof p);
    std::memset(&cd, 0, sizeof cd);
    p.cfg = *cfg;
    p.n = n;
    if ((rc = stage_pf(ctx, cfg, pf, cd.pf, stream)) != GR_OK) return rc;
    if (n == 0) return GR_OK;
    if ((rc = stage_metric_table(ctx, p, stream)) != GR_OK) return rc;
    Cold* slot = ctx->d_cold + ctx->cold_next;
    ctx->cold_next = (ctx->cold_next + 1) % ctx->queue_slots;
    GR_HIP(hipMemcpyAsync(slot, &cd, sizeof(Cold), hipMemcpyHostToDevice, stream));
    p.cold = slot;
    GR_HIP(kApply64[cfg->metric_id](&p, d_points, max_time, d_out, stream));
    GR_HIP(hipGetLastError());
    if ((cd.pf.n_plunge > 0 || cfg->metric_id == GR_METRIC_TABULATED) && (rc = tables_release(ctx, stream)) != GR_OK) return rc;
    return GR_OK;
}

static void prefault_output(void* dst, size_t bytes, bool huge);

int32_t gr_trace_paths(gr_ctx* ctx, const gr_config* cfg, const double* x, int64_t x_stride, const double* v, int64_t n,
                       int64_t cap, double* path, int64_t* n_rows, gr_point* endpoints)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (!x || !v || !path || !n_rows || cap < 2 || n < 0 || (x_stride != 0 && x_stride != 4))
        return fail(GR_ERR_INVALID_ARGUMENT, "x/v/path/n_rows is null, cap < 2, n < 0 or x_stride not 0 / 4");
    if (n == 0) return GR_OK;
    GR_HIP(hipSetDevice(ctx->device));
    const size_t path_bytes = sizeof(double) * 9 * (size_t)cap * (size_t)n;
    const size_t pt_bytes = sizeof(gr_point) * (size_t)n;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, path_bytes + pt_bytes + 8 * (size_t)n + 16)) != GR_OK) return rc;
    const size_t nx = x_stride ? (size_t)n : 1;
    if ((rc = ensure(&ctx->d_in, &ctx->in_bytes, sizeof(double) * 4 * (nx + (size_t)n) + 8)) != GR_OK) return rc;
    double* d_path = (double*)ctx->d_scratch;
    gr_point* d_pt = (gr_point*)((char*)ctx->d_scratch + path_bytes);
    unsigned long long* d_n = (unsigned long long*)((char*)d_pt + pt_bytes);
    double* d_x = (double*)ctx->d_in;
    double* d_v = d_x + 4 * nx;
    GR_HIP(hipMemcpyAsync(d_x, x, sizeof(double) * 4 * nx, hipMemcpyHostToDevice, ctx->stream));
    GR_HIP(hipMemcpyAsync(d_v, v, sizeof(double) * 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    Params p;
    Cold cd;
    std::memset(&p, 0, sizeof p);
    std::memset(&cd, 0, sizeof cd);
    p.cfg = *cfg;
    p.n = n;
    derive_params(p);
    cd.src_mode = 1;
    cd.out_mode = 1;
    cd.winding_plane = p.cfg.winding_plane;
    cd.x = d_x;
    cd.x_stride = x_stride;
    cd.v = d_v;
    cd.points = d_pt;
    cd.range = gr_range{ 0, n, n, 1 };
    if ((rc = stage_disc_table(ctx, p, ctx->stream)) != GR_OK) return rc;
    Cold* slot = ctx->d_cold + ctx->cold_next;
    ctx->cold_next = (ctx->cold_next + 1) % ctx->queue_slots;
    GR_HIP(hipMemcpyAsync(slot, &cd, sizeof(Cold), hipMemcpyHostToDevice, ctx->stream));
    p.cold = slot;
    GR_HIP(kPath64[cfg->metric_id](&p, d_path, cap, d_n, ctx->stream));
    GR_HIP(hipGetLastError());
    if ((p.disc_table || p.chart_table || cfg->metric_id == GR_METRIC_TABULATED) && (rc = tables_release(ctx, ctx->stream)) != GR_OK) return rc;
    static_assert(sizeof(unsigned long long) == sizeof(int64_t), "row counters are copied as int64");
    if (!is_pinned(path, path_bytes)) prefault_output(path, path_bytes, ctx->hugepages != 0);          // while the kernel runs (the path buffer of 16 384 geodesics is 600 MB)
    GR_HIP(hipMemcpyAsync(n_rows, d_n, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    GR_HIP(hipMemcpyAsync(path, d_path, path_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (endpoints) GR_HIP(hipMemcpyAsync(endpoints, d_pt, pt_bytes, hipMemcpyDeviceToHost, ctx->stream));
    GR_HIP(hipStreamSynchronize(ctx->stream));
    return GR_OK;
}

int32_t gr_trace_path(gr_ctx* ctx, const gr_config* cfg, const double* x, const double* v, int64_t cap, double* path,
                      int64_t* n_rows, gr_point* endpoint)
{
    return gr_trace_paths(ctx, cfg, x, 0, v, 1, cap, path, n_rows, endpoint);
}

static int32_t rays_params(gr_ctx* ctx, Params& p, Cold& cd, const gr_config* cfg, const gr_rayset* rays)
{
    if (!rays) return fail(GR_ERR_INVALID_ARGUMENT, "rayset is null");
    if (rays->n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (rays->sky_sampler) {
        // rays from a source into its sky (gr_rayset.sky_*): nothing per ray crosses but, for a caller's generator, its numbers
        if (rays->sky_sampler < 1 || rays->sky_sampler > 2 || rays->sky_generator < 0 || rays->sky_generator > 2)
            return fail(GR_ERR_INVALID_ARGUMENT, "sky source: unknown sampler / generator");
        if (rays->sky_generator == 2 && rays->n > 0 && !rays->sky_i) return fail(GR_ERR_INVALID_ARGUMENT, "sky source: generator 2 needs sky_i");
        if (rays->sky_sampler == 2 && !(rays->sky_resolution > 0.0)) return fail(GR_ERR_INVALID_ARGUMENT, "sky source: WeierstrassSampler needs a resolution > 0");
        if (rays->sep_r || rays->height) return fail(GR_ERR_INVALID_ARGUMENT, "sky source: separable tables / per-ray heights do not apply");
        if (rays->sky_first < 0 || rays->sky_total < 0 || (rays->sky_total > 0 && rays->sky_first + rays->n > rays->sky_total))
            return fail(GR_ERR_INVALID_ARGUMENT, "sky source: the share sky_first .. sky_first + n runs past sky_total");
        ctx->sky_first = rays->sky_total > 0 ? rays->sky_first : 0;
        ctx->sky_total = rays->sky_total > 0 ? rays->sky_total : rays->n;
        ctx->sky_rows = rays->sky_rows;
        std::memset(&p, 0, sizeof p);
        std::memset(&cd, 0, sizeof cd);
        p.cfg = *cfg;
        p.n = rays->n;
        cd.src_mode = 3;
        std::memcpy(cd.plane.x_obs, rays->x_obs, sizeof cd.plane.x_obs);
        std::memcpy(cd.plane.Mx, rays->Mx, sizeof cd.plane.Mx);
        cd.plane.width = rays->n; cd.plane.height = 1;
        cd.range = gr_range{ 0, rays->n, rays->n > 0 ? rays->n : 1, 1 };
        cd.sky_sampler = rays->sky_sampler; cd.sky_both = rays->sky_both ? 1 : 0; cd.sky_generator = rays->sky_generator;
        cd.sky_resolution = rays->sky_resolution;
        cd.sky_i = rays->sky_generator == 2 ? rays->sky_i : nullptr;
        cd.swizzle = 0;
        return GR_OK;
    }
    if (rays->sep_r) {
        if (!rays->sep_cos || !rays->sep_sin || rays->sep_nr < 1 || rays->sep_nt < 1)
            return fail(GR_ERR_INVALID_ARGUMENT, "separable ray set: tables missing or empty");
        if (rays->sep_nr > (int64_t)1 << 31 || rays->sep_nt > (int64_t)1 << 31 || rays->sep_first < 0 || rays->sep_block < 0
            || (rays->sep_block > 0 && rays->sep_stride < rays->sep_block))
            return fail(GR_ERR_INVALID_ARGUMENT, "separable ray set: bad sep_first / sep_block / sep_stride");
        if (rays->n > 0) {
            const int64_t j = rays->n - 1;
            const int64_t last = rays->sep_block > 0
                                     ? rays->sep_first + (j / rays->sep_block) * rays->sep_stride + j % rays->sep_block
                                     : rays->sep_first + j;
            if (last >= rays->sep_nr * rays->sep_nt)
                return fail(GR_ERR_INVALID_ARGUMENT, "separable ray set: the launch's rays run past sep_nr * sep_nt");
        }
        if (rays->height) return fail(GR_ERR_INVALID_ARGUMENT, "separable ray set: per-ray heights are not supported");
    } else if (rays->n > 0 && (!rays->alpha || !rays->beta)) return fail(GR_ERR_INVALID_ARGUMENT, "alpha/beta is null");
    std::memset(&p, 0, sizeof p);
    std::memset(&cd, 0, sizeof cd);
    p.cfg = *cfg;
    p.n = rays->n;
    cd.src_mode = 2;
    std::memcpy(cd.plane.x_obs, rays->x_obs, sizeof cd.plane.x_obs);
    std::memcpy(cd.plane.Mx, rays->Mx, sizeof cd.plane.Mx);
    cd.plane.width = rays->n; cd.plane.height = 1;
    cd.range = gr_range{ 0, rays->n, rays->n > 0 ? rays->n : 1, 1 };
    cd.alpha = rays->alpha; cd.beta = rays->beta; cd.area = rays->area;
    cd.height = cfg->disc_id == GR_DISC_DATUM ? rays->height : nullptr;
    if (rays->sep_r) {
        cd.sep_r = rays->sep_r; cd.sep_cos = rays->sep_cos; cd.sep_sin = rays->sep_sin;
        cd.sep_nr = rays->sep_nr; cd.sep_nt = rays->sep_nt;
        cd.sep_first = rays->sep_first; cd.sep_block = rays->sep_block; cd.sep_stride = rays->sep_stride;
        const bool tiled = rays->sep_tiled && rays->sep_nr >= 8 && rays->sep_nt >= 8;
        cd.sep_core_rows = tiled ? (rays->sep_nr / 8) * 8 : 0;
        cd.sep_core_cols = tiled ? (rays->sep_nt / 8) * 8 : 0;
        if (!tiled) { cd.sep_core_rows = 0; cd.sep_core_cols = 0; }
        cd.alpha = cd.beta = cd.area = nullptr;
    }
    cd.swizzle = 0;
    (void)ctx;
    return GR_OK;
}

int32_t gr_lineprofile_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                              const gr_binning* b, double* d_flux, gr_stats* d_stats, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (!b || b->n_bins < 1 || !b->bin_edges || !d_flux) return fail(GR_ERR_INVALID_ARGUMENT, "binning/flux is null or empty");
    if (!(b->r_max >= b->r_min)) return fail(GR_ERR_INVALID_ARGUMENT, "maxrₑ below minrₑ");
    if (cfg->disc_id == GR_DISC_NONE) return fail(GR_ERR_INVALID_ARGUMENT, "a line profile needs accretion geometry");
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    if ((rc = rays_params(ctx, p, cd, cfg, rays)) != GR_OK) return rc;
    if (!pf || pf->pf_id != GR_PF_REDSHIFT) return fail(GR_ERR_INVALID_ARGUMENT, "line profiles use the redshift point function");
    if ((rc = stage_pf(ctx, cfg, pf, cd.pf, stream)) != GR_OK) return rc;
    cd.out_mode = 2;
    if (b->eps_n != 0 && (b->eps_n < 2 || !b->eps_r || !b->eps_v))
        return fail(GR_ERR_INVALID_ARGUMENT, "tabulated emissivity: needs eps_n >= 2 radii and values");
    cd.lp_rmin = b->r_min; cd.lp_rmax = b->r_max; cd.lp_q = b->emissivity_index;
    cd.lp_eps_r = b->eps_n >= 2 ? b->eps_r : nullptr; cd.lp_eps_v = b->eps_n >= 2 ? b->eps_v : nullptr;
    cd.lp_eps_n = b->eps_n >= 2 ? b->eps_n : 0;
    cd.lp_nbins = b->n_bins; cd.lp_edges = b->bin_edges; cd.lp_flux = d_flux;
    p.stats = (unsigned long long*)d_stats;
    GR_HIP(hipMemsetAsync(d_flux, 0, sizeof(double) * (size_t)b->n_bins, stream));
    return launch_trace(ctx, p, cd, stream);
}

int32_t gr_redshift_radius_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                                  double r_min, double r_max, double* d_pairs, gr_stats* d_stats, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    if ((rc = rays_params(ctx, p, cd, cfg, rays)) != GR_OK) return rc;
    if (rays->n > 0 && !d_pairs) return fail(GR_ERR_INVALID_ARGUMENT, "pairs is null");
    if (!pf || pf->pf_id != GR_PF_REDSHIFT) return fail(GR_ERR_INVALID_ARGUMENT, "needs the redshift point function");
    if ((rc = stage_pf(ctx, cfg, pf, cd.pf, stream)) != GR_OK) return rc;
    cd.out_mode = 3;
    cd.lp_rmin = r_min; cd.lp_rmax = r_max;
    cd.lp_pairs = d_pairs;
    p.stats = (unsigned long long*)d_stats;
    return launch_trace(ctx, p, cd, stream);
}

int32_t gr_ray_summary_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                              double* d_out, gr_stats* d_stats, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    if ((rc = rays_params(ctx, p, cd, cfg, rays)) != GR_OK) return rc;
    if (rays->n > 0 && !d_out) return fail(GR_ERR_INVALID_ARGUMENT, "out is null");
    if (!pf || pf->pf_id != GR_PF_REDSHIFT) return fail(GR_ERR_INVALID_ARGUMENT, "needs the redshift point function");
    if ((rc = stage_pf(ctx, cfg, pf, cd.pf, stream)) != GR_OK) return rc;
    cd.out_mode = 4;
    cd.lp_rmin = 0.0;
    cd.lp_rmax = INFINITY;
    cd.lp_pairs = d_out;
    p.stats = (unsigned long long*)d_stats;
    const bool per_sample = rays->sky_sampler && rays->sky_rows;
    if (per_sample && pf->has_u_src)
        return fail(GR_ERR_INVALID_ARGUMENT, "a sky source with sky_rows brings a source velocity per ray: pf->has_u_src must be 0");
    if ((rc = launch_trace(ctx, p, cd, stream)) != GR_OK) return rc;
    if (per_sample && rays->n > 0) {
        // g of every row against ITS sample's source velocity (the factors were formed with the rays, sky_prepare)
        hipLaunchKernelGGL(k_sky_scale_g, dim3((unsigned)((rays->n + 255) / 256)), dim3(256), 0, stream, d_out, ctx->d_sky + 4 + 8 * rays->n, rays->n);
        GR_HIP(hipGetLastError());
    }
    return GR_OK;
}

int32_t gr_ray_tangent_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                              double* d_out, gr_stats* d_stats, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (cfg->disc_id == GR_DISC_NONE) return fail(GR_ERR_INVALID_ARGUMENT, "ray tangents are taken where the ray meets the geometry: none given");
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    if ((rc = rays_params(ctx, p, cd, cfg, rays)) != GR_OK) return rc;
    if (rays->n > 0 && !d_out) return fail(GR_ERR_INVALID_ARGUMENT, "out is null");
    if (!pf || pf->pf_id != GR_PF_REDSHIFT) return fail(GR_ERR_INVALID_ARGUMENT, "needs the redshift point function");
    if ((rc = stage_pf(ctx, cfg, pf, cd.pf, stream)) != GR_OK) return rc;
    cd.out_mode = 5;
    cd.lp_rmin = 0.0;
    cd.lp_rmax = INFINITY;
    cd.lp_pairs = d_out;
    p.stats = (unsigned long long*)d_stats;
    return launch_trace(ctx, p, cd, stream);
}

int32_t gr_rayset_endpoints_device(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, gr_point* d_points,
                                   gr_stats* d_stats, void* hip_stream)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    int32_t rc;
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    GR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    Params p;
    Cold cd;
    if ((rc = rays_params(ctx, p, cd, cfg, rays)) != GR_OK) return rc;
    if (rays->n > 0 && !d_points) return fail(GR_ERR_INVALID_ARGUMENT, "points is null");
    cd.out_mode = 1;
    cd.points = d_points;
    p.stats = (unsigned long long*)d_stats;
    return launch_trace(ctx, p, cd, stream);
}

// ---- host-buffer variants: stage through the context, block until done ----

// Touch every page of an output buffer from a few threads (its content is about to be overwritten).  A freshly
// allocated destination -- numpy's `empty`, Julia's `Vector{GeodesicPoint}(undef, n)` -- is not backed by pages yet, and
// taking those faults inside the device-to-host copy costs as much again as the copy (637 MB of end points at 2048²:
// 35 ms into touched memory, 60 ms into fresh memory); taken here they overlap the kernel that is already running.
static void prefault_output(void* dst, size_t bytes, bool huge)
{
    if (bytes < ((size_t)64 << 20)) return;
    if (huge) {   // ask for transparent huge pages on the 2 MB-aligned interior: 300 faults instead of 155 000 for 637 MB
        // (measured on the GPU box, THP mode "madvise": 10.7 ms instead of 27-64 ms with 8 threads; errors are ignored;
        // gr_ctx_set(ctx, "hugepages", 0) leaves the caller's mapping alone)
        const size_t two = (size_t)2 << 20;
        const size_t a0 = ((size_t)dst + two - 1) / two * two, a1 = ((size_t)dst + bytes) / two * two;
        if (a1 > a0) (void)madvise((void*)a0, a1 - a0, MADV_HUGEPAGE);
    }
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : (nt > 8 ? 8 : nt);
    const size_t page = 4096;
    char* base = (char*)dst;
    std::vector<std::thread> th;
    // thread creation can throw (std::system_error: out of threads) and nothing may unwind through the C ABI: pre-faulting
    // is an optimisation, so whatever threads exist do their stripes and the copy faults the rest in itself
    try {
        th.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([=]() {
                // every thread sweeps front to back (pages t, t + nt, ...): the first band's pages are ready first
                for (size_t off = (size_t)t * page; off < bytes; off += (size_t)nt * page) ((volatile char*)base)[off] = 0;
            });
    } catch (...) {
    }
    for (auto& x : th) x.join();
}
namespace {
void prefault_threads(char* base, size_t bytes)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : (nt > 8 ? 8 : nt);
    const size_t page = 4096;
    std::vector<std::thread> th;
    try {       // nothing may unwind through the C ABI: whatever threads exist do their stripes, the registration faults in the rest
        th.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([=]() {
                for (size_t off = (size_t)t * page; off < bytes; off += (size_t)nt * page) ((volatile char*)base)[off] = 0;
            });
    } catch (...) {
    }
    for (auto& x : th) x.join();
}
}
struct BackgroundPrefault {
    static constexpr unsigned kMax = 8;
    std::atomic<size_t> progress[kMax];
    std::vector<std::thread> th;
    unsigned nt = 0;
    size_t bytes = 0;
    void start(void* dst, size_t n, bool huge)
    {
        bytes = n;
        if (n < ((size_t)64 << 20)) return;
        const size_t two = (size_t)2 << 20;
        const size_t a0 = ((size_t)dst + two - 1) / two * two, a1 = ((size_t)dst + n) / two * two;
        if (huge && a1 > a0) (void)madvise((void*)a0, a1 - a0, MADV_HUGEPAGE);
        unsigned want = std::thread::hardware_concurrency();
        want = want == 0 ? 4 : (want > kMax ? kMax : want);
        char* base = (char*)dst;
        const unsigned n_threads = want;
        for (unsigned t = 0; t < kMax; ++t) progress[t].store(0, std::memory_order_relaxed);
        // no exception may cross the C ABI: if a thread cannot be created, the ones that exist sweep their stripes and
        // wait_until() waits for those only (nt counts the threads that really run)
        try {
            th.reserve(want);
            for (unsigned t = 0; t < want; ++t) {
                th.emplace_back([this, base, n, t, n_threads]() {
                    const size_t page = 4096;
                    size_t since = 0;
                    for (size_t off = (size_t)t * page; off < n; off += (size_t)n_threads * page) {
                        ((volatile char*)base)[off] = 0;
                        if (++since == 256) { progress[t].store(off, std::memory_order_release); since = 0; }
                    }
                    progress[t].store(n, std::memory_order_release);
                });
                nt = t + 1;
            }
        } catch (...) {
        }
    }
    void wait_until(size_t off) const
    {
        if (nt == 0) return;
        if (off > bytes) off = bytes;
        for (unsigned t = 0; t < nt; ++t)
            while (progress[t].load(std::memory_order_acquire) < off) std::this_thread::yield();
    }
    void finish()
    {
        for (auto& x : th) x.join();
        th.clear();
        nt = 0;
    }
    ~BackgroundPrefault() { finish(); }
};

static int32_t ensure_copy_stream(gr_ctx* ctx)
{
    if (!ctx->copy_stream) GR_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->band_stream) GR_HIP(hipStreamCreateWithFlags(&ctx->band_stream, hipStreamNonBlocking));
    if (!ctx->ev_fork) GR_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->ev_band)
        if (!e) GR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return GR_OK;
}

// How many bands a result of n records is returned in (1: not worth it), each a multiple of `unit` records.
static int band_count(const gr_ctx* ctx, int64_t n, int64_t unit, bool pinned)
{
    if (ctx->pipeline <= 1 || unit <= 0 || n < ((int64_t)1 << 21)) return 1;
    // into page-locked memory the copies run at the link's rate and need no page faults: more, smaller bands leave a
    // shorter last copy exposed (the only one nothing overlaps)
    int nb = (int)(ctx->pipeline > 8 ? 8 : ctx->pipeline);
    if (pinned && nb < 8) nb = 8;
    while (nb > 1 && (n / nb) < unit) --nb;
    return nb;
}

static int32_t begin_host_call(gr_ctx* ctx, gr_stats* stats)
{
    GR_HIP(hipSetDevice(ctx->device));
    if (stats) GR_HIP(hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long) * N_STAT, ctx->stream));
    GR_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    GR_HIP(hipEventRecord(ctx->ev_k, ctx->stream));      // re-recorded behind every trace kernel of the call (launch_trace)
    return GR_OK;
}

static int32_t end_host_call(gr_ctx* ctx, gr_stats* stats)
{
    GR_HIP(hipSetDevice(ctx->device));
    GR_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    unsigned long long h[N_STAT];
    if (stats) GR_HIP(hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    GR_HIP(hipStreamSynchronize(ctx->stream));
    if (stats) {
        stats_to_host(h, stats);
        float ms = 0.f;
        GR_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev_k));
        stats->kernel_ms = ms;      // start of the call's device work -> end of its last trace kernel (input staging included)
        GR_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        stats->call_ms = ms;        // ... -> end of the last copy back to the caller's buffer
        stats->enqueue_ms = 0.0;    // the *_multi entry points fill it in
    }
    return GR_OK;
}

// would this render of a plane run on the one-ray-per-lane kernel (whose waves store whole runs: 64-byte image segments of an
// 8 x 8 tile, 1216-byte runs of end-point records)?  Only those stores are worth sending across the link directly.
static bool plane_on_lane_kernel(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_range* range, int out_mode)
{
    if (validate_cfg(cfg) != GR_OK || validate_plane(plane, range) != GR_OK) return false;   // the regular path reports it
    Params p;
    Cold cd;
    plane_params(ctx, p, cd, cfg, plane, range);
    cd.out_mode = out_mode;
    return resolve_kernel(ctx, range->count, cd, cfg->metric_id) == 0;
}

int32_t gr_render(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_pointfunction* pf,
                  const gr_range* range, double* image, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!range) return fail(GR_ERR_INVALID_ARGUMENT, "range is null");
    if (!image && range->count > 0) return fail(GR_ERR_INVALID_ARGUMENT, "image is null");
    GR_HIP(hipSetDevice(ctx->device));      // before any allocation: ensure() mallocs on the current device
    int32_t rc;
    const size_t bytes = sizeof(double) * (size_t)(range->count > 0 ? range->count : 0);
    // an image in a block the library pinned is written by the kernel itself (as gr_render_endpoints does): no D2H copy
    if (ctx->direct_host && bytes && is_pinned(image, bytes) && plane_on_lane_kernel(ctx, cfg, plane, range, 0)) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, image, 0) == hipSuccess && dp) {
            if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
            if ((rc = gr_render_device(ctx, cfg, plane, pf, range, (double*)dp,
                                       stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
            return end_host_call(ctx, stats);
        }
        (void)hipGetLastError();
    }
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return rc;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = gr_render_device(ctx, cfg, plane, pf, range, (double*)ctx->d_scratch,
                               stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
    if (bytes) GR_HIP(hipMemcpyAsync(image, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

// ---------------------------------------------------------------------------------------
// ONE host thread, SEVERAL devices (include/gradus_mi355x.h, "*_multi"): enqueue on every context, queue the copies home,
// wait for all.
// ---------------------------------------------------------------------------------------
extern "C++" {
namespace {
double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int32_t validate_ctxs(gr_ctx* const* ctxs, int32_t n)
{
    if (!ctxs || n < 1) return fail(GR_ERR_INVALID_ARGUMENT, "no contexts");
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k]) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
        for (int q = 0; q < k; ++q)
            if (ctxs[q] == ctxs[k]) return fail(GR_ERR_INVALID_ARGUMENT, "the same context twice: every share needs its own stream and staging buffers");
    }
    return GR_OK;
}

// enq(k): stage and launch context k's share (begins with begin_host_call); copy(k): queue its copy home.
// Whatever was started is waited for before the call returns, also on an error: the buffers belong to the caller.
template <class Enqueue, class CopyBack>
int32_t multi_drive(gr_ctx* const* ctxs, int32_t n, gr_stats* stats, Enqueue enq, CopyBack copy)
{
    double t_enq[64];
    int32_t rc = GR_OK;
    int started = 0;
    for (int k = 0; k < n && rc == GR_OK; ++k) {
        const double t0 = now_ms();
        rc = enq(k);
        if (k < 64) t_enq[k] = now_ms() - t0;
        started = k + 1;      // begin_host_call may have queued work even if a later step of enq failed
    }
    for (int k = 0; k < n && rc == GR_OK; ++k) rc = copy(k);
    for (int k = 0; k < started; ++k) {
        gr_stats* st = stats ? &stats[k] : nullptr;
        int32_t e;
        if (rc == GR_OK) {
            e = end_host_call(ctxs[k], st);
        } else {
            // an earlier failure: keep its message, just make sure nothing of this call is still in flight
            const std::string keep = g_last_error;
            (void)hipSetDevice(ctxs[k]->device);
            (void)hipStreamSynchronize(ctxs[k]->stream);
            g_last_error = keep;
            e = GR_OK;
        }
        if (rc == GR_OK) rc = e;
        if (st && rc == GR_OK) st->enqueue_ms = k < 64 ? t_enq[k] : 0.0;
    }
    return rc;
}

// the block-cyclic deal of an image's columns over n contexts: `block` rays per block, `n_blocks` blocks = `count` rays each
int32_t plane_deal(const gr_plane* plane, int32_t n, int64_t block_cols, int64_t* block, int64_t* n_blocks, int64_t* count)
{
    if (!plane) return fail(GR_ERR_INVALID_ARGUMENT, "plane is null");
    const int64_t W = plane->width, H = plane->height;
    if (W <= 0 || H <= 0) return fail(GR_ERR_INVALID_ARGUMENT, "image dimensions must be positive");
    int64_t bc = block_cols > 0 ? block_cols : 8;
    while (bc > 1 && W % (bc * n) != 0) bc /= 2;
    if (W % (bc * n) != 0) return fail(GR_ERR_INVALID_ARGUMENT, "image width cannot be dealt in column blocks over the contexts");
    *block = bc * H;
    *n_blocks = W / (bc * n);
    *count = *n_blocks * *block;
    return GR_OK;
}

// the device-side address of a block the library pinned, as the CURRENT device sees it (blocks are registered portable)
void* pinned_device_pointer(void* host)
{
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, host, 0) == hipSuccess && dp) return dp;
    (void)hipGetLastError();
    return nullptr;
}

// contiguous shares of n_rays rays over n contexts, in multiples of 64 rays (the last may be short or empty)
void contiguous_share(int64_t n_rays, int32_t n, int k, int64_t* off, int64_t* cnt)
{
    int64_t per = (n_rays + n - 1) / n;
    per = (per + 63) / 64 * 64;
    int64_t a = (int64_t)k * per, b = a + per;
    if (a > n_rays) a = n_rays;
    if (b > n_rays) b = n_rays;
    *off = a;
    *cnt = b - a;
}
}  // namespace
}  // extern "C++"

int32_t gr_render_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_plane* plane,
                        const gr_pointfunction* pf, int64_t block_cols, double* image, gr_stats* stats)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    if (!image) return fail(GR_ERR_INVALID_ARGUMENT, "image is null");
    int64_t block, n_blocks, count;
    if ((rc = plane_deal(plane, n, block_cols, &block, &n_blocks, &count)) != GR_OK) return rc;
    const size_t bytes = sizeof(double) * (size_t)count;
    // An image in a block the library pinned: every device's kernel stores its pixels at their final place, across the link
    // (what gr_render does for one device) -- no staging image, no copy, nothing for the host to wait on but the kernels.
    bool direct = is_pinned(image, sizeof(double) * (size_t)(plane->width * plane->height));
    for (int k = 0; k < n && direct; ++k) {
        const gr_range rg{ (int64_t)k * block, count, block, (int64_t)n };
        direct = ctxs[k]->direct_host && plane_on_lane_kernel(ctxs[k], cfg, plane, &rg, 0);
    }
    auto enq = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        int32_t r;
        GR_HIP(hipSetDevice(c->device));      // context k's scratch image must live on device k
        double* dst = nullptr;
        if (direct) dst = (double*)pinned_device_pointer(image);
        const bool global = dst != nullptr;
        if (!global) {
            if ((r = ensure(&c->d_scratch, &c->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return r;
            dst = (double*)c->d_scratch;
        }
        c->multi_direct = global;
        if ((r = begin_host_call(c, stats ? &stats[k] : nullptr)) != GR_OK) return r;
        const gr_range rg{ (int64_t)k * block, count, block, (int64_t)n };
        return render_device_impl(c, cfg, plane, pf, &rg, dst, stats ? (gr_stats*)c->d_stats : nullptr, c->stream, global);
    };
    auto copy = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        if (c->multi_direct || count == 0) return GR_OK;
        GR_HIP(hipSetDevice(c->device));
        // local block b of context k is image block b*n + k: one 2-D copy places all of them
        GR_HIP(hipMemcpy2DAsync(image + (size_t)k * block, sizeof(double) * (size_t)(n * block), c->d_scratch,
                                sizeof(double) * (size_t)block, sizeof(double) * (size_t)block, (size_t)n_blocks,
                                hipMemcpyDeviceToHost, c->stream));
        return GR_OK;
    };
    return multi_drive(ctxs, n, stats, enq, copy);
}

int32_t gr_render_endpoints_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_plane* plane,
                                  int64_t block_cols, gr_point* points, gr_stats* stats)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    if (!points) return fail(GR_ERR_INVALID_ARGUMENT, "points is null");
    int64_t block, n_blocks, count;
    if ((rc = plane_deal(plane, n, block_cols, &block, &n_blocks, &count)) != GR_OK) return rc;
    const size_t bytes = sizeof(gr_point) * (size_t)count;
    const size_t total = sizeof(gr_point) * (size_t)(plane->width * plane->height);
    bool direct = is_pinned(points, total);
    for (int k = 0; k < n && direct; ++k) {
        const gr_range rg{ (int64_t)k * block, count, block, (int64_t)n };
        direct = ctxs[k]->direct_host && ctxs[k]->lds_points && plane_on_lane_kernel(ctxs[k], cfg, plane, &rg, 1);
    }
    const bool pinned = direct || is_pinned(points, total);
    auto enq = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        int32_t r;
        GR_HIP(hipSetDevice(c->device));
        gr_point* dst = nullptr;
        if (direct) dst = (gr_point*)pinned_device_pointer(points);
        const bool global = dst != nullptr;
        if (!global) {
            if ((r = ensure(&c->d_scratch, &c->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return r;
            dst = (gr_point*)c->d_scratch;
        }
        c->multi_direct = global;
        if ((r = begin_host_call(c, stats ? &stats[k] : nullptr)) != GR_OK) return r;
        const gr_range rg{ (int64_t)k * block, count, block, (int64_t)n };
        return render_endpoints_device_impl(c, cfg, plane, &rg, dst, stats ? (gr_stats*)c->d_stats : nullptr, c->stream, global);
    };
    bool faulted = false;
    auto copy = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        if (c->multi_direct || count == 0) return GR_OK;
        // every kernel is running by now: the pages of a fresh pageable destination are faulted in under them, once
        if (!faulted && !pinned) prefault_output(points, total, c->hugepages != 0);
        faulted = true;
        GR_HIP(hipSetDevice(c->device));
        GR_HIP(hipMemcpy2DAsync(points + (size_t)k * block, sizeof(gr_point) * (size_t)(n * block), c->d_scratch,
                                sizeof(gr_point) * (size_t)block, sizeof(gr_point) * (size_t)block, (size_t)n_blocks,
                                hipMemcpyDeviceToHost, c->stream));
        return GR_OK;
    };
    return multi_drive(ctxs, n, stats, enq, copy);
}

int32_t gr_render_endpoints(gr_ctx* ctx, const gr_config* cfg, const gr_plane* plane, const gr_range* range,
                            gr_point* points, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!range) return fail(GR_ERR_INVALID_ARGUMENT, "range is null");
    if (!points && range->count > 0) return fail(GR_ERR_INVALID_ARGUMENT, "points is null");
    GR_HIP(hipSetDevice(ctx->device));      // before any allocation: ensure() mallocs on the current device
    int32_t rc;
    const size_t bytes = sizeof(gr_point) * (size_t)(range->count > 0 ? range->count : 0);
    // A contiguous range of a large plane goes out in bands of whole 8-column tile strips: band k is copied back on a
    // second stream while bands k+1.. are traced (2048²: 20 ms of kernel + 15 ms of copy become 25 ms), and the
    // destination's pages are faulted in by helper threads meanwhile.
    // Into a block the library pinned (gr_host_alloc) the kernel stores the records itself: one launch, no staging copy in
    // HBM, no copy engine.  The wave-transposed stores (points_epilogue) cross the link as full-size packets, 637 MB spread
    // over the 20 ms of the 2048² trace is about half the link's rate, and the call ends when the kernel does.
    if (ctx->direct_host && ctx->lds_points && bytes && is_pinned(points, bytes) && plane_on_lane_kernel(ctx, cfg, plane, range, 1)) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, points, 0) == hipSuccess && dp) {
            if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
            if ((rc = gr_render_endpoints_device(ctx, cfg, plane, range, (gr_point*)dp,
                                                 stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
            return end_host_call(ctx, stats);
        }
        (void)hipGetLastError();
    }
    // only the staged routes need the copy of the records in HBM (637 MB at 2048²)
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return rc;
    const bool contiguous = plane && (range->stride_blocks == 1 || range->count <= range->block);
    const int64_t unit = plane ? 8 * plane->height : 0;
    const int nb = (contiguous && unit > 0 && range->first % unit == 0) ? band_count(ctx, range->count, unit, is_pinned(points, bytes)) : 1;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if (nb <= 1) {
        if ((rc = gr_render_endpoints_device(ctx, cfg, plane, range, (gr_point*)ctx->d_scratch,
                                             stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
        if (!is_pinned(points, bytes)) prefault_output(points, bytes, ctx->hugepages != 0);
        if (bytes) GR_HIP(hipMemcpyAsync(points, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return end_host_call(ctx, stats);
    }
    if ((rc = ensure_copy_stream(ctx)) != GR_OK) return rc;
    // equal bands except the last, which is half a band: its copy is the only one nothing overlaps
    const int64_t per = ((2 * range->count / (2 * nb - 1) + unit - 1) / unit) * unit;
    int64_t j0[9];
    int used = 0;
    for (int64_t j = 0; j < range->count && used < 8; j += per) j0[used++] = j;
    j0[used] = range->count;
    ctx->lpt_suspend = true;
    // Bands alternate between two streams: a launch ends when its slowest wave does, and on ONE stream the next band would
    // only start then (four bands: 22.5 ms of kernels against 19.6 ms for the plane in one launch).  Side by side the next
    // band's waves fill the SIMDs the draining one leaves idle.  The second stream starts behind the call's begin marker
    // (the statistics counters are zeroed on the first).
    hipError_t fe = hipEventRecord(ctx->ev_fork, ctx->stream);
    if (fe == hipSuccess) fe = hipStreamWaitEvent(ctx->band_stream, ctx->ev_fork, 0);
    if (fe != hipSuccess) { ctx->lpt_suspend = false; return fail(GR_ERR_HIP, std::string("band stream fork: ") + hipGetErrorString(fe)); }
    for (int k = 0; k < used && rc == GR_OK; ++k) {
        hipStream_t s = (k & 1) ? ctx->band_stream : ctx->stream;
        const gr_range band{ range->first + j0[k], j0[k + 1] - j0[k], j0[k + 1] - j0[k], 1 };
        rc = gr_render_endpoints_device(ctx, cfg, plane, &band, (gr_point*)ctx->d_scratch + j0[k],
                                        stats ? (gr_stats*)ctx->d_stats : nullptr, s);
        if (rc == GR_OK && hipEventRecord(ctx->ev_band[k], s) != hipSuccess) rc = fail(GR_ERR_HIP, "hipEventRecord failed");
    }
    ctx->lpt_suspend = false;
    // join: everything the second stream did is ordered before the call's end marker on the first
    if (used > 1 && rc == GR_OK) {
        const int last_odd = (used - 1) & 1 ? used - 1 : used - 2;
        if (hipStreamWaitEvent(ctx->stream, ctx->ev_band[last_odd], 0) != hipSuccess) rc = fail(GR_ERR_HIP, "hipStreamWaitEvent failed");
    }
    if (rc == GR_OK && hipEventRecord(ctx->ev_k, ctx->stream) != hipSuccess) rc = fail(GR_ERR_HIP, "hipEventRecord failed");
    if (rc != GR_OK) { (void)hipStreamSynchronize(ctx->band_stream); (void)hipStreamSynchronize(ctx->stream); return rc; }
    BackgroundPrefault pf;
    if (!is_pinned(points, bytes)) pf.start(points, bytes, ctx->hugepages != 0);
    hipError_t ce = hipSuccess;
    const char* what = "";
    for (int k = 0; k < used && ce == hipSuccess; ++k) {
        pf.wait_until(sizeof(gr_point) * (size_t)j0[k + 1]);
        ce = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_band[k], 0);
        what = "hipStreamWaitEvent";
        if (ce != hipSuccess) break;
        ce = hipMemcpyAsync(points + j0[k], (gr_point*)ctx->d_scratch + j0[k], sizeof(gr_point) * (size_t)(j0[k + 1] - j0[k]),
                            hipMemcpyDeviceToHost, ctx->copy_stream);
        what = "hipMemcpyAsync (band)";
    }
    if (ce != hipSuccess) {
        // bands already queued are still writing into the caller's memory: nothing may be in flight when the call returns
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->band_stream);
        (void)hipStreamSynchronize(ctx->stream);
        return fail(GR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(ce));
    }
    {
        const hipError_t se = hipStreamSynchronize(ctx->copy_stream);
        if (se != hipSuccess) {
            (void)hipStreamSynchronize(ctx->band_stream);
            (void)hipStreamSynchronize(ctx->stream);
            return fail(GR_ERR_HIP, std::string("hipStreamSynchronize(copy_stream): ") + hipGetErrorString(se));
        }
    }
    return end_host_call(ctx, stats);
}

int32_t gr_trace_endpoints(gr_ctx* ctx, const gr_config* cfg, const double* x, int64_t x_stride, const double* v,
                           int64_t n, gr_point* points, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (x_stride != 0 && x_stride != 4) return fail(GR_ERR_INVALID_ARGUMENT, "x_stride must be 0 or 4");
    if (n > 0 && (!x || !v || !points)) return fail(GR_ERR_INVALID_ARGUMENT, "x/v/points is null");
    GR_HIP(hipSetDevice(ctx->device));      // before any allocation: ensure() mallocs on the current device
    int32_t rc;
    const size_t nx = (size_t)(x_stride == 0 ? 4 : 4 * n), nv = (size_t)(4 * n);
    const size_t out_bytes = sizeof(gr_point) * (size_t)n;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, out_bytes ? out_bytes : 8)) != GR_OK) return rc;
    if ((rc = ensure(&ctx->d_in, &ctx->in_bytes, sizeof(double) * (nx + nv) + 8)) != GR_OK) return rc;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    double* d_x = (double*)ctx->d_in;
    double* d_v = d_x + nx;
    if (n > 0) {
        GR_HIP(hipMemcpyAsync(d_x, x, sizeof(double) * nx, hipMemcpyHostToDevice, ctx->stream));
        GR_HIP(hipMemcpyAsync(d_v, v, sizeof(double) * nv, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = gr_trace_endpoints_device(ctx, cfg, d_x, x_stride, d_v, n, (gr_point*)ctx->d_scratch,
                                        stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
    if (!is_pinned(points, out_bytes)) prefault_output(points, out_bytes, ctx->hugepages != 0);
    if (out_bytes) GR_HIP(hipMemcpyAsync(points, ctx->d_scratch, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

int32_t gr_apply_pointfunction(gr_ctx* ctx, const gr_config* cfg, const gr_pointfunction* pf, const gr_point* points,
                               int64_t n, double max_time, double* out)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (n > 0 && (!points || !out)) return fail(GR_ERR_INVALID_ARGUMENT, "points/out is null");
    int32_t rc;
    const size_t in_bytes = sizeof(gr_point) * (size_t)n, out_bytes = sizeof(double) * (size_t)n;
    GR_HIP(hipSetDevice(ctx->device));      // before any allocation: ensure() mallocs on the current device
    if ((rc = ensure(&ctx->d_in, &ctx->in_bytes, in_bytes ? in_bytes : 8)) != GR_OK) return rc;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, out_bytes ? out_bytes : 8)) != GR_OK) return rc;
    if (n > 0) GR_HIP(hipMemcpyAsync(ctx->d_in, points, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gr_apply_pointfunction_device(ctx, cfg, pf, (const gr_point*)ctx->d_in, n, max_time,
                                            (double*)ctx->d_scratch, ctx->stream)) != GR_OK) return rc;
    if (n > 0) GR_HIP(hipMemcpyAsync(out, ctx->d_scratch, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    GR_HIP(hipStreamSynchronize(ctx->stream));
    return GR_OK;
}

// stage a host rayset on the device: alpha | beta | area | height contiguous in ctx->d_in
static int32_t stage_rays(gr_ctx* ctx, const gr_rayset* rays, gr_rayset& dev, size_t extra_bytes, void** extra)
{
    if (!rays) return fail(GR_ERR_INVALID_ARGUMENT, "rayset is null");
    if (rays->n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (rays->sky_sampler) {
        // a source's sky: only a caller's generator has per-ray input (8 B per ray)
        const size_t n = (rays->sky_generator == 2) ? (size_t)rays->n : 0;
        if (n && !rays->sky_i) return fail(GR_ERR_INVALID_ARGUMENT, "sky source: generator 2 needs sky_i");
        // ... and a source of many positions its rows (GR_SKY_ROW doubles per ray)
        const size_t nr = rays->sky_rows ? (size_t)rays->n * GR_SKY_ROW : 0;
        int32_t rcs;
        if ((rcs = ensure(&ctx->d_in, &ctx->in_bytes, sizeof(double) * (n + nr) + extra_bytes + 64)) != GR_OK) return rcs;
        double* b = (double*)ctx->d_in;
        dev = *rays;
        dev.alpha = dev.beta = dev.area = dev.height = nullptr;
        dev.sky_i = n ? b : nullptr;
        dev.sky_rows = nr ? b + n : nullptr;
        if (n) GR_HIP(hipMemcpyAsync(b, rays->sky_i, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        if (nr) GR_HIP(hipMemcpyAsync(b + n, rays->sky_rows, sizeof(double) * nr, hipMemcpyHostToDevice, ctx->stream));
        if (extra) *extra = (void*)(b + n + nr);
        return GR_OK;
    }
    if (rays->sep_r) {
        // separable set: three small tables instead of 24 B per ray
        if (!rays->sep_cos || !rays->sep_sin || rays->sep_nr < 1 || rays->sep_nt < 1)
            return fail(GR_ERR_INVALID_ARGUMENT, "separable ray set: tables missing or empty");
        if (rays->height) return fail(GR_ERR_INVALID_ARGUMENT, "separable ray set: per-ray heights are not supported");
        const size_t nr = (size_t)rays->sep_nr, nt = (size_t)rays->sep_nt;
        int32_t rcs;
        if ((rcs = ensure(&ctx->d_in, &ctx->in_bytes, sizeof(double) * (nr + 2 * nt) + extra_bytes + 64)) != GR_OK) return rcs;
        double* b = (double*)ctx->d_in;
        dev = *rays;
        dev.alpha = dev.beta = dev.area = dev.height = nullptr;
        dev.sep_r = b; dev.sep_cos = b + nr; dev.sep_sin = b + nr + nt;
        GR_HIP(hipMemcpyAsync(b, rays->sep_r, sizeof(double) * nr, hipMemcpyHostToDevice, ctx->stream));
        GR_HIP(hipMemcpyAsync(b + nr, rays->sep_cos, sizeof(double) * nt, hipMemcpyHostToDevice, ctx->stream));
        GR_HIP(hipMemcpyAsync(b + nr + nt, rays->sep_sin, sizeof(double) * nt, hipMemcpyHostToDevice, ctx->stream));
        if (extra) *extra = (void*)(b + nr + 2 * nt);
        return GR_OK;
    }
    if (rays->n > 0 && (!rays->alpha || !rays->beta)) return fail(GR_ERR_INVALID_ARGUMENT, "alpha/beta is null");
    const size_t n = (size_t)rays->n;
    int32_t rc;
    if ((rc = ensure(&ctx->d_in, &ctx->in_bytes, sizeof(double) * 4 * n + extra_bytes + 64)) != GR_OK) return rc;
    double* base = (double*)ctx->d_in;
    dev = *rays;
    dev.alpha = base; dev.beta = base + n; dev.area = rays->area ? base + 2 * n : nullptr;
    dev.height = rays->height ? base + 3 * n : nullptr;
    if (n) {
        GR_HIP(hipMemcpyAsync(base, rays->alpha, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        GR_HIP(hipMemcpyAsync(base + n, rays->beta, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        if (rays->area) GR_HIP(hipMemcpyAsync(base + 2 * n, rays->area, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        if (rays->height) GR_HIP(hipMemcpyAsync(base + 3 * n, rays->height, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    }
    if (extra) *extra = (void*)(base + 4 * n);
    return GR_OK;
}

int32_t gr_lineprofile(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                       const gr_binning* b, double* flux, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!b || b->n_bins < 1 || !b->bin_edges || !flux) return fail(GR_ERR_INVALID_ARGUMENT, "binning/flux is null or empty");
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    gr_rayset dev;
    void* extra = nullptr;
    const size_t nb = (size_t)b->n_bins;
    const size_t ne = (b->eps_n >= 2 && b->eps_r && b->eps_v) ? (size_t)b->eps_n : 0;
    if ((rc = stage_rays(ctx, rays, dev, sizeof(double) * (2 * nb + 2 * ne), &extra)) != GR_OK) return rc;
    double* d_edges = (double*)extra;
    double* d_flux = d_edges + nb;
    GR_HIP(hipMemcpyAsync(d_edges, b->bin_edges, sizeof(double) * nb, hipMemcpyHostToDevice, ctx->stream));
    gr_binning db = *b;
    db.bin_edges = d_edges;
    if (ne) {
        double* d_er = d_flux + nb;
        GR_HIP(hipMemcpyAsync(d_er, b->eps_r, sizeof(double) * ne, hipMemcpyHostToDevice, ctx->stream));
        GR_HIP(hipMemcpyAsync(d_er + ne, b->eps_v, sizeof(double) * ne, hipMemcpyHostToDevice, ctx->stream));
        db.eps_r = d_er;
        db.eps_v = d_er + ne;
    }
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = gr_lineprofile_device(ctx, cfg, &dev, pf, &db, d_flux, stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
    GR_HIP(hipMemcpyAsync(flux, d_flux, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

int32_t gr_redshift_radius(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                           double r_min, double r_max, double* pairs, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (rays && rays->n > 0 && !pairs) return fail(GR_ERR_INVALID_ARGUMENT, "pairs is null");
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    gr_rayset dev;
    if ((rc = stage_rays(ctx, rays, dev, 0, nullptr)) != GR_OK) return rc;
    const size_t bytes = sizeof(double) * 2 * (size_t)rays->n;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return rc;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = gr_redshift_radius_device(ctx, cfg, &dev, pf, r_min, r_max, (double*)ctx->d_scratch,
                                        stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream)) != GR_OK) return rc;
    if (bytes) GR_HIP(hipMemcpyAsync(pairs, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

int32_t gr_ray_summary(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                       double* out, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (rays && rays->n > 0 && !out) return fail(GR_ERR_INVALID_ARGUMENT, "out is null");
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    gr_rayset dev;
    if ((rc = stage_rays(ctx, rays, dev, 0, nullptr)) != GR_OK) return rc;
    const size_t bytes = sizeof(double) * 4 * (size_t)rays->n;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return rc;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = gr_ray_summary_device(ctx, cfg, &dev, pf, (double*)ctx->d_scratch, stats ? (gr_stats*)ctx->d_stats : nullptr,
                                    ctx->stream)) != GR_OK) return rc;
    if (bytes) GR_HIP(hipMemcpyAsync(out, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

// ---- corona -> disc: the reductions of emissivity_profile (src/corona/radial.jl:38-100) on the summaries gr_corona_trace keeps ----
extern "C++" {
namespace {
// Over the rays that hit (finite g): min and max of ρ, max |g| and max |t| -- as ordered bit patterns (non-negative doubles: the
// order of the bits is the order of the values) -- and their number.  red[0] = min ρ, [1] = max ρ, [2] = count, [3] = max |g|,
// [4] = max |t|.
__global__ void __launch_bounds__(256) k_corona_minmax(const double* __restrict__ rows, int64_t n, unsigned long long* red)
{
    unsigned long long lo = ~0ull, hi = 0ull, cnt = 0ull, gm = 0ull, tm = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double g = rows[4 * i], rho = rows[4 * i + 1], t = rows[4 * i + 2];
        if (g == g) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(rho);
            const unsigned long long bg = (unsigned long long)__double_as_longlong(fabs(g));
            const unsigned long long bt = (unsigned long long)__double_as_longlong(fabs(t));
            lo = b < lo ? b : lo;
            hi = b > hi ? b : hi;
            gm = bg > gm ? bg : gm;
            tm = bt > tm ? bt : tm;
            ++cnt;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64), c2 = __shfl_down(cnt, off, 64);
        const unsigned long long g2 = __shfl_down(gm, off, 64), t2 = __shfl_down(tm, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        gm = g2 > gm ? g2 : gm;
        tm = t2 > tm ? t2 : tm;
        cnt += c2;
    }
    // one set of atomics per workgroup (per wave, 8192 waves on five addresses, the atomics WERE the kernel: 0.48 ms of a 9 ms call)
    __shared__ unsigned long long part[4][5];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[w][0] = lo; part[w][1] = hi; part[w][2] = cnt; part[w][3] = gm; part[w][4] = tm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            lo = part[k][0] < lo ? part[k][0] : lo;
            hi = part[k][1] > hi ? part[k][1] : hi;
            cnt += part[k][2];
            gm = part[k][3] > gm ? part[k][3] : gm;
            tm = part[k][4] > tm ? part[k][4] : tm;
        }
        if (cnt) {
            atomicMin(red, lo);
            atomicMax(red + 1, hi);
            atomicAdd(red + 2, cnt);
            atomicMax(red + 3, gm);
            atomicMax(red + 4, tm);
        }
    }
}

// A double as two integers of a fixed-point grid whose step is a power of two chosen from the largest magnitude and the number
// of values (CoronaScale): hi = round(v / step), lo = round((v / step - hi) 2^K).  Integer sums do not depend on the order of
// the additions, so the per-bin sums -- and with them the whole profile -- are the same bits on every run and for every launch
// shape, which floating-point atomics are not; the grid resolves step 2^-K, below one ulp of any value within 2^10 of the
// largest, so the sums are also as accurate as a sorted pairwise fp64 sum.
struct CoronaScale { double inv_step, two_k; };
__device__ __forceinline__ void corona_split(double v, const CoronaScale& sc, long long& hi, long long& lo)
{
    const double s = v * sc.inv_step;             // exact: a power of two
    hi = __double2ll_rn(s);
    lo = __double2ll_rn((s - (double)hi) * sc.two_k);
}

// bucket(Simple(), ρ, ...) per hit: the last edge <= ρ, clamped to the first / last bin (the rule the reference's golden emissivity
// vector pins, test/unit/emissivity.jl:27-48); per bin the count and the fixed-point sums of g and t.  acc: 5 x nb integers
// (count, g hi, g lo, t hi, t lo).  LDS = 1: the histograms are private to the workgroup in LDS and leave as one global atomic
// per non-empty entry; LDS = 0 (more bins than 40 KB of LDS hold): global atomics.
template <int LDS>
__global__ void __launch_bounds__(256) k_corona_bin(const double* __restrict__ rows, int64_t n, const double* __restrict__ edges, int nb,
                                                    CoronaScale sg, CoronaScale st, unsigned long long* acc)
{
    extern __shared__ unsigned long long hist[];
    unsigned long long* h = LDS ? hist : acc;
    if (LDS) {
        for (int i = threadIdx.x; i < 5 * nb; i += blockDim.x) hist[i] = 0ull;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double g = rows[4 * i], rho = rows[4 * i + 1], t = rows[4 * i + 2];
        if (g == g) {
            int lo = 0, hi = nb;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (edges[mid] <= rho) lo = mid + 1; else hi = mid;
            }
            lo = lo > 0 ? lo - 1 : 0;
            long long gh, gl, th, tl;
            corona_split(g, sg, gh, gl);
            corona_split(t, st, th, tl);
            atomicAdd(h + lo, 1ull);
            atomicAdd(h + nb + lo, (unsigned long long)gh);
            atomicAdd(h + 2 * nb + lo, (unsigned long long)gl);
            atomicAdd(h + 3 * nb + lo, (unsigned long long)th);
            atomicAdd(h + 4 * nb + lo, (unsigned long long)tl);
        }
    }
    if (LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < 5 * nb; i += blockDim.x)
            if (hist[i] != 0ull) atomicAdd(acc + i, hist[i]);
    }
}

// step = 2^(e + en - 62) with vmax < 2^e and n <= 2^en: Σ |hi| < 2^62, every hi below 2^(62 - en) <= 2^52 (exact in a double),
// K = 62 - en: Σ |lo| < 2^61.
struct CoronaGrid { CoronaScale sc; double step, lo_unit; };
CoronaGrid corona_grid(double vmax, int64_t n)
{
    int en = 10;
    while (en < 40 && ((int64_t)1 << en) < n) ++en;
    int e = 0;
    if (vmax > 0.0 && std::isfinite(vmax)) (void)std::frexp(vmax, &e);      // vmax = f 2^e, f in [0.5, 1)
    const int k = 62 - en;
    CoronaGrid g;
    g.step = std::ldexp(1.0, e + en - 62);
    g.sc.inv_step = std::ldexp(1.0, -(e + en - 62));
    g.sc.two_k = std::ldexp(1.0, k);
    g.lo_unit = std::ldexp(1.0, -k);
    return g;
}
int32_t rayset_share(const gr_rayset* rays, int32_t n, int k, gr_rayset& out, int64_t* off_out);      // (below, with the other *_multi helpers)
}  // namespace
}  // extern "C++"

// gr_corona_trace in two halves, so that gr_corona_trace_multi can queue every context's share before it waits for any:
// corona_enqueue stages the rays, traces, reduces (min ρ, max ρ, hits, max |g|, max |t|) and queues the 40-byte copy into `h`;
// corona_collect waits for the context and reads them.
static int32_t corona_enqueue(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf, gr_stats* stats,
                              unsigned long long* h /* 5, pinned or pageable host memory that outlives the wait */)
{
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    ctx->corona_n = -1;
    gr_rayset dev;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = stage_rays(ctx, rays, dev, 0, nullptr)) != GR_OK) return rc;
    const size_t bytes = sizeof(double) * 4 * (size_t)rays->n + 64;
    if ((rc = ensure((void**)&ctx->d_corona, &ctx->corona_bytes, bytes)) != GR_OK) return rc;
    unsigned long long* red = (unsigned long long*)(ctx->d_corona + 4 * (size_t)rays->n);
    static const unsigned long long init[5] = { ~0ull, 0ull, 0ull, 0ull, 0ull };
    GR_HIP(hipMemcpyAsync(red, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    ctx->sky_any_order = true;      // (the rows are reduced and binned: their order is the library's to choose)
    rc = gr_ray_summary_device(ctx, cfg, &dev, pf, ctx->d_corona, stats ? (gr_stats*)ctx->d_stats : nullptr, ctx->stream);
    ctx->sky_any_order = false;
    if (rc != GR_OK) return rc;
    if (rays->n > 0) {
        int64_t blocks = (rays->n + 255) / 256;
        blocks = blocks > 512 ? 512 : blocks;
        hipLaunchKernelGGL(k_corona_minmax, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->d_corona, rays->n, red);
        GR_HIP(hipGetLastError());
    }
    GR_HIP(hipMemcpyAsync(h, red, sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, ctx->stream));
    return GR_OK;
}
static int32_t corona_collect(gr_ctx* ctx, gr_stats* stats, const unsigned long long* h, int64_t n_rays, double v[5], int64_t* hits)
{
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    if ((rc = end_host_call(ctx, stats)) != GR_OK) return rc;      // (synchronises the stream)
    std::memcpy(v, h, sizeof(double) * 5);
    *hits = (int64_t)h[2];
    ctx->corona_gmax = v[3];
    ctx->corona_tmax = v[4];
    ctx->corona_hits = *hits;
    ctx->corona_n = n_rays;
    return GR_OK;
}

int32_t gr_corona_trace(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                        double* rho_min_max, int64_t* n_hits, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!rays || !rays->sky_sampler) return fail(GR_ERR_INVALID_ARGUMENT, "gr_corona_trace traces a sky source (gr_rayset.sky_sampler != 0)");
    if (!rho_min_max || !n_hits) return fail(GR_ERR_INVALID_ARGUMENT, "rho_min_max / n_hits is null");
    if (cfg && cfg->disc_id == GR_DISC_NONE) return fail(GR_ERR_INVALID_ARGUMENT, "a corona illuminates accretion geometry: none given");
    int32_t rc;
    ctx->corona_n = -1;
    unsigned long long h[5];
    double v[5];
    if ((rc = corona_enqueue(ctx, cfg, rays, pf, stats, h)) != GR_OK) return rc;
    if ((rc = corona_collect(ctx, stats, h, rays->n, v, n_hits)) != GR_OK) return rc;
    rho_min_max[0] = *n_hits ? v[0] : NAN;
    rho_min_max[1] = *n_hits ? v[1] : NAN;
    return GR_OK;
}

int32_t gr_corona_trace_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                              double* rho_min_max, int64_t* n_hits, gr_stats* stats)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    if (!rays || !rays->sky_sampler) return fail(GR_ERR_INVALID_ARGUMENT, "gr_corona_trace_multi traces a sky source (gr_rayset.sky_sampler != 0)");
    if (!rho_min_max || !n_hits) return fail(GR_ERR_INVALID_ARGUMENT, "rho_min_max / n_hits is null");
    if (cfg && cfg->disc_id == GR_DISC_NONE) return fail(GR_ERR_INVALID_ARGUMENT, "a corona illuminates accretion geometry: none given");
    for (int k = 0; k < n; ++k) ctxs[k]->corona_n = -1;
    std::vector<unsigned long long> h(5 * (size_t)n);
    std::vector<gr_rayset> share((size_t)n);
    // every context's share is queued before any is waited for
    for (int k = 0; k < n; ++k) {
        int64_t off;
        if ((rc = rayset_share(rays, n, k, share[(size_t)k], &off)) != GR_OK) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = corona_enqueue(ctxs[k], cfg, &share[(size_t)k], pf, stats ? stats + k : nullptr, h.data() + 5 * (size_t)k)) != GR_OK) return rc;
        if (stats) stats[k].enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    double lo = INFINITY, hi = -INFINITY, gmax = 0.0, tmax = 0.0;
    int64_t hits = 0;
    for (int k = 0; k < n; ++k) {
        double v[5];
        int64_t hk;
        const double enq = stats ? stats[k].enqueue_ms : 0.0;
        if ((rc = corona_collect(ctxs[k], stats ? stats + k : nullptr, h.data() + 5 * (size_t)k, share[(size_t)k].n, v, &hk)) != GR_OK) return rc;
        if (stats) stats[k].enqueue_ms = enq;
        if (hk) { lo = std::fmin(lo, v[0]); hi = std::fmax(hi, v[1]); }
        gmax = std::fmax(gmax, v[3]);
        tmax = std::fmax(tmax, v[4]);
        hits += hk;
    }
    // the fixed-point grid of the bins is formed from these: every context gets the values of ALL shares
    for (int k = 0; k < n; ++k) {
        ctxs[k]->corona_gmax = gmax;
        ctxs[k]->corona_tmax = tmax;
        ctxs[k]->corona_hits = hits;
    }
    *n_hits = hits;
    rho_min_max[0] = hits ? lo : NAN;
    rho_min_max[1] = hits ? hi : NAN;
    return GR_OK;
}

// gr_corona_bin's device half: the context's rows into integer accumulators (count, Σg hi / lo, Σt hi / lo) on the grid of its
// corona_gmax / corona_tmax / corona_hits; `acc` (5 nb, host) is filled when the stream has drained
static int32_t corona_bin_enqueue(gr_ctx* ctx, const double* edges, size_t nb, long long* acc)
{
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    if ((rc = ensure(&ctx->d_in, &ctx->in_bytes, sizeof(double) * 6 * nb + 64)) != GR_OK) return rc;
    double* d_edges = (double*)ctx->d_in;
    unsigned long long* d_acc = (unsigned long long*)(d_edges + nb);
    GR_HIP(hipMemcpyAsync(d_edges, edges, sizeof(double) * nb, hipMemcpyHostToDevice, ctx->stream));
    GR_HIP(hipMemsetAsync(d_acc, 0, sizeof(unsigned long long) * 5 * nb, ctx->stream));
    const CoronaGrid gg = corona_grid(ctx->corona_gmax, ctx->corona_hits), gt = corona_grid(ctx->corona_tmax, ctx->corona_hits);
    if (ctx->corona_n > 0) {
        int64_t blocks = (ctx->corona_n + 255) / 256;
        blocks = blocks > 1024 ? 1024 : blocks;
        if (nb <= 1024) {
            hipLaunchKernelGGL(k_corona_bin<1>, dim3((unsigned)blocks), dim3(256), sizeof(unsigned long long) * 5 * nb, ctx->stream,
                               ctx->d_corona, ctx->corona_n, d_edges, (int)nb, gg.sc, gt.sc, d_acc);
        } else {
            hipLaunchKernelGGL(k_corona_bin<0>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->d_corona, ctx->corona_n,
                               d_edges, (int)nb, gg.sc, gt.sc, d_acc);
        }
        GR_HIP(hipGetLastError());
    }
    GR_HIP(hipMemcpyAsync(acc, d_acc, sizeof(long long) * 5 * nb, hipMemcpyDeviceToHost, ctx->stream));
    return GR_OK;
}
static void corona_bins_out(const long long* acc, size_t nb, double gmax, double tmax, int64_t hits, double* out)
{
    const CoronaGrid gg = corona_grid(gmax, hits), gt = corona_grid(tmax, hits);
    for (size_t i = 0; i < nb; ++i) {
        out[i] = (double)acc[i];
        out[nb + i] = (double)(((long double)acc[nb + i] + (long double)acc[2 * nb + i] * (long double)gg.lo_unit) * (long double)gg.step);
        out[2 * nb + i] = (double)(((long double)acc[3 * nb + i] + (long double)acc[4 * nb + i] * (long double)gt.lo_unit) * (long double)gt.step);
    }
}
static int32_t corona_bin_args(const double* edges, int64_t n_edges, const double* out)
{
    if (!edges || !out || n_edges < 1 || n_edges > 65536) return fail(GR_ERR_INVALID_ARGUMENT, "edges / out is null or n_edges not in 1..65536");
    for (int64_t i = 1; i < n_edges; ++i)
        if (!(edges[i] >= edges[i - 1])) return fail(GR_ERR_INVALID_ARGUMENT, "bin edges must ascend");
    return GR_OK;
}

int32_t gr_corona_bin(gr_ctx* ctx, const double* edges, int64_t n_edges, double* out)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (ctx->corona_n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "gr_corona_bin bins the rays of the context's last gr_corona_trace: there is none");
    int32_t rc;
    if ((rc = corona_bin_args(edges, n_edges, out)) != GR_OK) return rc;
    const size_t nb = (size_t)n_edges;
    std::vector<long long> acc(5 * nb);
    if ((rc = corona_bin_enqueue(ctx, edges, nb, acc.data())) != GR_OK) return rc;
    GR_HIP(hipStreamSynchronize(ctx->stream));
    corona_bins_out(acc.data(), nb, ctx->corona_gmax, ctx->corona_tmax, ctx->corona_hits, out);
    return GR_OK;
}

int32_t gr_corona_bin_multi(gr_ctx* const* ctxs, int32_t n, const double* edges, int64_t n_edges, double* out)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    for (int k = 0; k < n; ++k) {
        if (ctxs[k]->corona_n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "gr_corona_bin_multi bins the rays of the contexts' last gr_corona_trace_multi: there is none");
        if (ctxs[k]->corona_gmax != ctxs[0]->corona_gmax || ctxs[k]->corona_tmax != ctxs[0]->corona_tmax || ctxs[k]->corona_hits != ctxs[0]->corona_hits)
            return fail(GR_ERR_INVALID_ARGUMENT, "gr_corona_bin_multi: the contexts do not hold the shares of ONE gr_corona_trace_multi");
    }
    if ((rc = corona_bin_args(edges, n_edges, out)) != GR_OK) return rc;
    const size_t nb = (size_t)n_edges;
    std::vector<long long> acc(5 * nb * (size_t)n);
    for (int k = 0; k < n; ++k)
        if ((rc = corona_bin_enqueue(ctxs[k], edges, nb, acc.data() + 5 * nb * (size_t)k)) != GR_OK) return rc;
    for (int k = 0; k < n; ++k) {
        GR_HIP(hipSetDevice(ctxs[k]->device));
        GR_HIP(hipStreamSynchronize(ctxs[k]->stream));
    }
    // integer addition is associative: the sums of the shares are the sums one context would have formed of all rays
    for (int k = 1; k < n; ++k)
        for (size_t i = 0; i < 5 * nb; ++i) acc[i] += acc[5 * nb * (size_t)k + i];
    corona_bins_out(acc.data(), nb, ctxs[0]->corona_gmax, ctxs[0]->corona_tmax, ctxs[0]->corona_hits, out);
    return GR_OK;
}

int32_t gr_ray_tangent(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                       double* out, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (rays && rays->n > 0 && !out) return fail(GR_ERR_INVALID_ARGUMENT, "out is null");
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    gr_rayset dev;
    if ((rc = stage_rays(ctx, rays, dev, 0, nullptr)) != GR_OK) return rc;
    const size_t bytes = sizeof(double) * 8 * (size_t)rays->n;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return rc;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = gr_ray_tangent_device(ctx, cfg, &dev, pf, (double*)ctx->d_scratch, stats ? (gr_stats*)ctx->d_stats : nullptr,
                                    ctx->stream)) != GR_OK) return rc;
    if (bytes) GR_HIP(hipMemcpyAsync(out, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

int32_t gr_rayset_endpoints(gr_ctx* ctx, const gr_config* cfg, const gr_rayset* rays, gr_point* points, gr_stats* stats)
{
    if (!ctx) return fail(GR_ERR_INVALID_ARGUMENT, "ctx is null");
    if (rays && rays->n > 0 && !points) return fail(GR_ERR_INVALID_ARGUMENT, "points is null");
    int32_t rc;
    GR_HIP(hipSetDevice(ctx->device));
    gr_rayset dev;
    if ((rc = stage_rays(ctx, rays, dev, 0, nullptr)) != GR_OK) return rc;
    const size_t bytes = sizeof(gr_point) * (size_t)rays->n;
    if ((rc = ensure(&ctx->d_scratch, &ctx->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return rc;
    if ((rc = begin_host_call(ctx, stats)) != GR_OK) return rc;
    if ((rc = gr_rayset_endpoints_device(ctx, cfg, &dev, (gr_point*)ctx->d_scratch, stats ? (gr_stats*)ctx->d_stats : nullptr,
                                         ctx->stream)) != GR_OK) return rc;
    if (!is_pinned(points, bytes)) prefault_output(points, bytes, ctx->hugepages != 0);
    if (bytes) GR_HIP(hipMemcpyAsync(points, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return end_host_call(ctx, stats);
}

// ---- *_multi on ray arrays and ray sets ----

int32_t gr_trace_endpoints_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const double* x, int64_t x_stride,
                                 const double* v, int64_t n_rays, gr_point* points, gr_stats* stats)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    if (n_rays < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (x_stride != 0 && x_stride != 4) return fail(GR_ERR_INVALID_ARGUMENT, "x_stride must be 0 or 4");
    if (n_rays > 0 && (!x || !v || !points)) return fail(GR_ERR_INVALID_ARGUMENT, "x/v/points is null");
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    const bool pinned = is_pinned(points, sizeof(gr_point) * (size_t)n_rays);
    auto enq = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        int64_t off, cnt;
        contiguous_share(n_rays, n, k, &off, &cnt);
        int32_t r;
        GR_HIP(hipSetDevice(c->device));
        const size_t nx = (size_t)(x_stride == 0 ? 4 : 4 * cnt), nv = (size_t)(4 * cnt);
        const size_t out_bytes = sizeof(gr_point) * (size_t)cnt;
        if ((r = ensure(&c->d_scratch, &c->scratch_bytes, out_bytes ? out_bytes : 8)) != GR_OK) return r;
        if ((r = ensure(&c->d_in, &c->in_bytes, sizeof(double) * (nx + nv) + 8)) != GR_OK) return r;
        if ((r = begin_host_call(c, stats ? &stats[k] : nullptr)) != GR_OK) return r;
        if (cnt == 0) return GR_OK;
        double* d_x = (double*)c->d_in;
        double* d_v = d_x + nx;
        GR_HIP(hipMemcpyAsync(d_x, x_stride == 0 ? x : x + 4 * off, sizeof(double) * nx, hipMemcpyHostToDevice, c->stream));
        GR_HIP(hipMemcpyAsync(d_v, v + 4 * off, sizeof(double) * nv, hipMemcpyHostToDevice, c->stream));
        return gr_trace_endpoints_device(c, cfg, d_x, x_stride, d_v, cnt, (gr_point*)c->d_scratch,
                                         stats ? (gr_stats*)c->d_stats : nullptr, c->stream);
    };
    bool faulted = false;
    auto copy = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        int64_t off, cnt;
        contiguous_share(n_rays, n, k, &off, &cnt);
        if (cnt == 0) return GR_OK;
        if (!faulted && !pinned) prefault_output(points, sizeof(gr_point) * (size_t)n_rays, c->hugepages != 0);
        faulted = true;
        GR_HIP(hipSetDevice(c->device));
        GR_HIP(hipMemcpyAsync(points + off, c->d_scratch, sizeof(gr_point) * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
        return GR_OK;
    };
    return multi_drive(ctxs, n, stats, enq, copy);
}

extern "C++" {
namespace {
// context k's contiguous share of a ray set (host pointers)
int32_t rayset_share(const gr_rayset* rays, int32_t n, int k, gr_rayset& out, int64_t* off_out)
{
    if (!rays) return fail(GR_ERR_INVALID_ARGUMENT, "rayset is null");
    if (rays->n < 0) return fail(GR_ERR_INVALID_ARGUMENT, "n must be non-negative");
    if (rays->sep_r && (rays->sep_block != 0 || rays->sep_tiled))
        return fail(GR_ERR_INVALID_ARGUMENT, "*_multi: a separable ray set with one output row per ray must come whole and in ray order "
                                             "(sep_block = 0, sep_tiled = 0)");
    int64_t off, cnt;
    contiguous_share(rays->n, n, k, &off, &cnt);
    out = *rays;
    out.n = cnt;
    if (rays->sky_sampler) {
        // a share of a source's samples: the sample numbers go on counting where the previous share stopped
        out.sky_total = rays->sky_total > 0 ? rays->sky_total : rays->n;
        out.sky_first = (rays->sky_total > 0 ? rays->sky_first : 0) + off;
        if (rays->sky_generator == 2) {
            if (rays->n > 0 && !rays->sky_i) return fail(GR_ERR_INVALID_ARGUMENT, "sky source: generator 2 needs sky_i");
            out.sky_i = rays->sky_i ? rays->sky_i + off : nullptr;
        }
        if (rays->sky_rows) out.sky_rows = rays->sky_rows + GR_SKY_ROW * off;
    } else if (rays->sep_r) {
        out.sep_first = rays->sep_first + off;
    } else if (rays->n > 0) {
        if (!rays->alpha || !rays->beta) return fail(GR_ERR_INVALID_ARGUMENT, "alpha/beta is null");
        out.alpha = rays->alpha + off;
        out.beta = rays->beta + off;
        out.area = rays->area ? rays->area + off : nullptr;
        out.height = rays->height ? rays->height + off : nullptr;
    }
    *off_out = off;
    return GR_OK;
}

// One output row of `row_bytes` per ray: context k stages its share of the rays, `launch(ctx, device rayset, d_out, d_stats)`
// queues the kernel, the rows go home with one contiguous copy.
template <class Launch>
int32_t rayset_rows_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, size_t row_bytes,
                          void* out, gr_stats* stats, Launch launch)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (!rays) return fail(GR_ERR_INVALID_ARGUMENT, "rayset is null");
    if (rays->n > 0 && !out) return fail(GR_ERR_INVALID_ARGUMENT, "output is null");
    {
        gr_rayset probe;
        int64_t o;
        if ((rc = rayset_share(rays, n, 0, probe, &o)) != GR_OK) return rc;
    }
    const size_t total = row_bytes * (size_t)(rays->n > 0 ? rays->n : 0);
    const bool pinned = is_pinned(out, total);
    auto enq = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        gr_rayset share, dev;
        int64_t off;
        int32_t r;
        if ((r = rayset_share(rays, n, k, share, &off)) != GR_OK) return r;
        GR_HIP(hipSetDevice(c->device));
        // (before the rays are staged: kernel_ms / call_ms count from the start of the call's device work, input staging included,
        // as in the single-context entry points)
        if ((r = begin_host_call(c, stats ? &stats[k] : nullptr)) != GR_OK) return r;
        if ((r = stage_rays(c, &share, dev, 0, nullptr)) != GR_OK) return r;
        const size_t bytes = row_bytes * (size_t)share.n;
        if ((r = ensure(&c->d_scratch, &c->scratch_bytes, bytes ? bytes : 8)) != GR_OK) return r;
        if (share.n == 0) return GR_OK;
        return launch(c, &dev, c->d_scratch, stats ? (gr_stats*)c->d_stats : nullptr);
    };
    bool faulted = false;
    auto copy = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        int64_t off, cnt;
        contiguous_share(rays->n, n, k, &off, &cnt);
        if (cnt == 0) return GR_OK;
        if (!faulted && !pinned) prefault_output(out, total, c->hugepages != 0);
        faulted = true;
        GR_HIP(hipSetDevice(c->device));
        GR_HIP(hipMemcpyAsync((char*)out + row_bytes * (size_t)off, c->d_scratch, row_bytes * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
        return GR_OK;
    };
    return multi_drive(ctxs, n, stats, enq, copy);
}
}  // namespace
}  // extern "C++"

int32_t gr_rayset_endpoints_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, gr_point* points,
                                  gr_stats* stats)
{
    return rayset_rows_multi(ctxs, n, cfg, rays, sizeof(gr_point), points, stats,
                             [&](gr_ctx* c, const gr_rayset* dev, void* d_out, gr_stats* d_st) {
                                 return gr_rayset_endpoints_device(c, cfg, dev, (gr_point*)d_out, d_st, c->stream);
                             });
}

int32_t gr_ray_summary_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                             double* out, gr_stats* stats)
{
    return rayset_rows_multi(ctxs, n, cfg, rays, sizeof(double) * 4, out, stats,
                             [&](gr_ctx* c, const gr_rayset* dev, void* d_out, gr_stats* d_st) {
                                 return gr_ray_summary_device(c, cfg, dev, pf, (double*)d_out, d_st, c->stream);
                             });
}

int32_t gr_ray_tangent_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                             double* out, gr_stats* stats)
{
    return rayset_rows_multi(ctxs, n, cfg, rays, sizeof(double) * 8, out, stats,
                             [&](gr_ctx* c, const gr_rayset* dev, void* d_out, gr_stats* d_st) {
                                 return gr_ray_tangent_device(c, cfg, dev, pf, (double*)d_out, d_st, c->stream);
                             });
}

int32_t gr_redshift_radius_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays,
                                 const gr_pointfunction* pf, double r_min, double r_max, double* pairs, gr_stats* stats)
{
    return rayset_rows_multi(ctxs, n, cfg, rays, sizeof(double) * 2, pairs, stats,
                             [&](gr_ctx* c, const gr_rayset* dev, void* d_out, gr_stats* d_st) {
                                 return gr_redshift_radius_device(c, cfg, dev, pf, r_min, r_max, (double*)d_out, d_st, c->stream);
                             });
}

int32_t gr_lineprofile_multi(gr_ctx* const* ctxs, int32_t n, const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf,
                             const gr_binning* b, double* flux, gr_stats* stats)
{
    int32_t rc;
    if ((rc = validate_ctxs(ctxs, n)) != GR_OK) return rc;
    if (n > 64) return fail(GR_ERR_INVALID_ARGUMENT, "at most 64 contexts");
    if ((rc = validate_cfg(cfg)) != GR_OK) return rc;
    if (!rays) return fail(GR_ERR_INVALID_ARGUMENT, "rayset is null");
    if (!b || b->n_bins < 1 || !b->bin_edges || !flux) return fail(GR_ERR_INVALID_ARGUMENT, "binning/flux is null or empty");
    if (rays->sep_r && rays->sep_block != 0)
        return fail(GR_ERR_INVALID_ARGUMENT, "gr_lineprofile_multi: a separable ray set must come whole (sep_block = 0): the call deals it itself");
    const size_t nb = (size_t)b->n_bins;
    const size_t ne = (b->eps_n >= 2 && b->eps_r && b->eps_v) ? (size_t)b->eps_n : 0;
    // the deal of a separable plane: blocks of one strip of 8 x 8 tiles (all radii of 8 neighbouring angles) in the set's
    // visiting order, block k, k + n, ... to context k -- every context sees every radius and an even sample of the angles
    // (gradus.jl_amd/distributed.py: ray_shard does the same across processes)
    const int64_t N = rays->n;
    int64_t sep_blk = 0;
    if (rays->sep_r) {
        const int64_t nr = rays->sep_nr, nt = rays->sep_nt;
        sep_blk = (nr >= 8 && nt >= 8) ? 8 * ((nr / 8) * 8) : std::max<int64_t>(64, (N + (int64_t)n * 8 - 1) / ((int64_t)n * 8));
    }
    std::vector<double> part;
    try {
        part.assign(nb * (size_t)n, 0.0);
    } catch (...) {
        return fail(GR_ERR_OUT_OF_MEMORY, "gr_lineprofile_multi: partial histograms");
    }
    auto share_of = [&](int k, gr_rayset& sh) -> int32_t {
        if (!rays->sep_r) {
            int64_t off;
            return rayset_share(rays, n, k, sh, &off);
        }
        sh = *rays;
        const int64_t blocks = (N + sep_blk - 1) / sep_blk;
        const int64_t mine = k < blocks ? (blocks - 1 - k) / n + 1 : 0;
        int64_t cnt = 0;
        if (mine > 0) {
            const int64_t last = k + (mine - 1) * n;
            cnt = (mine - 1) * sep_blk + std::min<int64_t>(sep_blk, N - last * sep_blk);
        }
        sh.n = cnt;
        sh.sep_first = rays->sep_first + (int64_t)k * sep_blk;
        sh.sep_block = sep_blk;
        sh.sep_stride = (int64_t)n * sep_blk;
        return GR_OK;
    };
    std::vector<double*> d_part((size_t)n, nullptr);
    auto enq = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        gr_rayset sh, dev;
        int32_t r;
        if ((r = share_of(k, sh)) != GR_OK) return r;
        GR_HIP(hipSetDevice(c->device));
        void* extra = nullptr;
        if ((r = begin_host_call(c, stats ? &stats[k] : nullptr)) != GR_OK) return r;      // (before the staging, see rayset_rows_multi)
        if ((r = stage_rays(c, &sh, dev, sizeof(double) * (2 * nb + 2 * ne), &extra)) != GR_OK) return r;
        double* d_edges = (double*)extra;
        double* d_flux = d_edges + nb;
        GR_HIP(hipMemcpyAsync(d_edges, b->bin_edges, sizeof(double) * nb, hipMemcpyHostToDevice, c->stream));
        gr_binning db = *b;
        db.bin_edges = d_edges;
        if (ne) {
            double* d_er = d_flux + nb;
            GR_HIP(hipMemcpyAsync(d_er, b->eps_r, sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
            GR_HIP(hipMemcpyAsync(d_er + ne, b->eps_v, sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
            db.eps_r = d_er;
            db.eps_v = d_er + ne;
        }
        d_part[(size_t)k] = d_flux;
        // (a context without rays still zeroes its histogram: gr_lineprofile_device does so before it looks at n)
        return gr_lineprofile_device(c, cfg, &dev, pf, &db, d_flux, stats ? (gr_stats*)c->d_stats : nullptr, c->stream);
    };
    auto copy = [&](int k) -> int32_t {
        gr_ctx* c = ctxs[k];
        GR_HIP(hipSetDevice(c->device));
        GR_HIP(hipMemcpyAsync(part.data() + nb * (size_t)k, d_part[(size_t)k], sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream));
        return GR_OK;
    };
    if ((rc = multi_drive(ctxs, n, stats, enq, copy)) != GR_OK) return rc;
    for (size_t i = 0; i < nb; ++i) {
        double sum = 0.0;
        for (int k = 0; k < n; ++k) sum += part[nb * (size_t)k + i];     // context order: the same bytes run after run
        flux[i] = sum;
    }
    return GR_OK;
}

}  // extern "C"

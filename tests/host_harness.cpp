// host_harness.cpp -- TEST INFRASTRUCTURE.  Compiles the device integrator (gr_device.hpp) for
// the host with g++ so that tests can run the exact kernel logic ray by ray on a CPU, next to the
// oracle, and log its steps.  Never linked into libgradus_mi355x.so.
#define GR_HOST_HARNESS 1
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../gradus.jl_amd/csrc/gr_device.hpp"
#include "../gradus.jl_amd/csrc/gr_mesh_grid.hpp"

using namespace GR_NS;

template <class Metric, int DISC>
static void run(const Params& p, int64_t n, double* tlog, double* hlog, int64_t cap, int64_t* nlog)
{
    Metric m;
    m.load(p.cfg);
    for (int64_t j = 0; j < n; ++j) {
        Ray<Metric, DISC> ray;
        ray.init(m, p, j);
        int64_t k = 0;
        if (tlog && k < cap) { tlog[k] = ray.t; hlog[k] = ray.dt; ++k; }
        for (;;) {
            const bool fin = ray.step(m, p);
            if (tlog && k < cap) { tlog[k] = ray.t; hlog[k] = ray.x[2]; ++k; }
            if (fin) break;
        }
        const LdsView no_lds{ nullptr, nullptr, nullptr, nullptr, nullptr };
        ray.finalize(m, p, no_lds);
        if (tlog && k < cap) { tlog[k] = -1.0; hlog[k] = ray.dbg_e2; ++k; }
        if (nlog) *nlog = k;
    }
}

static void dispatch(Params& p, double* tlog, double* hlog, int64_t cap, int64_t* nlog)
{
    derive_params(p);
    p.disc_table = p.cfg.disc_table;      // host pointer is directly usable here
    static thread_local std::vector<double> mesh_table;
    if (p.cfg.disc_id == GR_DISC_MESH) {  // ... a mesh goes through the builder the host unit uses (gr_mesh_grid.hpp)
        if (!gr_mesh::vertices_finite(p.cfg.disc_table, p.cfg.disc_table_n)) return;
        gr_mesh::build_table(p.cfg.disc_table, p.cfg.disc_table_n, mesh_table);
        p.disc_table = mesh_table.data();
    }
    p.cfg.upper_hemisphere = (p.cfg.upper_hemisphere ? 1 : 0) | (p.cfg.count_windings ? 4 : 0);   // as stage_disc_table does
    if (p.cfg.metric_id == GR_METRIC_TABULATED) {       // ... and stage_metric_table: the table's header into cfg.params (the host pointer stays)
        gr_tab::stage_params(p.cfg.metric_table, p.cfg.params);
    }
    const int disc = p.cfg.disc_id;
#define HH_RUN(M) \
    do { if (disc == GR_DISC_THIN) run<M, GR_DISC_THIN>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_SHAKURA_SUNYAEV) run<M, GR_DISC_SHAKURA_SUNYAEV>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_TABULATED) run<M, GR_DISC_TABULATED>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_DATUM) run<M, GR_DISC_DATUM>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_ELLIPTICAL) run<M, GR_DISC_ELLIPTICAL>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_PRECESSING_THIN) run<M, GR_DISC_PRECESSING_THIN>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_COMPOSITE) run<M, GR_DISC_COMPOSITE>(p, p.n, tlog, hlog, cap, nlog); \
         else if (disc == GR_DISC_MESH) run<M, GR_DISC_MESH>(p, p.n, tlog, hlog, cap, nlog); \
         else run<M, GR_DISC_NONE>(p, p.n, tlog, hlog, cap, nlog); } while (0)
    if (p.cfg.metric_id == GR_METRIC_KERR) HH_RUN(KerrMetric);
    else if (p.cfg.metric_id == GR_METRIC_KERR_NEWMAN) HH_RUN(KerrNewmanMetric);
    else if (p.cfg.metric_id == GR_METRIC_JOHANNSEN) HH_RUN(JohannsenMetric);
    else if (p.cfg.metric_id == GR_METRIC_TABULATED) HH_RUN(TabulatedMetric);
    else HH_RUN(GenericMetric);
#undef HH_RUN
}

// Lock-step emulation of one wave (64 rays of an 8x8 pixel tile): at wave iteration k every still-active lane does
// its k-th attempted step, as the lanes of k_trace_lane do.  Counts, per wave-iteration, how often ANY lane takes a
// divergent block (those blocks then cost the whole wave their instructions): full sincos per stage, event sampling
// past the reach bound, past the θ samples.  out[0] = wave-iterations, out[1..5] = full-sincos stages 1..5,
// out[6] = sampling level 1, out[7] = level 2, out[8] = lane-steps, out[9] = lane-level sum of level-1.
template <class Metric, int DISC>
static void wave_stats(const Params& p, const int64_t* tiles, int64_t n_tiles, double* out)
{
    Metric m;
    m.load(p.cfg);
    const Cold& cd = *p.cold;
    const int64_t H = cd.plane.height;
    for (int64_t t = 0; t < n_tiles; ++t) {
        Ray<Metric, DISC> ray[64];
        bool act[64];
        const int64_t tiles_per_col = H >> 3;
        const int64_t tx = tiles[t] / tiles_per_col, ty = tiles[t] - tx * tiles_per_col;
        for (int l = 0; l < 64; ++l) {
            const int64_t j = ((tx << 3) + (l >> 3)) * H + (ty << 3) + (l & 7);
            ray[l].init(m, p, j);
            act[l] = true;
        }
        for (;;) {
            int bits = 0, nact = 0;
            double dmax = 0.0;
            for (int l = 0; l < 64; ++l) {
                if (!act[l]) continue;
                ++nact;
                if (ray[l].step(m, p)) act[l] = false;
                bits |= ray[l].dbg_bits;
                dmax = std::fmax(dmax, (double)ray[l].dbg_dmax);
                out[8] += 1.0;
                if (ray[l].dbg_bits & (1 << 8)) out[9] += 1.0;
            }
            if (!nact) break;
            out[0] += 1.0;
            for (int s = 1; s <= 5; ++s) if (bits & (1 << s)) out[s] += 1.0;
            if (bits & (1 << 8)) out[6] += 1.0;
            if (bits & (1 << 9)) out[7] += 1.0;
            // histogram of the wave's largest |δ| in this iteration: out[10 + k] counts 2^-(k+1) < dmax <= 2^-k, k = 0..23
            int k = dmax > 0.0 ? (int)std::floor(-std::log2(dmax)) : 23;
            k = k < 0 ? 0 : (k > 23 ? 23 : k);
            out[10 + k] += 1.0;
        }
    }
}

extern "C" {

int hh_wave_stats(const gr_config* cfg, const gr_plane* plane, const int64_t* tiles, int64_t n_tiles, double* out)
{
    Params p; Cold c;
    std::memset(&p, 0, sizeof p); std::memset(&c, 0, sizeof c);
    const int64_t n = plane->width * plane->height;
    p.cfg = *cfg; p.n = n; p.cold = &c; c.winding_plane = cfg->winding_plane;
    c.src_mode = 0; c.out_mode = 0; c.plane = *plane; c.range = gr_range{ 0, n, n, 1 };
    derive_params(p);
    p.cfg.upper_hemisphere = 0;
    for (int i = 0; i < 34; ++i) out[i] = 0.0;
    if (cfg->metric_id != GR_METRIC_KERR || cfg->disc_id != GR_DISC_THIN) return -1;
    wave_stats<KerrMetric, GR_DISC_THIN>(p, tiles, n_tiles, out);
    return 0;
}

// both forms of the right-hand side at one point: the metric's fused rhs() and eval() + the generic contraction
int hh_rhs_both(const gr_config* cfg, double r, double th, const double* v, double* fused, double* generic)
{
    const double s = std::sin(th), c = std::cos(th);
    if (cfg->metric_id == GR_METRIC_KERR) {
        KerrMetric m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_KERR_NEWMAN) {
        KerrNewmanMetric m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_JOHANNSEN) {
        JohannsenMetric m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_JOHANNSEN_PSALTIS) {
        GenericMetricT<GR_METRIC_JOHANNSEN_PSALTIS> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_BUMBLEBEE) {
        GenericMetricT<GR_METRIC_BUMBLEBEE> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_KERR_DARK_MATTER) {
        GenericMetricT<GR_METRIC_KERR_DARK_MATTER> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_DILATON_AXION) {
        GenericMetricT<GR_METRIC_DILATON_AXION> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_NOZ) {
        GenericMetricT<GR_METRIC_NOZ> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_SPHERICAL) {
        GenericMetricT<GR_METRIC_SPHERICAL> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_KERR_REFRACTIVE) {
        GenericMetricT<GR_METRIC_KERR_REFRACTIVE> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else if (cfg->metric_id == GR_METRIC_MORRIS_THORNE) {
        GenericMetricT<GR_METRIC_MORRIS_THORNE> m; m.load(*cfg);
        m.rhs(r, s, c, v[0], v[1], v[2], v[3], fused[0], fused[1], fused[2], fused[3]);
        geodesic_rhs_generic(m, r, s, c, v[0], v[1], v[2], v[3], generic[0], generic[1], generic[2], generic[3]);
    } else {
        return -1;
    }
    return 0;
}

int hh_render_endpoints(const gr_config* cfg, const gr_plane* plane, const gr_range* rg, gr_point* out)
{
    Params p; Cold c;
    std::memset(&p, 0, sizeof p); std::memset(&c, 0, sizeof c);
    p.cfg = *cfg; p.n = rg->count; p.cold = &c; c.winding_plane = cfg->winding_plane;
    c.src_mode = 0; c.out_mode = 1; c.plane = *plane; c.range = *rg; c.points = out;
    dispatch(p, nullptr, nullptr, 0, nullptr);
    return 0;
}

int hh_render(const gr_config* cfg, const gr_plane* plane, const gr_range* rg, const gr_pointfunction* pf, double* image)
{
    Params p; Cold c;
    std::memset(&p, 0, sizeof p); std::memset(&c, 0, sizeof c);
    p.cfg = *cfg; p.n = rg->count; p.cold = &c; c.winding_plane = cfg->winding_plane;
    c.src_mode = 0; c.out_mode = 0; c.plane = *plane; c.range = *rg; c.image = image;
    c.pf.pf_id = pf->pf_id; c.pf.filter_id = pf->filter_id; c.pf.fill = pf->fill; c.pf.r_isco = pf->r_isco;
    c.pf.n_plunge = pf->n_plunge; c.pf.plunge_r = pf->plunge_r; c.pf.plunge_vt = pf->plunge_vt;
    c.pf.plunge_vr = pf->plunge_vr; c.pf.plunge_vphi = pf->plunge_vphi;
    dispatch(p, nullptr, nullptr, 0, nullptr);
    return 0;
}

int hh_trace_endpoints(const gr_config* cfg, const double* x, int64_t x_stride, const double* v, int64_t n, gr_point* out)
{
    Params p; Cold c;
    std::memset(&p, 0, sizeof p); std::memset(&c, 0, sizeof c);
    p.cfg = *cfg; p.n = n; p.cold = &c; c.winding_plane = cfg->winding_plane;
    c.src_mode = 1; c.out_mode = 1; c.x = x; c.x_stride = x_stride; c.v = v; c.points = out;
    c.range = gr_range{ 0, n, n > 0 ? n : 1, 1 };
    dispatch(p, nullptr, nullptr, 0, nullptr);
    return 0;
}

// one ray of a plane with a log of (t, EEst^2) after every attempted step
int64_t hh_step_log(const gr_config* cfg, const gr_plane* plane, int64_t i, gr_point* out, double* tlog, double* hlog, int64_t cap)
{
    Params p; Cold c;
    std::memset(&p, 0, sizeof p); std::memset(&c, 0, sizeof c);
    p.cfg = *cfg; p.n = 1; p.cold = &c; c.winding_plane = cfg->winding_plane;
    c.src_mode = 0; c.out_mode = 1; c.plane = *plane; c.range = gr_range{ i, 1, 1, 1 }; c.points = out;
    int64_t n = 0;
    dispatch(p, tlog, hlog, cap, &n);
    return n;
}

// The grid of GR_DISC_MESH against the reference's loop, for one point: how many triangles have their first vertex within 3 of
// `q` (the whole list walked, meshes.jl:53-64), how many of those the grid walk of the kernels reaches (gr_mesh_grid.hpp +
// Ray::mesh_cells: must be all of them), and how many triangles the grid walk visits in all.  table: 6 extents + 9 n doubles.
int hh_mesh_candidates(const double* table, int64_t n, const double* q, int64_t* out3)
{
    if (!gr_mesh::vertices_finite(table, n)) return -1;
    std::vector<double> tb;
    gr_mesh::build_table(table, n, tb);
    typedef Ray<KerrMetric, GR_DISC_MESH> R;
    const real Q2[3] = { q[0], q[1], q[2] };
    int lo[3], hi[3];
    R::mesh_cells(tb.data(), Q2, lo, hi);
    const int nx = (int)tb[10], ny = (int)tb[11];
    const uint32_t* cs = reinterpret_cast<const uint32_t*>(tb.data() + 16);
    const double* T0 = tb.data() + (int64_t)tb[13];
    int64_t brute = 0, visited = 0, reached = 0;
    std::vector<char> seen((size_t)n, 0);           // by sorted slot
    if (hi[0] >= lo[0])
        for (int iz = lo[2]; iz <= hi[2]; ++iz)
            for (int iy = lo[1]; iy <= hi[1]; ++iy) {
                const int64_t row = ((int64_t)iz * ny + iy) * nx;
                for (uint32_t k = cs[row + lo[0]]; k < cs[row + hi[0] + 1]; ++k) { seen[k] = 1; ++visited; }
            }
    // every triangle of the caller's list within 3 must be among the visited slots: match by its vertices (slot order differs)
    for (int64_t t = 0; t < n; ++t) {
        const double* V = table + 6 + 9 * t;
        const double dx = V[0] - q[0], dy = V[1] - q[1], dz = V[2] - q[2];
        if (!(dx * dx + dy * dy + dz * dz < 9.0)) continue;
        ++brute;
        for (int64_t k = 0; k < n; ++k)
            if (seen[(size_t)k] && std::memcmp(T0 + 3 * k, V, 3 * sizeof(double)) == 0
                && std::memcmp(T0 + 3 * n + 6 * k, V + 3, 6 * sizeof(double)) == 0) { ++reached; break; }
    }
    out3[0] = brute; out3[1] = reached; out3[2] = visited;
    return 0;
}
}

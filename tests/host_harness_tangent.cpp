// host_harness_tangent.cpp -- TEST INFRASTRUCTURE.  The device integrator (gr_device.hpp) compiled for the host in its
// TANGENT flavour (real = gr_tan2: value + ∂/∂α + ∂/∂β), so the CPU suite can check "dual numbers through the
// integrator" against central differences of the plain build and of the oracle without a GPU.
#define GR_HOST_HARNESS 1
#define GR_REAL_IS_TAN2 1
#define GR_NS grt
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../gradus.jl_amd/csrc/gr_device.hpp"

using namespace grt;

static int32_t* g_steps = nullptr;      // per-ray attempted steps of the next run (hht_set_steps_out), or null

template <class Metric, int DISC>
static void run(const Params& p)
{
    Metric m;
    m.load(p.cfg);
    const LdsView no_lds{ nullptr, nullptr, nullptr, nullptr, nullptr };
    for (int64_t j = 0; j < p.n; ++j) {
        Ray<Metric, DISC> ray;
        ray.init(m, p, j);
        while (!ray.step(m, p)) {}
        ray.finalize(m, p, no_lds);
        if (g_steps) g_steps[j] = ray.nacc + ray.nrej;
    }
}

static int g_tangent_norm = 1;      // the library's default (gr_ctx: tangent_norm = 1)

extern "C" {

// where the next hht_ray_tangent writes every ray's number of attempted steps (n int32; null = nowhere)
void hht_set_steps_out(int32_t* steps) { g_steps = steps; }

// 1 = the error norm over values and tangents (what gr_ctx_set(ctx, "tangent_norm", 1) selects in the library)
void hht_set_tangent_norm(int on) { g_tangent_norm = on ? 1 : 0; }

// rays given by impact parameters; out: n x 8 = (g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, t, status)
int hht_ray_tangent(const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf, double* out)
{
    Params p; Cold c;
    std::memset(&p, 0, sizeof p); std::memset(&c, 0, sizeof c);
    p.cfg = *cfg; p.n = rays->n; p.cold = &c;
    c.src_mode = 2; c.out_mode = 5;
    std::memcpy(c.plane.x_obs, rays->x_obs, sizeof c.plane.x_obs);
    std::memcpy(c.plane.Mx, rays->Mx, sizeof c.plane.Mx);
    c.plane.width = rays->n; c.plane.height = 1;
    c.range = gr_range{ 0, rays->n, rays->n > 0 ? rays->n : 1, 1 };
    c.alpha = rays->alpha; c.beta = rays->beta; c.area = rays->area;
    if (rays->sep_r) {                                    // as rays_params() of the host unit
        c.sep_r = rays->sep_r; c.sep_cos = rays->sep_cos; c.sep_sin = rays->sep_sin;
        c.sep_nr = rays->sep_nr; c.sep_nt = rays->sep_nt; c.sep_first = rays->sep_first; c.sep_block = rays->sep_block; c.sep_stride = rays->sep_stride;
        const bool tiled = rays->sep_tiled && rays->sep_nr >= 8 && rays->sep_nt >= 8;
        c.sep_core_rows = tiled ? (rays->sep_nr / 8) * 8 : 0;
        c.sep_core_cols = tiled ? (rays->sep_nt / 8) * 8 : 0;
        c.alpha = c.beta = c.area = nullptr;
    }
    c.height = cfg->disc_id == GR_DISC_DATUM ? rays->height : nullptr;
    c.lp_rmin = 0.0; c.lp_rmax = INFINITY; c.lp_pairs = out;
    c.pf.pf_id = pf->pf_id; c.pf.filter_id = pf->filter_id; c.pf.fill = pf->fill; c.pf.r_isco = pf->r_isco;
    c.pf.n_plunge = pf->n_plunge; c.pf.plunge_r = pf->plunge_r; c.pf.plunge_vt = pf->plunge_vt;
    c.pf.plunge_vr = pf->plunge_vr; c.pf.plunge_vphi = pf->plunge_vphi;
    c.winding_plane = cfg->winding_plane;
    derive_params(p);
    p.tangent_norm = g_tangent_norm;
    p.disc_table = p.cfg.disc_table;
    p.cfg.upper_hemisphere = (p.cfg.upper_hemisphere ? 1 : 0);
    const int disc = p.cfg.disc_id;
#define HH_RUN(M) \
    do { if (disc == GR_DISC_THIN) run<M, GR_DISC_THIN>(p); \
         else if (disc == GR_DISC_SHAKURA_SUNYAEV) run<M, GR_DISC_SHAKURA_SUNYAEV>(p); \
         else if (disc == GR_DISC_TABULATED) run<M, GR_DISC_TABULATED>(p); \
         else if (disc == GR_DISC_DATUM) run<M, GR_DISC_DATUM>(p); \
         else return -1; } while (0)
    if (p.cfg.metric_id == GR_METRIC_KERR) HH_RUN(KerrMetric);
    else if (p.cfg.metric_id == GR_METRIC_KERR_NEWMAN) HH_RUN(KerrNewmanMetric);
    else if (p.cfg.metric_id == GR_METRIC_JOHANNSEN) HH_RUN(JohannsenMetric);
    else HH_RUN(GenericMetric);
#undef HH_RUN
    return 0;
}
}

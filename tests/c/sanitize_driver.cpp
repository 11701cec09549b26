// sanitize_driver.cpp -- TEST INFRASTRUCTURE.  One executable that links the CPU oracle (oracle/gradus_oracle.c) and the
// host build of the device integrator (tests/host_harness.cpp) and pushes a handful of scenes through both, for
// AddressSanitizer / UndefinedBehaviorSanitizer runs (GPU sanitizers are not available on the test pool):
//
//     make -C tests/c asan        (builds and runs; any report makes the run fail)
//
// Scenes: Kerr / Johannsen / Kerr-Newman(q) / Johannsen-Psaltis x no disc / thin disc / Shakura-Sunyaev / the
// smoke-test torus, 12 x 12 pixels each, plus a plunging table and the fused point function.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../oracle/gradus_oracle.h"
#include "../../include/gradus_mi355x.h"

extern "C" {
int hh_render_endpoints(const gr_config* cfg, const gr_plane* plane, const gr_range* rg, gr_point* out);
int hh_render(const gr_config* cfg, const gr_plane* plane, const gr_range* rg, const gr_pointfunction* pf, double* image);
// tests/host_harness_tangent.cpp: the tangent flavour (value + d/dalpha + d/dbeta through the integrator)
int hht_ray_tangent(const gr_config* cfg, const gr_rayset* rays, const gr_pointfunction* pf, double* out);
}

static const double PI = 3.141592653589793;

// (metric_table.hip reports through the host unit's error sink: a stand-in for this driver; its table check is internal to the library)
int32_t gr_metric_table_check(const double* table, int64_t table_n);
int32_t gr_set_last_error(int32_t code, const char* msg) { std::printf("library error %d: %s\n", code, msg); return code; }

int main()
{
    // the product's config is the oracle's plus the two fields of a tabulated metric at its end (ABI 7: metric_table, metric_table_n)
    static_assert(sizeof(orc_config) + sizeof(const double*) + sizeof(int64_t) == sizeof(gr_config), "oracle and product configs share their layout up to the oracle's end");
    static_assert(sizeof(orc_point) == sizeof(gr_point) && sizeof(gr_point) == 152, "GeodesicPoint is 152 bytes");
    const int W = 12, H = 12, N = W * H;
    const double x[4] = { 0.0, 100.0, 85.0 * PI / 180.0, 0.0 };
    struct Scene { int metric; double p[4]; int disc; double q; };
    const Scene scenes[] = {
        { ORC_METRIC_KERR, { 1.0, 0.998, 0, 0 }, ORC_DISC_NONE, 0 },   { ORC_METRIC_KERR, { 1.0, 0.998, 0, 0 }, ORC_DISC_THIN, 0 },
        { ORC_METRIC_KERR, { 1.0, 0.0, 0, 0 }, ORC_DISC_SHAKURA_SUNYAEV, 0 }, { ORC_METRIC_JOHANNSEN, { 1.0, 0.7, 2.0, 0 }, ORC_DISC_THIN, 0 },
        { ORC_METRIC_KERR_NEWMAN, { 1.0, 0.5, 0.4, 0 }, ORC_DISC_THIN, 0.7 }, { ORC_METRIC_JOHANNSEN_PSALTIS, { 1.0, 0.6, 1.0, 0 }, ORC_DISC_THIN, 0 },
    };
    int bad = 0;
    for (const Scene& s : scenes) {
        orc_config c;
        std::memset(&c, 0, sizeof c);
        c.metric_id = s.metric;
        for (int i = 0; i < 4; ++i) c.params[i] = s.p[i];
        if (s.metric == ORC_METRIC_JOHANNSEN) { c.params[5] = 1.0; c.params[3] = 0.0; }
        const double M = s.p[0], a = s.p[1], Q = s.metric == ORC_METRIC_KERR_NEWMAN ? s.p[2] : 0.0;
        c.r_inner = 1.01 * (M + std::sqrt(M * M - a * a - Q * Q));
        c.r_outer = 12000.0;
        c.disc_id = s.disc;
        c.disc_r_in = s.disc == ORC_DISC_SHAKURA_SUNYAEV ? 6.0 : 2.0;
        c.disc_r_out = s.disc == ORC_DISC_SHAKURA_SUNYAEV ? INFINITY : 40.0;
        c.disc_params[0] = 0.3; c.disc_params[1] = 1.0 / (1.0 - std::sqrt(8.0 / 9.0));
        c.gtol = 1e-2; c.lambda0 = 0.0; c.lambda1 = 200.0; c.abstol = c.reltol = 1e-9; c.maxiters = 1000000;
        c.hemi_delta = 1e-4; c.q = s.q; c.winding_plane = PI / 2;
        std::vector<double> v(4 * N);
        orc_render_velocities(&c, x, -9.5, 9.5, -9.5, 9.5, W, H, 0, N, v.data());
        std::vector<orc_point> ref(N);
        if (orc_trace(&c, x, 0, v.data(), N, ref.data(), nullptr, 2) != 0) { std::printf("orc_trace failed\n"); return 2; }
        // the same scene through the host build of the device integrator
        gr_config g;
        std::memset(&g, 0, sizeof g);
        std::memcpy(&g, &c, sizeof c);
        g.disc_id = s.disc;      // THIN / SHAKURA_SUNYAEV / NONE share their ids between the two enums
        gr_plane pl;
        std::memset(&pl, 0, sizeof pl);
        std::memcpy(pl.x_obs, x, sizeof x);
        orc_lnr_transform(&c, x, pl.Mx);
        pl.alpha0 = -9.5; pl.alpha1 = 9.5; pl.beta0 = -9.5; pl.beta1 = 9.5; pl.width = W; pl.height = H; pl.offset = 1e-6;
        const gr_range rg{ 0, N, N, 1 };
        std::vector<gr_point> got(N);
        hh_render_endpoints(&g, &pl, &rg, got.data());
        int mism = 0;
        double worst = 0.0;
        for (int i = 0; i < N; ++i) {
            if (got[i].status != ref[i].status) { ++mism; continue; }
            if (ref[i].status == 1) continue;      // captured rays stop at whichever step lands inside the chart
            for (int k = 1; k < 3; ++k) worst = std::fmax(worst, std::fabs(got[i].x[k] - ref[i].x[k]) / std::fmax(1.0, std::fabs(ref[i].x[k])));
        }
        std::printf("metric %d disc %d: status mismatches %d / %d, worst end-point difference %.2e\n", s.metric, s.disc, mism, N, worst);
        if (mism > 3 || worst > 1e-6) ++bad;
        // fused point function + the oracle's
        gr_pointfunction pf;
        std::memset(&pf, 0, sizeof pf);
        pf.pf_id = GR_PF_AFFINE_TIME; pf.filter_id = GR_FILTER_EARLY_TERM; pf.fill = NAN;
        std::vector<double> img(N);
        hh_render(&g, &pl, &rg, &pf, img.data());
        orc_pf opf;
        std::memset(&opf, 0, sizeof opf);
        opf.pf_id = ORC_PF_AFFINE_TIME; opf.filter_id = ORC_FILTER_EARLY_TERM; opf.fill = NAN;
        std::vector<double> oimg(N);
        orc_apply_pf(&c, &opf, ref.data(), N, 200.0, oimg.data(), 2);
    }
    {   // dual numbers through the integrator: Kerr and Johannsen-Psaltis rays against the datum plane and a thin disc
        for (int metric : { (int)GR_METRIC_KERR, (int)GR_METRIC_JOHANNSEN_PSALTIS }) {
            for (int disc : { (int)GR_DISC_DATUM, (int)GR_DISC_THIN }) {
                orc_config c;
                std::memset(&c, 0, sizeof c);
                c.metric_id = metric; c.params[0] = 1.0; c.params[1] = 0.6; c.params[2] = metric == GR_METRIC_KERR ? 0.0 : 1.0;
                c.r_inner = 1.01 * (1.0 + std::sqrt(1.0 - 0.36)); c.r_outer = 12000.0;
                c.gtol = 1e-2; c.lambda0 = 0.0; c.lambda1 = 200.0; c.abstol = c.reltol = 1e-9; c.maxiters = 1000000;
                c.hemi_delta = 1e-4; c.winding_plane = PI / 2;
                gr_config g;
                std::memset(&g, 0, sizeof g);
                std::memcpy(&g, &c, sizeof c);
                g.disc_id = disc; g.disc_r_in = 0.0; g.disc_r_out = disc == GR_DISC_THIN ? 40.0 : INFINITY;
                gr_rayset rs;
                std::memset(&rs, 0, sizeof rs);
                std::memcpy(rs.x_obs, x, sizeof x);
                orc_lnr_transform(&c, x, rs.Mx);
                std::vector<double> al(40), be(40), out(8 * 40);
                for (int i = 0; i < 40; ++i) { al[i] = -8.0 + 0.4 * i; be[i] = 1.0 + 0.1 * i; }
                rs.alpha = al.data(); rs.beta = be.data(); rs.n = 40;
                gr_pointfunction pf;
                std::memset(&pf, 0, sizeof pf);
                pf.pf_id = GR_PF_REDSHIFT; pf.filter_id = GR_FILTER_NONE; pf.fill = NAN; pf.r_isco = 0.5;   // Keplerian branch everywhere
                if (hht_ray_tangent(&g, &rs, &pf, out.data()) != 0) { std::printf("hht_ray_tangent failed\n"); ++bad; continue; }
                int hits = 0, finite = 0;
                for (int i = 0; i < 40; ++i) {
                    if (out[8 * i + 7] == 2.0) ++hits;
                    if (std::isfinite(out[8 * i + 2]) && std::isfinite(out[8 * i + 4])) ++finite;
                }
                std::printf("tangent flavour, metric %d disc %d: %d of 40 rays hit, %d with finite tangents\n", metric, disc, hits, finite);
                if (hits < 10 || finite < hits) ++bad;
            }
        }
    }
    {   // a tabulated metric: the host-only table functions (plan, nodes, fit with and without the pole factor, eval at and beyond
        // the edges of the range, check) and a small image through the table on the host build of the kernel logic, against Kerr
        const double M = 1.0, a = 0.9, rh = M + std::sqrt(M * M - a * a);
        orc_config c;
        std::memset(&c, 0, sizeof c);
        c.metric_id = ORC_METRIC_KERR; c.params[0] = M; c.params[1] = a;
        c.r_inner = 1.01 * rh; c.r_outer = 2000.0;
        c.disc_id = ORC_DISC_THIN; c.disc_r_in = 2.0; c.disc_r_out = 40.0;
        c.gtol = 1e-2; c.lambda0 = 0.0; c.lambda1 = 200.0; c.abstol = c.reltol = 1e-9; c.maxiters = 1000000; c.hemi_delta = 1e-4; c.winding_plane = PI / 2;
        for (int pole = 2; pole >= 0; --pole) {
            gr_metric_grid grid;
            // (form 2 -- axis terms -- on a grid with two break radii, one of them with a scale: five segments, cores, cut rows)
            const gr_metric_break brk[2] = { { 9.0, 0.0 }, { 20.0, 1e-2 } };
            if (gr_metric_grid_plan_breaks(c.r_inner * 0.999, c.r_outer, rh * 0.999, 8, 32, pole == 2 ? 2 : 0, pole == 2 ? brk : nullptr, &grid) != GR_OK) { std::printf("grid plan failed\n"); ++bad; break; }
            if (pole == 2 && grid.n_seg != 4) { std::printf("expected 4 segments, got %d\n", grid.n_seg); ++bad; }
            grid.pole_factor = pole;
            std::vector<double> rn(grid.n_r_nodes), tn(grid.n_theta_nodes), samples((size_t)grid.n_r_nodes * grid.n_theta_nodes * 5);
            gr_metric_grid_nodes(&grid, rn.data(), tn.data());
            for (int64_t i = 0; i < grid.n_r_nodes; ++i)
                for (int64_t j = 0; j < grid.n_theta_nodes; ++j) { double d1[5], d2[5]; orc_metric_jacobian(&c, rn[i], tn[j], &samples[(size_t)(i * grid.n_theta_nodes + j) * 5], d1, d2); }
            std::vector<double> table(grid.table_doubles);
            double err[3];
            if (gr_metric_table_fit(&grid, samples.data(), table.data(), err) != GR_OK) { std::printf("table fit failed\n"); ++bad; break; }
            double worst = 0.0;
            const double pts[][2] = { { 3.0, 1.0 }, { c.r_inner, 0.01 }, { 1999.0, PI - 0.01 }, { 5.0, -0.3 }, { 5.0, PI + 0.3 }, { 0.5 * c.r_inner, 1.0 }, { 5000.0, 1.0 },
                                      { 8.9999999, 0.7 }, { 9.0, 0.7 }, { 19.999, 2.0 }, { 20.0, 2.0 }, { 20.0001, 2.0 }, { 15.0, 1e-4 }, { INFINITY, 1.0 }, { NAN, 1.0 } };
            for (const auto& pt : pts) {
                double g[5], dr[5], dth[5], ref[5];
                if (gr_metric_table_eval(table.data(), grid.table_doubles, pt[0], pt[1], g, dr, dth) != GR_OK) { ++bad; continue; }
                if (!(pt[0] >= c.r_inner * 0.999) || pt[0] > c.r_outer) {          // outside the range (or NaN): the nearest patch, any FINITE value
                    for (int k = 0; k < 5; ++k) if (!std::isfinite(g[k]) || !std::isfinite(dr[k])) ++bad;
                    continue;
                }
                { double d1[5], d2[5]; orc_metric_jacobian(&c, pt[0], std::fabs(pt[1] > PI ? 2 * PI - pt[1] : pt[1]), ref, d1, d2); }
                for (int k = 0; k < 5; ++k) worst = std::fmax(worst, std::fabs(g[k] - ref[k]) / std::fmax(std::fabs(ref[k]), 1e-3));
            }
            std::printf("tabulated Kerr, pole factor %d: estimates %.1e %.1e %.1e, worst component error at the probes %.1e\n", pole, err[0], err[1], err[2], worst);
            // (as sampled, g_ϕϕ next to the axis -- the probe at θ = 1e-4 -- has an absolute accuracy only: what forms 1 and 2 are for)
            if (worst > (pole ? 1e-6 : 1e-4) || gr_metric_table_check(table.data(), grid.table_doubles) != GR_OK) ++bad;
            if (gr_metric_table_check(table.data(), grid.table_doubles - 1) == GR_OK || gr_metric_table_check(nullptr, 0) == GR_OK) ++bad;
            if (pole != 1) continue;
            gr_config g, gt;
            std::memset(&g, 0, sizeof g);
            std::memcpy(&g, &c, sizeof c);
            gt = g;
            gt.metric_id = GR_METRIC_TABULATED; gt.metric_table = table.data(); gt.metric_table_n = grid.table_doubles;
            gr_plane pl;
            std::memset(&pl, 0, sizeof pl);
            std::memcpy(pl.x_obs, x, sizeof x);
            orc_lnr_transform(&c, x, pl.Mx);
            pl.alpha0 = -9.5; pl.alpha1 = 9.5; pl.beta0 = -9.5; pl.beta1 = 9.5; pl.width = W; pl.height = H; pl.offset = 1e-6;
            const gr_range rg{ 0, N, N, 1 };
            std::vector<gr_point> pa(N), pb(N);
            hh_render_endpoints(&g, &pl, &rg, pa.data());
            hh_render_endpoints(&gt, &pl, &rg, pb.data());
            int mism = 0;
            double w2 = 0.0;
            for (int i = 0; i < N; ++i) {
                if (pa[i].status != pb[i].status) { ++mism; continue; }
                if (pa[i].status == 1) continue;
                for (int k = 1; k < 3; ++k) w2 = std::fmax(w2, std::fabs(pa[i].x[k] - pb[i].x[k]) / std::fmax(1.0, std::fabs(pa[i].x[k])));
            }
            std::printf("tabulated Kerr through the host kernel logic: status mismatches %d / %d, worst end-point difference %.2e (grid 8 x 32)\n", mism, N, w2);
            if (mism > 3 || w2 > 1e-4) ++bad;
        }
    }
    {   // plunging table (mu = 1 trace with every step saved)
        orc_config c;
        std::memset(&c, 0, sizeof c);
        c.metric_id = ORC_METRIC_JOHANNSEN; c.params[0] = 1.0; c.params[1] = 0.7; c.params[2] = 2.0; c.params[5] = 1.0;
        c.r_inner = 1.000001 * (1.0 + std::sqrt(1.0 - 0.49)); c.r_outer = 12000.0; c.lambda1 = 50000.0; c.abstol = c.reltol = 1e-9;
        c.maxiters = 1000000; c.mu = 1.0; c.gtol = 1e-2;
        const double isco = orc_isco(&c);
        std::vector<double> r(4096), vt(4096), vr(4096), vp(4096);
        const long long n = orc_plunging_table(&c, isco, r.data(), vt.data(), vr.data(), vp.data(), 4096);
        std::printf("plunging table: isco %.6f, %lld rows\n", isco, n);
        if (n < 50) ++bad;
    }
    std::printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 1 : 0;
}

// metric_table.hip -- host side of GR_METRIC_TABULATED (include/gradus_mi355x.h, "tabulated metrics"): plan a patch grid, name
// its sample nodes, fit the caller's samples of metric_components(m, (r, θ)) and evaluate a table at a point.  No device code,
// no context: these run wherever the library loads.  The kernels' side is TabulatedMetric (gr_device.hpp); both read a patch
// through gr_tab::locate / gr_tab::eval_patch (gr_tabmetric.hpp), so gr_metric_table_eval IS the device's arithmetic.
#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gradus_mi355x.h"
#include "gr_tabmetric.hpp"

int32_t gr_set_last_error(int32_t code, const char* msg);      // gradus_mi355x.hip

namespace {

using namespace gr_tab;

std::atomic<uint64_t> g_build_counter{ 1 };

// radial patch ir of the grid -> [ra, rb]
void radial_patch(const gr_metric_grid& g, int ir, double& ra, double& rb)
{
    const int e = g.e_min + ir / g.m_r, j = ir % g.m_r;
    const double s = std::ldexp(1.0, e);
    ra = g.r0 + s * (1.0 + (double)j / g.m_r);
    rb = g.r0 + s * (1.0 + (double)(j + 1) / g.m_r);
}

bool grid_ok(const gr_metric_grid* g)
{
    return g && g->m_r >= 1 && g->m_r <= 1024 && g->n_theta >= 1 && g->n_theta <= 4096 && g->n_oct >= 1 && g->n_oct <= 64
           && g->degree == kDegree && g->fit_nodes == kFitNodes && g->e_min > -1000 && g->e_min < 1000
           && (g->pole_factor == 0 || g->pole_factor == 1)
           && g->n_r_nodes == (int64_t)g->n_oct * g->m_r * kFitNodes && g->n_theta_nodes == (int64_t)g->n_theta * kFitNodes
           && g->table_doubles == kHeaderDoubles + (int64_t)g->n_oct * g->m_r * g->n_theta * kPatchDoubles;
}

// Chebyshev machinery for N nodes x_k = cos(π (k + ½) / N): W[i][k] maps samples to coefficients, T[n][m] = coefficient of x^m in T_n
struct Cheb {
    double x[kFitNodes];
    double W[kFitNodes][kFitNodes];
    double T[kDegree + 1][kDegree + 1];
    Cheb()
    {
        const int N = kFitNodes;
        for (int k = 0; k < N; ++k) x[k] = std::cos(M_PI * (k + 0.5) / N);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < N; ++k) W[i][k] = (i == 0 ? 1.0 : 2.0) / N * std::cos(M_PI * i * (k + 0.5) / N);
        std::memset(T, 0, sizeof T);
        T[0][0] = 1.0;
        if (kDegree >= 1) T[1][1] = 1.0;
        for (int n = 2; n <= kDegree; ++n)
            for (int m = 0; m <= n; ++m) T[n][m] = (m > 0 ? 2.0 * T[n - 1][m - 1] : 0.0) - T[n - 2][m];
    }
};
const Cheb& cheb()
{
    static const Cheb c;
    return c;
}

}  // namespace

extern "C" {

int32_t gr_metric_grid_plan(double r_min, double r_max, double r0, int32_t m_r, int32_t n_theta, gr_metric_grid* grid)
{
    if (!grid) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "grid is null");
    if (!(r_min > r0) || !(r_max > r_min) || !std::isfinite(r_max) || !std::isfinite(r0))
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a metric grid needs r0 < r_min < r_max");
    if (m_r < 1 || m_r > 1024 || n_theta < 1 || n_theta > 4096)
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a metric grid needs 1 <= m_r <= 1024 and 1 <= n_theta <= 4096");
    int e_lo, e_hi;
    (void)std::frexp(r_min - r0, &e_lo);      // r_min - r0 = f 2^e_lo, f in [0.5, 1): octave e_lo - 1
    (void)std::frexp(r_max - r0, &e_hi);
    e_lo -= 1;
    e_hi -= 1;
    if (std::ldexp(1.0, e_hi) == r_max - r0) e_hi -= 1;      // r_max on an octave boundary: the octave below ends there
    if (e_hi < e_lo) e_hi = e_lo;
    if (e_hi - e_lo + 1 > 64) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a metric grid spans at most 64 octaves of r - r0: move r0 away from r_min");
    std::memset(grid, 0, sizeof *grid);
    grid->r0 = r0;
    grid->r_min = r_min;
    grid->r_max = r_max;
    grid->e_min = e_lo;
    grid->n_oct = e_hi - e_lo + 1;
    grid->m_r = m_r;
    grid->n_theta = n_theta;
    grid->degree = kDegree;
    grid->fit_nodes = kFitNodes;
    grid->pole_factor = 1;
    grid->n_r_nodes = (int64_t)grid->n_oct * m_r * kFitNodes;
    grid->n_theta_nodes = (int64_t)n_theta * kFitNodes;
    grid->table_doubles = kHeaderDoubles + (int64_t)grid->n_oct * m_r * n_theta * kPatchDoubles;
    return GR_OK;
}

int32_t gr_metric_grid_nodes(const gr_metric_grid* grid, double* r_nodes, double* theta_nodes)
{
    if (!grid_ok(grid) || !r_nodes || !theta_nodes) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "bad metric grid (use gr_metric_grid_plan) or null node arrays");
    const Cheb& cb = cheb();
    const int N = kFitNodes;
    for (int ir = 0; ir < grid->n_oct * grid->m_r; ++ir) {
        double ra, rb;
        radial_patch(*grid, ir, ra, rb);
        const double mid = 0.5 * (ra + rb), half = 0.5 * (rb - ra);
        for (int k = 0; k < N; ++k) r_nodes[(int64_t)ir * N + k] = mid + half * cb.x[k];
    }
    for (int it = 0; it < grid->n_theta; ++it) {
        const double ta = M_PI * it / grid->n_theta, tb = M_PI * (it + 1) / grid->n_theta;
        const double mid = 0.5 * (ta + tb), half = 0.5 * (tb - ta);
        for (int k = 0; k < N; ++k) theta_nodes[(int64_t)it * N + k] = mid + half * cb.x[k];
    }
    return GR_OK;
}

int32_t gr_metric_table_fit(const gr_metric_grid* grid, const double* samples, double* table, double* err)
{
    if (!grid_ok(grid) || !samples || !table) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "bad metric grid (use gr_metric_grid_plan) or null samples / table");
    const Cheb& cb = cheb();
    const int N = kFitNodes, p = kDegree;
    const int64_t nth_nodes = grid->n_theta_nodes;
    const int n_pr = grid->n_oct * grid->m_r;
    double e_val = 0.0, e_dr = 0.0, e_dt = 0.0;
    std::memset(table, 0, sizeof(double) * (size_t)grid->table_doubles);
    // 1 / sin²θ at the θ nodes (pole_factor: g_ϕϕ and g_tϕ are fitted without the factor they share on the axis)
    std::vector<double> inv_s2((size_t)nth_nodes, 1.0);
    if (grid->pole_factor) {
        std::vector<double> rn((size_t)grid->n_r_nodes), tn((size_t)nth_nodes);
        (void)gr_metric_grid_nodes(grid, rn.data(), tn.data());
        for (int64_t b = 0; b < nth_nodes; ++b) {
            const double sn = std::sin(tn[(size_t)b]);
            inv_s2[(size_t)b] = 1.0 / (sn * sn);
        }
    }
    for (int ir = 0; ir < n_pr; ++ir) {
        double ra, rb;
        radial_patch(*grid, ir, ra, rb);
        // the estimate of the radial derivative's error is quoted per unit of ln(r - r0): d/du -> (centre - r0) / half-width
        const double log_scale = (0.5 * (ra + rb) - grid->r0) / (0.5 * (rb - ra));
        const double th_scale = 2.0 * grid->n_theta / M_PI;
        for (int it = 0; it < grid->n_theta; ++it) {
            double* patch = table + kHeaderDoubles + ((int64_t)ir * grid->n_theta + it) * kPatchDoubles;
            double F[kComps][kFitNodes][kFitNodes], fmax[kComps];
            for (int k = 0; k < kComps; ++k) fmax[k] = 0.0;
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b) {
                    const double* s = samples + (((int64_t)ir * N + a) * nth_nodes + (int64_t)it * N + b) * kComps;
                    for (int k = 0; k < kComps; ++k) {
                        if (!std::isfinite(s[k])) {
                            const std::string msg = "metric samples must be finite (the sample at r node " + std::to_string((int64_t)ir * N + a)
                                                    + ", θ node " + std::to_string((int64_t)it * N + b) + " is not)";
                            return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, msg.c_str());
                        }
                        const double val = k >= 3 ? s[k] * inv_s2[(size_t)((int64_t)it * N + b)] : s[k];
                        F[k][a][b] = val;
                        fmax[k] = std::fmax(fmax[k], std::fabs(val));
                    }
                }
            // a vanishing (or round-off sized) g_tϕ is measured against the block it couples
            const double tp_floor = 1e-2 * std::sqrt(fmax[0] * fmax[3]);
            for (int k = 0; k < kComps; ++k) {
                // Chebyshev coefficients A = W F Wᵀ
                double G[kFitNodes][kFitNodes], A[kFitNodes][kFitNodes];
                for (int i = 0; i < N; ++i)
                    for (int b = 0; b < N; ++b) {
                        double acc = 0.0;
                        for (int a = 0; a < N; ++a) acc += cb.W[i][a] * F[k][a][b];
                        G[i][b] = acc;
                    }
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j) {
                        double acc = 0.0;
                        for (int b = 0; b < N; ++b) acc += G[i][b] * cb.W[j][b];
                        A[i][j] = acc;
                    }
                // what the truncation to total degree p drops
                double d0 = 0.0, d1 = 0.0, d2 = 0.0;
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j)
                        if (i + j > p) {
                            const double a = std::fabs(A[i][j]);
                            d0 += a;
                            d1 += a * i * i;
                            d2 += a * j * j;
                        }
                double scale = fmax[k];
                if (k == 4) scale = std::fmax(scale, tp_floor);
                if (scale > 0.0) {
                    e_val = std::fmax(e_val, d0 / scale);
                    e_dr = std::fmax(e_dr, d1 * log_scale / scale);
                    e_dt = std::fmax(e_dt, d2 * th_scale / scale);
                }
                // monomials: c[m1][m2] = Σ_{i + j <= p} A[i][j] T[i][m1] T[j][m2]
                double c[kDegree + 1][kDegree + 1];
                for (int m1 = 0; m1 <= p; ++m1)
                    for (int m2 = 0; m2 <= p; ++m2) {
                        double acc = 0.0;
                        for (int i = m1; i <= p; ++i)
                            for (int j = m2; i + j <= p; ++j) acc += A[i][j] * cb.T[i][m1] * cb.T[j][m2];
                        c[m1][m2] = acc;
                    }
                // rows i = p .. 0, inside a row j = p - i .. 0: the order eval_patch consumes
                double* out = patch + k * kCoefs;
                for (int i = p; i >= 0; --i)
                    for (int t = 0; t <= p - i; ++t) out[row_offset(i) + t] = c[i][(p - i) - t];
            }
        }
    }
    double* h = table;
    h[H_MAGIC] = kMagic;
    h[H_DEGREE] = kDegree;
    h[H_R0] = grid->r0;
    h[H_EMIN] = grid->e_min;
    h[H_NOCT] = grid->n_oct;
    h[H_MR] = grid->m_r;
    h[H_NTHETA] = grid->n_theta;
    h[H_STRIDE] = kPatchDoubles;
    // distinguishes this table from every other one this process has fitted (the contexts' device copies are keyed by it) and,
    // through a digest of the coefficients, from tables of other processes
    uint64_t dig = 1469598103934665603ull;
    for (int64_t i = kHeaderDoubles; i < grid->table_doubles; i += 97) {
        uint64_t bits;
        std::memcpy(&bits, table + i, 8);
        dig = (dig ^ bits) * 1099511628211ull;
    }
    h[H_BUILD_ID] = (double)(((g_build_counter.fetch_add(1) & 0xFFFFF) << 32) | (dig & 0xFFFFFFFFull));
    h[H_ERR_VAL] = e_val;
    h[H_ERR_DR] = e_dr;
    h[H_ERR_DTH] = e_dt;
    h[H_RMIN] = grid->r_min;
    h[H_RMAX] = grid->r_max;
    h[H_POLE_FACTOR] = grid->pole_factor;
    if (err) { err[0] = e_val; err[1] = e_dr; err[2] = e_dt; }
    return GR_OK;
}

}  // extern "C"

// shared with gradus_mi355x.hip (validate_cfg): is this a table gr_metric_table_fit wrote, of the length the caller states?
int32_t gr_metric_table_check(const double* table, int64_t table_n)
{
    if (!table || table_n < kHeaderDoubles) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "GR_METRIC_TABULATED needs cfg.metric_table (from gr_metric_table_fit) and its length");
    if (table[H_MAGIC] != kMagic || table[H_DEGREE] != (double)kDegree || table[H_STRIDE] != (double)kPatchDoubles)
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "cfg.metric_table is not a table written by this library's gr_metric_table_fit");
    const double n_oct = table[H_NOCT], m_r = table[H_MR], n_theta = table[H_NTHETA];
    if (!(n_oct >= 1 && n_oct <= 64 && m_r >= 1 && m_r <= 1024 && n_theta >= 1 && n_theta <= 4096)
        || (double)table_n != kHeaderDoubles + n_oct * m_r * n_theta * kPatchDoubles)
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "cfg.metric_table_n does not match the table's header");
    return GR_OK;
}

extern "C" int32_t gr_metric_table_eval(const double* table, int64_t table_n, double r, double theta, double* g, double* dr, double* dth)
{
    const int32_t rc = gr_metric_table_check(table, table_n);
    if (rc != GR_OK) return rc;
    if (!g || !dr || !dth) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "null output");
    int patch;
    double u, v, su, sv;
    locate(make_gridk(table[H_R0], (int)table[H_EMIN], (int)table[H_NOCT], (int)table[H_MR], (int)table[H_NTHETA]), r, theta, patch, u, v, su, sv);
    const double* pc = table + kHeaderDoubles + (int64_t)patch * kPatchDoubles;
    double P[kComps], Pu[kComps], Pv[kComps];
    eval_patch<double>([pc](int k) { return pc[k]; }, HostOps{}, u, v, P, Pu, Pv);
    for (int k = 0; k < kComps; ++k) {
        g[k] = P[k];
        dr[k] = Pu[k] * su;
        dth[k] = Pv[k] * sv;
    }
    if (table[H_POLE_FACTOR] != 0.0) {
        const double sn = std::sin(theta), cs = std::cos(theta);
        pole_factor_apply(sn * sn, 2.0 * sn * cs, g, dr, dth);
    }
    return GR_OK;
}

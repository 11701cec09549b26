// metric_table.hip -- host side of GR_METRIC_TABULATED (include/gradus_mi355x.h, "tabulated metrics"): plan a patch grid, name
// its sample nodes, fit the caller's samples of metric_components(m, (r, θ)) and evaluate a table at a point.  No device code,
// no context: these run wherever the library loads.  The kernels' side is TabulatedMetricT (gr_device.hpp); both read a patch
// through gr_tab::locate* / gr_tab::eval_patch (gr_tabmetric.hpp), so gr_metric_table_eval IS the device's arithmetic.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/gradus_mi355x.h"
#include "gr_tabmetric.hpp"

int32_t gr_set_last_error(int32_t code, const char* msg);      // gradus_mi355x.hip

namespace {

using namespace gr_tab;

static_assert(GR_METRIC_MAX_SEG == kMaxSeg, "header and gr_tabmetric.hpp disagree on the number of segments");

std::atomic<uint64_t> g_build_counter{ 1 };
constexpr double kInf = std::numeric_limits<double>::infinity();

int64_t table_doubles_of(int n_rows, int n_theta) { return kBodyOff + ((int64_t)n_rows * kAxisDoubles + 7) / 8 * 8 + (int64_t)n_rows * n_theta * kPatchDoubles; }
int64_t axis_off() { return kBodyOff; }
int64_t patch_off(int n_rows) { return kBodyOff + ((int64_t)n_rows * kAxisDoubles + 7) / 8 * 8; }

bool grid_ok(const gr_metric_grid* g)
{
    if (!(g && g->m_r >= 1 && g->m_r <= 1024 && g->n_theta >= 1 && g->n_theta <= 4096 && g->n_oct >= 1 && g->n_oct <= 64
          && g->degree == kDegree && g->fit_nodes == kFitNodes && g->e_min > -1000 && g->e_min < 1000
          && g->pole_factor >= 0 && g->pole_factor <= 2 && g->n_seg >= 1 && g->n_seg <= kMaxSeg && g->n_rows >= 1))
        return false;
    int rows = 0;
    for (int s = 0; s < g->n_seg; ++s) {
        const gr_metric_segment& q = g->seg[s];
        if (q.first_row != rows || q.e_hi < q.e_lo || q.e_hi - q.e_lo > 63 || q.e_lo < -1000 || q.e_hi > 1000 || (q.dir != 1 && q.dir != -1)
            || (q.core != 0 && q.core != 1) || q.n_rows != (q.core + q.e_hi - q.e_lo + 1) * g->m_r || !(q.r_hi > q.r_lo))
            return false;
        rows += q.n_rows;
    }
    return rows == g->n_rows && g->n_r_nodes == (int64_t)g->n_rows * kFitNodes && g->n_theta_nodes == (int64_t)g->n_theta * kFitNodes
           && g->table_doubles == table_doubles_of(g->n_rows, g->n_theta) && g->seg[0].core == 0 && g->seg[0].dir == 1
           && g->seg[0].anchor == g->r0 && g->seg[0].e_lo == g->e_min && g->seg[0].e_hi == g->e_min + g->n_oct - 1;
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

// octave of a positive length y as the grid counts them: y in (2^e, 2^(e+1)] -- a length that IS a power of two ends the octave below
int octave_ending_at(double y)
{
    int e;
    (void)std::frexp(y, &e);
    e -= 1;
    if (std::ldexp(1.0, e) == y) e -= 1;
    return e;
}

// One radial row: its patch [pa, pb] and the interval [fa, fb] its polynomial is fitted on, both in the segment's x = ±(r - anchor).
// The two differ at a "hard" end of a segment -- a radius where the metric's functions change form: a row there is fitted on its
// part inside only (and re-expanded about the patch's own centre), and a row with less than a quarter of itself inside shares the
// fit of its neighbour -- one polynomial over both, each row in its own coordinates.
struct RowGeom {
    int seg;
    double pa, pb, fa, fb;
};
bool row_geometry(const gr_metric_grid& g, std::vector<RowGeom>& rows, std::string& why)
{
    rows.assign((size_t)g.n_rows, RowGeom{});
    for (int s = 0; s < g.n_seg; ++s) {
        const gr_metric_segment& q = g.seg[s];
        // where the fit may sample, in x
        double hx_lo = q.dir > 0 ? q.fit_lo - q.anchor : q.anchor - q.fit_hi;
        double hx_hi = q.dir > 0 ? q.fit_hi - q.anchor : q.anchor - q.fit_lo;
        if (!(hx_lo > 0.0)) hx_lo = q.core ? 0.0 : -kInf;
        std::vector<double> ia((size_t)q.n_rows), ib((size_t)q.n_rows);
        std::vector<char> good((size_t)q.n_rows, 0);
        bool any = false;
        for (int k = 0; k < q.n_rows; ++k) {
            const int oct = k / g.m_r, j = k % g.m_r;
            double pa, pb;
            if (q.core && oct == 0) {
                pa = q.xmin * j / g.m_r;
                pb = q.xmin * (j + 1) / g.m_r;
            } else {
                const double sc = std::ldexp(1.0, q.e_lo + oct - q.core);
                pa = sc * (1.0 + (double)j / g.m_r);
                pb = sc * (1.0 + (double)(j + 1) / g.m_r);
            }
            RowGeom& r = rows[(size_t)(q.first_row + k)];
            r.seg = s;
            r.pa = pa;
            r.pb = pb;
            ia[(size_t)k] = std::max(pa, hx_lo);
            ib[(size_t)k] = std::min(pb, hx_hi);
            good[(size_t)k] = (ib[(size_t)k] - ia[(size_t)k]) >= 0.25 * (pb - pa);
            any = any || good[(size_t)k];
        }
        if (!any) {
            why = "segment " + std::to_string(s) + " [" + std::to_string(q.r_lo) + ", " + std::to_string(q.r_hi) + ") holds no whole patch: break radii too close for this grid";
            return false;
        }
        for (int k = 0; k < q.n_rows; ++k) {
            RowGeom& r = rows[(size_t)(q.first_row + k)];
            if (good[(size_t)k]) {
                r.fa = ia[(size_t)k];
                r.fb = ib[(size_t)k];
                continue;
            }
            int n = -1;
            for (int d = 1; d < q.n_rows && n < 0; ++d) {
                if (k - d >= 0 && good[(size_t)(k - d)]) n = k - d;
                else if (k + d < q.n_rows && good[(size_t)(k + d)]) n = k + d;
            }
            r.fa = ia[(size_t)n];
            r.fb = ib[(size_t)n];
            if (ib[(size_t)k] > ia[(size_t)k] && (n == k - 1 || n == k + 1)) {      // a sliver next to a good row: one fit over both
                r.fa = std::min(r.fa, ia[(size_t)k]);
                r.fb = std::max(r.fb, ib[(size_t)k]);
            }
        }
    }
    return true;
}

void fill_counts(gr_metric_grid* g)
{
    int rows = 0;
    for (int s = 0; s < g->n_seg; ++s) {
        g->seg[s].first_row = rows;
        g->seg[s].n_rows = (g->seg[s].core + g->seg[s].e_hi - g->seg[s].e_lo + 1) * g->m_r;
        g->seg[s].xmin = std::ldexp(1.0, g->seg[s].e_lo);
        rows += g->seg[s].n_rows;
    }
    g->n_rows = rows;
    g->degree = kDegree;
    g->fit_nodes = kFitNodes;
    g->pole_factor = 1;
    g->n_r_nodes = (int64_t)rows * kFitNodes;
    g->n_theta_nodes = (int64_t)g->n_theta * kFitNodes;
    g->table_doubles = table_doubles_of(rows, g->n_theta);
    g->e_min = g->seg[0].e_lo;
    g->n_oct = g->seg[0].e_hi - g->seg[0].e_lo + 1;
}

}  // namespace

extern "C" {

int32_t gr_metric_grid_plan_breaks(double r_min, double r_max, double r0, int32_t m_r, int32_t n_theta, int32_t n_breaks,
                                   const gr_metric_break* breaks, gr_metric_grid* grid)
{
    if (!grid) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "grid is null");
    if (!(r_min > r0) || !(r_max > r_min) || !std::isfinite(r_max) || !std::isfinite(r0))
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a metric grid needs r0 < r_min < r_max");
    if (m_r < 1 || m_r > 1024 || n_theta < 1 || n_theta > 4096)
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a metric grid needs 1 <= m_r <= 1024 and 1 <= n_theta <= 4096");
    if (n_breaks < 0 || (n_breaks > 0 && !breaks)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "n_breaks < 0 or breaks is null");
    std::vector<gr_metric_break> bs(breaks, breaks + n_breaks);
    std::sort(bs.begin(), bs.end(), [](const gr_metric_break& a, const gr_metric_break& b) { return a.radius < b.radius; });
    for (int k = 0; k < n_breaks; ++k) {
        if (!(bs[(size_t)k].radius > r_min) || !(bs[(size_t)k].radius < r_max) || !(bs[(size_t)k].scale >= 0.0) || !std::isfinite(bs[(size_t)k].scale))
            return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a break radius must lie strictly inside (r_min, r_max) and its scale be finite and >= 0");
        if (k > 0 && !(bs[(size_t)k].radius > bs[(size_t)(k - 1)].radius)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "break radii must be distinct");
    }
    std::memset(grid, 0, sizeof *grid);
    grid->r0 = r0;
    grid->r_min = r_min;
    grid->r_max = r_max;
    grid->m_r = m_r;
    grid->n_theta = n_theta;
    int ns = 0;
    auto push = [&](gr_metric_segment q) -> bool {
        if (ns >= kMaxSeg) return false;
        grid->seg[ns++] = q;
        return true;
    };
    // the points between which the metric is smooth: r_min | breaks | r_max (the two ends are "soft": nothing changes form there)
    for (int k = 0; k <= n_breaks; ++k) {
        const double lo = k == 0 ? r_min : bs[(size_t)(k - 1)].radius, hi = k == n_breaks ? r_max : bs[(size_t)k].radius;
        const double sc_lo = k == 0 ? 0.0 : bs[(size_t)(k - 1)].scale, sc_hi = k == n_breaks ? 0.0 : bs[(size_t)k].scale;
        const double fit_lo = k == 0 ? -kInf : lo, fit_hi = k == n_breaks ? kInf : hi;
        const double L = hi - lo;
        // a feature centred at the upper end: geometric patches towards it over the upper part [hi - W, hi) of the interval
        const double W = sc_hi > 0.0 ? std::ldexp(1.0, (int)std::floor(std::log2(0.5 * L))) : 0.0;
        const double top = hi - W;
        gr_metric_segment q{};
        q.r_lo = lo;
        q.r_hi = top;
        q.fit_lo = fit_lo;
        q.fit_hi = fit_hi;
        q.dir = 1;
        if (k == 0) {      // anchored just inside the horizon
            q.anchor = r0;
            q.core = 0;
            int e_lo;
            (void)std::frexp(r_min - r0, &e_lo);      // r_min - r0 = f 2^e_lo, f in [0.5, 1): octave e_lo - 1
            q.e_lo = e_lo - 1;
            q.e_hi = std::max(q.e_lo, octave_ending_at(top - r0));
        } else {           // anchored at the break it starts from
            q.anchor = lo;
            q.core = 1;
            const double Lp = top - lo;
            int e_main;
            (void)std::frexp(lo - r0, &e_main);
            const int e_len = (int)std::floor(std::log2(Lp));
            q.e_lo = sc_lo > 0.0 ? (int)std::floor(std::log2(sc_lo)) : std::min(e_main - 2, e_len);
            q.e_lo = std::min(q.e_lo, e_len);
            q.e_hi = std::max(q.e_lo, octave_ending_at(Lp));
        }
        if (q.e_hi - q.e_lo + 1 > 64)
            return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a metric grid spans at most 64 octaves per segment: move r0 away from r_min, or raise a break's scale");
        if (!push(q)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "too many break radii: a metric grid has at most GR_METRIC_MAX_SEG segments");
        if (W > 0.0) {
            gr_metric_segment c{};
            c.r_lo = top;
            c.r_hi = hi;
            c.fit_lo = fit_lo;
            c.fit_hi = fit_hi;
            c.anchor = hi;
            c.dir = -1;
            c.core = 1;
            c.e_lo = std::min((int)std::floor(std::log2(sc_hi)), octave_ending_at(W));
            c.e_hi = octave_ending_at(W);
            if (c.e_hi - c.e_lo + 1 > 64) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "a break's scale is more than 64 octaves below its interval");
            if (!push(c)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "too many break radii: a metric grid has at most GR_METRIC_MAX_SEG segments");
        }
    }
    grid->n_seg = ns;
    fill_counts(grid);
    std::vector<RowGeom> rows;
    std::string why;
    if (!row_geometry(*grid, rows, why)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, why.c_str());
    return GR_OK;
}

int32_t gr_metric_grid_plan(double r_min, double r_max, double r0, int32_t m_r, int32_t n_theta, gr_metric_grid* grid)
{
    return gr_metric_grid_plan_breaks(r_min, r_max, r0, m_r, n_theta, 0, nullptr, grid);
}

int32_t gr_metric_grid_nodes(const gr_metric_grid* grid, double* r_nodes, double* theta_nodes)
{
    if (!grid_ok(grid) || !r_nodes || !theta_nodes) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "bad metric grid (use gr_metric_grid_plan) or null node arrays");
    const Cheb& cb = cheb();
    const int N = kFitNodes;
    std::vector<RowGeom> rows;
    std::string why;
    if (!row_geometry(*grid, rows, why)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, why.c_str());
    for (int ir = 0; ir < grid->n_rows; ++ir) {
        const RowGeom& rg = rows[(size_t)ir];
        const gr_metric_segment& q = grid->seg[rg.seg];
        const double mid = 0.5 * (rg.fa + rg.fb), half = 0.5 * (rg.fb - rg.fa);
        for (int k = 0; k < N; ++k) r_nodes[(int64_t)ir * N + k] = q.anchor + q.dir * (mid + half * cb.x[k]);
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
    const int n_pr = grid->n_rows, form = grid->pole_factor;
    std::vector<RowGeom> rows;
    std::string why;
    if (!row_geometry(*grid, rows, why)) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, why.c_str());
    double e_val_all = 0.0, e_dr_all = 0.0, e_dt_all = 0.0;
    std::memset(table, 0, sizeof(double) * (size_t)grid->table_doubles);
    double* axis_tab = table + axis_off();
    double* patch_tab = table + patch_off(n_pr);
    // sin²θ, cos θ at the θ nodes (forms 1, 2: g_ϕϕ and g_tϕ are fitted without the factor they share on the axis)
    std::vector<double> tn((size_t)nth_nodes), s2((size_t)nth_nodes, 1.0), cs((size_t)nth_nodes, 0.0);
    {
        std::vector<double> rn((size_t)grid->n_r_nodes);
        (void)gr_metric_grid_nodes(grid, rn.data(), tn.data());
        for (int64_t b = 0; b < nth_nodes; ++b) {
            const double sn = std::sin(tn[(size_t)b]);
            s2[(size_t)b] = sn * sn;
            cs[(size_t)b] = std::cos(tn[(size_t)b]);
        }
    }
    // form 2: Lagrange weights that take a function of z = sin²θ from the kAx nodes nearest a pole to the pole itself.  The nodes
    // crowd towards z = 0 (Chebyshev in θ, squared), the weight of the nearest one is ~1 and the others fall off by orders of
    // magnitude: K = g(axis) comes out to a relative 1e-15 of ITSELF although the far nodes hold values r² sin²θ >> K.
    constexpr int kAx = 6;
    double wN[kAx], wS[kAx];
    int64_t bN[kAx], bS[kAx];
    {
        // nodes of polar patch 0 in ascending θ are k = N-1 .. 0 (cb.x descends); of the last patch, descending distance from π: k = 0 ..
        for (int i = 0; i < kAx; ++i) {
            bN[i] = (N - 1 - i);
            bS[i] = (int64_t)(grid->n_theta - 1) * N + i;
        }
        for (int i = 0; i < kAx; ++i) {
            double a = 1.0, b = 1.0;
            for (int j = 0; j < kAx; ++j)
                if (j != i) {
                    a *= s2[(size_t)bN[j]] / (s2[(size_t)bN[j]] - s2[(size_t)bN[i]]);
                    b *= s2[(size_t)bS[j]] / (s2[(size_t)bS[j]] - s2[(size_t)bS[i]]);
                }
            wN[i] = a;
            wS[i] = b;
        }
    }
    // binomials for the re-expansion w = α u + β of a row fitted on part of its patch (or over its neighbour as well)
    double binom[kDegree + 1][kDegree + 1];
    for (int n = 0; n <= p; ++n)
        for (int m = 0; m <= n; ++m) binom[n][m] = (m == 0 || m == n) ? 1.0 : binom[n - 1][m - 1] + binom[n - 1][m];
    // (rows are independent: a few threads share them -- 280 rows x 96 polar patches x 5 components of 12 x 12 samples each take
    // 12 s on one core)
    std::atomic<int> next_row{ 0 }, bad_r{ -1 };
    std::atomic<int64_t> bad_t{ -1 };
    std::mutex err_mutex;
    auto work = [&]() {
    double e_val = 0.0, e_dr = 0.0, e_dt = 0.0;      // this thread's maxima (shadow the call's)
    for (int ir = next_row.fetch_add(1); ir < n_pr; ir = next_row.fetch_add(1)) {
        const RowGeom& rg = rows[(size_t)ir];
        const gr_metric_segment& sg = grid->seg[rg.seg];
        const double fm = 0.5 * (rg.fa + rg.fb), fh = 0.5 * (rg.fb - rg.fa), pm = 0.5 * (rg.pa + rg.pb), ph = 0.5 * (rg.pb - rg.pa);
        const double alpha = ph / fh, beta = (pm - fm) / fh;
        const bool moved = !(alpha == 1.0 && beta == 0.0);
        double apow[kDegree + 1], bpow[kDegree + 1];
        apow[0] = bpow[0] = 1.0;
        for (int n = 1; n <= p; ++n) { apow[n] = apow[n - 1] * alpha; bpow[n] = bpow[n - 1] * beta; }
        // the estimate of the radial derivative's error is quoted per unit of ln x (x = |r - anchor|): d/dw -> centre / half-width
        const double log_scale = std::fmax(fm, sg.xmin) / fh;
        const double th_scale = 2.0 * grid->n_theta / M_PI;
        // form 2: the axis terms of this row at its radial nodes, and their polynomials in u
        double Km[2][kFitNodes], Kd[2][kFitNodes];
        if (form == 2) {
            for (int a = 0; a < N; ++a)
                for (int c = 0; c < 2; ++c) {
                    double kn = 0.0, ks = 0.0;
                    for (int i = 0; i < kAx; ++i) {
                        kn += wN[i] * samples[(((int64_t)ir * N + a) * nth_nodes + bN[i]) * kComps + 3 + c];
                        ks += wS[i] * samples[(((int64_t)ir * N + a) * nth_nodes + bS[i]) * kComps + 3 + c];
                    }
                    Km[c][a] = 0.5 * (kn + ks);
                    Kd[c][a] = 0.5 * (kn - ks);
                }
            double* ax = axis_tab + (int64_t)ir * kAxisDoubles;
            for (int c = 0; c < 2; ++c)
                for (int which = 0; which < 2; ++which) {
                    const double* f = which == 0 ? Km[c] : Kd[c];
                    double A[kFitNodes], fmax = 0.0, drop = 0.0, drop1 = 0.0;
                    for (int i = 0; i < N; ++i) {
                        double acc = 0.0;
                        for (int a = 0; a < N; ++a) acc += cb.W[i][a] * f[a];
                        A[i] = acc;
                    }
                    for (int a = 0; a < N; ++a) fmax = std::fmax(fmax, std::fabs(Km[c][a]) + std::fabs(Kd[c][a]));
                    // (axis terms that are rounding residue of a component that does vanish there -- 1e-17 for Kerr -- are measured
                    // against the component, not against themselves)
                    {
                        double gm = 0.0;
                        for (int a = 0; a < N; ++a)
                            for (int64_t b = 0; b < nth_nodes; b += 7) gm = std::fmax(gm, std::fabs(samples[(((int64_t)ir * N + a) * nth_nodes + b) * kComps + 3 + c]));
                        fmax = std::fmax(fmax, 1e-9 * gm);
                    }
                    for (int i = p + 1; i < N; ++i) { drop += std::fabs(A[i]); drop1 += std::fabs(A[i]) * i * i; }
                    if (fmax > 0.0) {
                        e_val = std::fmax(e_val, drop / fmax);
                        e_dr = std::fmax(e_dr, drop1 * log_scale / fmax);
                    }
                    double mono[kDegree + 1], out[kDegree + 1];
                    for (int m = 0; m <= p; ++m) {
                        double acc = 0.0;
                        for (int i = m; i <= p; ++i) acc += A[i] * cb.T[i][m];
                        mono[m] = acc;
                    }
                    for (int n = 0; n <= p; ++n) {
                        double acc = 0.0;
                        for (int m = n; m <= p; ++m) acc += mono[m] * binom[m][n] * apow[n] * bpow[m - n];
                        out[n] = acc;
                    }
                    for (int t = 0; t <= p; ++t) ax[(2 * c + which) * (p + 1) + t] = out[p - t];      // leading coefficient first
                }
        }
        for (int it = 0; it < grid->n_theta; ++it) {
            double* patch = patch_tab + ((int64_t)ir * grid->n_theta + it) * kPatchDoubles;
            double F[kComps][kFitNodes][kFitNodes], fmax[kComps], gmax[kComps], s2max = 0.0;
            for (int k = 0; k < kComps; ++k) fmax[k] = gmax[k] = 0.0;
            for (int b = 0; b < N; ++b) s2max = std::fmax(s2max, s2[(size_t)((int64_t)it * N + b)]);
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b) {
                    const int64_t bb = (int64_t)it * N + b;
                    const double* s = samples + (((int64_t)ir * N + a) * nth_nodes + bb) * kComps;
                    for (int k = 0; k < kComps; ++k) {
                        if (!std::isfinite(s[k])) {
                            int expect = -1;
                            if (bad_r.compare_exchange_strong(expect, ir * N + a)) bad_t.store(bb);
                            return;
                        }
                        double val = s[k];
                        gmax[k] = std::fmax(gmax[k], std::fabs(val));
                        if (k >= 3 && form == 2) val = (val - Km[k - 3][a] - Kd[k - 3][a] * cs[(size_t)bb]) / s2[(size_t)bb];
                        else if (k >= 3 && form == 1) val = val / s2[(size_t)bb];
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
                // (forms 1, 2: the error of g = [K +] sin²θ h is sin²θ times the error of the stored h -- measured against g's size
                // on the patch.  Form 1 next to the axis: the same as against h; form 2: the samples of h nearest the axis carry the
                // rounding of K / sin²θ, which the product with sin²θ takes out again)
                if (k >= 3 && form == 2) {
                    // The samples of h = (g - K) / sin²θ carry the rounding of K divided by sin²θ -- 1e-16 K / 2e-7 at the node nearest
                    // the axis -- which is not truncation error (the product with sin²θ takes it out of g again) but would be
                    // counted as such, weighted with i², j²: what data wrong by η_b = 8 ulp(K) / sin²θ_b could at most contribute
                    // to the three sums comes off them.
                    double kabs = 0.0, rowsum[kFitNodes], colw[kFitNodes];
                    for (int a = 0; a < N; ++a) kabs = std::fmax(kabs, std::fabs(Km[k - 3][a]) + std::fabs(Kd[k - 3][a]));
                    for (int i = 0; i < N; ++i) {
                        double rs = 0.0, cw = 0.0;
                        for (int a = 0; a < N; ++a) rs += std::fabs(cb.W[i][a]);
                        for (int b = 0; b < N; ++b) cw += std::fabs(cb.W[i][b]) * (8.0 * 2.220446049250313e-16 * kabs / s2[(size_t)((int64_t)it * N + b)]);
                        rowsum[i] = rs;
                        colw[i] = cw;
                    }
                    double n0 = 0.0, n1 = 0.0, n2 = 0.0;
                    for (int i = 0; i < N; ++i)
                        for (int j = 0; j < N; ++j)
                            if (i + j > p) {
                                const double a = rowsum[i] * colw[j];
                                n0 += a;
                                n1 += a * i * i;
                                n2 += a * j * j;
                            }
                    d0 = std::fmax(0.0, d0 - n0);
                    d1 = std::fmax(0.0, d1 - n1);
                    d2 = std::fmax(0.0, d2 - n2);
                }
                double scale = fmax[k];
                if (k >= 3 && form != 0) scale = std::fmax(scale, gmax[k] / s2max);
                if (k == 4) scale = std::fmax(scale, tp_floor);
                if (scale > 0.0) {
                    e_val = std::fmax(e_val, d0 / scale);
                    e_dr = std::fmax(e_dr, d1 * log_scale / scale);
                    e_dt = std::fmax(e_dt, d2 * th_scale / scale);
                }
                // monomials in (w, v): c[m1][m2] = Σ_{i + j <= p} A[i][j] T[i][m1] T[j][m2]
                double c[kDegree + 1][kDegree + 1];
                for (int m1 = 0; m1 <= p; ++m1)
                    for (int m2 = 0; m2 <= p; ++m2) {
                        double acc = 0.0;
                        for (int i = m1; i <= p; ++i)
                            for (int j = m2; i + j <= p; ++j) acc += A[i][j] * cb.T[i][m1] * cb.T[j][m2];
                        c[m1][m2] = acc;
                    }
                if (moved) {      // w = α u + β: the total degree stays
                    double d[kDegree + 1][kDegree + 1];
                    for (int n = 0; n <= p; ++n)
                        for (int m2 = 0; m2 <= p; ++m2) {
                            double acc = 0.0;
                            for (int m1 = n; m1 + m2 <= p; ++m1) acc += c[m1][m2] * binom[m1][n] * apow[n] * bpow[m1 - n];
                            d[n][m2] = acc;
                        }
                    std::memcpy(c, d, sizeof c);
                }
                // rows i = p .. 0, inside a row j = p - i .. 0: the order eval_patch consumes
                double* out = patch + k * kCoefs;
                for (int i = p; i >= 0; --i)
                    for (int t = 0; t <= p - i; ++t) out[row_offset(i) + t] = c[i][(p - i) - t];
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(err_mutex);
        e_val_all = std::fmax(e_val_all, e_val);
        e_dr_all = std::fmax(e_dr_all, e_dr);
        e_dt_all = std::fmax(e_dt_all, e_dt);
    }
    };
    {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
        if ((int)nt > n_pr) nt = (unsigned)n_pr;
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    if (bad_r.load() >= 0) {
        const std::string msg = "metric samples must be finite (the sample at r node " + std::to_string(bad_r.load()) + ", θ node " + std::to_string(bad_t.load()) + " is not)";
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, msg.c_str());
    }
    const double e_val = e_val_all, e_dr = e_dr_all, e_dt = e_dt_all;
    double* h = table;
    h[H_MAGIC] = kMagic;
    h[H_DEGREE] = kDegree;
    h[H_R0] = grid->r0;
    h[H_EMIN] = grid->e_min;
    h[H_NOCT] = grid->n_oct;
    h[H_MR] = grid->m_r;
    h[H_NTHETA] = grid->n_theta;
    h[H_STRIDE] = kPatchDoubles;
    h[H_NSEG] = grid->n_seg;
    h[H_NROWS] = grid->n_rows;
    h[H_AXIS_OFF] = (double)axis_off();
    h[H_PATCH_OFF] = (double)patch_off(n_pr);
    for (int s = 0; s < grid->n_seg; ++s) {
        const gr_metric_segment& q = grid->seg[s];
        SegRec rec{};
        rec.r_lo = q.r_lo; rec.r_hi = q.r_hi; rec.anchor = q.anchor; rec.xmin = q.xmin;
        rec.e_lo = q.e_lo; rec.e_hi = q.e_hi; rec.first_row = q.first_row; rec.n_rows = q.n_rows;
        rec.dir = q.dir; rec.core = q.core;
        rec.hard_lo = std::isfinite(q.fit_lo) && q.fit_lo == q.r_lo;
        rec.hard_hi = std::isfinite(q.fit_hi) && q.fit_hi == q.r_hi;
        std::memcpy(table + kSegOff + (int64_t)s * kSegDoubles, &rec, sizeof rec);
    }
    // distinguishes this table from every other one this process has fitted (the contexts' device copies are keyed by it) and,
    // through a digest of the coefficients, from tables of other processes
    uint64_t dig = 1469598103934665603ull;
    for (int64_t i = kBodyOff; i < grid->table_doubles; i += 97) {
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
    if (!table || table_n < kBodyOff) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "GR_METRIC_TABULATED needs cfg.metric_table (from gr_metric_table_fit) and its length");
    if (table[H_MAGIC] != kMagic || table[H_DEGREE] != (double)kDegree || table[H_STRIDE] != (double)kPatchDoubles)
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "cfg.metric_table is not a table written by this library's gr_metric_table_fit");
    const double n_seg = table[H_NSEG], n_rows = table[H_NROWS], m_r = table[H_MR], n_theta = table[H_NTHETA], form = table[H_POLE_FACTOR];
    if (!(n_seg >= 1 && n_seg <= kMaxSeg && n_rows >= 1 && n_rows <= 64.0 * 1024 * kMaxSeg && m_r >= 1 && m_r <= 1024 && n_theta >= 1 && n_theta <= 4096)
        || !(form == 0.0 || form == 1.0 || form == 2.0) || (double)table_n != (double)table_doubles_of((int)n_rows, (int)n_theta)
        || table[H_PATCH_OFF] != (double)patch_off((int)n_rows) || table[H_AXIS_OFF] != (double)axis_off())
        return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "cfg.metric_table_n does not match the table's header");
    int rows = 0;
    for (int s = 0; s < (int)n_seg; ++s) {
        SegRec rec;
        std::memcpy(&rec, table + kSegOff + (int64_t)s * kSegDoubles, sizeof rec);
        if (rec.first_row != rows || rec.n_rows < 1 || (rec.dir != 1 && rec.dir != -1) || rec.n_rows != (rec.core + rec.e_hi - rec.e_lo + 1) * (int)m_r)
            return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "cfg.metric_table: inconsistent segment records");
        rows += rec.n_rows;
    }
    if (rows != (int)n_rows) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "cfg.metric_table: segment records do not add up to the table's rows");
    return GR_OK;
}

extern "C" int32_t gr_metric_table_eval(const double* table, int64_t table_n, double r, double theta, double* g, double* dr, double* dth)
{
    const int32_t rc = gr_metric_table_check(table, table_n);
    if (rc != GR_OK) return rc;
    if (!g || !dr || !dth) return gr_set_last_error(GR_ERR_INVALID_ARGUMENT, "null output");
    int row, patch;
    double u, v, su, sv;
    const GridK gk = make_gridk(table[H_R0], (int)table[H_EMIN], (int)table[H_NOCT], (int)table[H_MR], (int)table[H_NTHETA], (int)table[H_NSEG]);
    SegRec segs[kMaxSeg];
    std::memcpy(segs, table + kSegOff, sizeof(SegRec) * (size_t)gk.n_seg);
    if (gk.n_seg == 1) locate(gk, r, theta, row, patch, u, v, su, sv);
    else locate_segments(gk, (const SegRec*)segs, r, theta, row, patch, u, v, su, sv);
    const double* pc = table + (int64_t)table[H_PATCH_OFF] + (int64_t)patch * kPatchDoubles;
    double P[kComps], Pu[kComps], Pv[kComps];
    eval_patch<double>([pc](int k) { return pc[k]; }, HostOps{}, u, v, P, Pu, Pv);
    for (int k = 0; k < kComps; ++k) {
        g[k] = P[k];
        dr[k] = Pu[k] * su;
        dth[k] = Pv[k] * sv;
    }
    const int form = (int)table[H_POLE_FACTOR];
    if (form != 0) {
        const double sn = std::sin(theta), cs = std::cos(theta);
        pole_factor_apply(sn * sn, 2.0 * sn * cs, g, dr, dth);
        if (form == 2) {
            const double* ax = table + (int64_t)table[H_AXIS_OFF] + (int64_t)row * kAxisDoubles;
            double axis[8];
            for (int q = 0; q < 4; ++q) {
                double K, Ku;
                eval_axis_poly<double>([ax](int k) { return ax[k]; }, q * (kDegree + 1), u, K, Ku);
                axis[2 * q] = K;
                axis[2 * q + 1] = Ku * su;
            }
            axis_terms_apply(axis, sn, cs, g, dr, dth);
        }
    }
    return GR_OK;
}

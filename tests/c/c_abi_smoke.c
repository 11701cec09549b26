/* A plain-C client of libgradus_mi355x.so: proves that include/gradus_mi355x.h is a C header, that
 * the struct layouts are what the bindings assume, and (on a GPU box) that the library can be driven
 * without Python.  Renders the reference's 20 x 20 Schwarzschild shadow fingerprint
 * (test/smoke-tests/rendergeodesics.jl:16-44: Σ λ_max = 9009.452876609641).
 *
 *   gcc -std=c11 -Iinclude tests/c/c_abi_smoke.c -ldl -lm -o c_abi_smoke && ./c_abi_smoke <path to .so>
 * exit code 0 = rendered and checked, 3 = library loaded but no device (expected on a CPU-only host). */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "gradus_mi355x.h"

_Static_assert(sizeof(gr_point) == 152, "GeodesicPoint layout");
_Static_assert(sizeof(gr_range) == 32, "gr_range");
_Static_assert(sizeof(gr_plane) == 8 * (4 + 16 + 4) + 16 + 8, "gr_plane");
_Static_assert(sizeof(gr_stats) == 96, "gr_stats");
_Static_assert(sizeof(gr_pointfunction) == 8 + 16 + 8 + 32 + 8 + 32, "gr_pointfunction (ABI 7: has_u_src, u_src)");
_Static_assert(sizeof(gr_rayset) == 8 * (4 + 16) + 8 * 4 + 8 + 8 * 5 + 8 + 24 + 16 + 8 + 8 + 16 + 8, "gr_rayset (ABI 7: sky_*; ABI 8: sky_first, sky_total, sky_rows)");
_Static_assert(sizeof(gr_metric_segment) == 6 * 8 + 6 * 4, "gr_metric_segment (ABI 8)");
_Static_assert(sizeof(gr_metric_break) == 16, "gr_metric_break (ABI 8)");
_Static_assert(sizeof(gr_metric_grid) == 24 + 8 * 4 + 24 + 8 + GR_METRIC_MAX_SEG * sizeof(gr_metric_segment), "gr_metric_grid (ABI 8: n_seg, n_rows, seg[])");

typedef int32_t (*ctx_create_t)(int32_t, gr_ctx**);
typedef int32_t (*ctx_destroy_t)(gr_ctx*);
typedef int32_t (*render_t)(gr_ctx*, const gr_config*, const gr_plane*, const gr_pointfunction*, const gr_range*, double*,
                            gr_stats*);
typedef const char* (*last_error_t)(void);
typedef int32_t (*abi_t)(void);

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s libgradus_mi355x.so\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    abi_t abi = (abi_t)dlsym(h, "gr_abi_version");
    ctx_create_t create = (ctx_create_t)dlsym(h, "gr_ctx_create");
    ctx_destroy_t destroy = (ctx_destroy_t)dlsym(h, "gr_ctx_destroy");
    render_t render = (render_t)dlsym(h, "gr_render");
    last_error_t last_error = (last_error_t)dlsym(h, "gr_last_error");
    if (!abi || !create || !destroy || !render || !last_error) { fprintf(stderr, "missing symbol\n"); return 2; }
    if (abi() != GR_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", abi(), GR_ABI_VERSION); return 2; }

    gr_ctx* ctx = NULL;
    int32_t rc = create(0, &ctx);
    if (rc == GR_ERR_NO_DEVICE) { printf("no device: %s\n", last_error()); return 3; }
    if (rc != GR_OK) { fprintf(stderr, "gr_ctx_create: %d %s\n", rc, last_error()); return 1; }

    /* KerrMetric(M = 1, a = 0), observer (0, 100, 85 deg, 0), no disc, λ in (0, 200), default chart and tolerances */
    gr_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.metric_id = GR_METRIC_KERR;
    cfg.params[0] = 1.0;
    cfg.r_inner = 2.0 * 1.01;
    cfg.r_outer = 12000.0;
    cfg.gtol = 1e-2;
    cfg.lambda1 = 200.0;
    cfg.abstol = cfg.reltol = 1e-9;
    cfg.maxiters = 1000000;
    cfg.hemi_delta = 1e-4;

    const double r = 100.0, th = 85.0 * 3.14159265358979323846 / 180.0;
    gr_plane pl;
    memset(&pl, 0, sizeof pl);
    pl.x_obs[1] = r; pl.x_obs[2] = th;
    /* Mx = g^-1 * hcat(lnrbasis(g)...) (tracing/utility.jl:32-40) for a diagonal metric:
     * diag(-1/sqrt(-g_tt), 1/sqrt(g_rr), 1/sqrt(g_thth), 1/sqrt(g_phph)); the time row only seeds v^t, which
     * constrain_all replaces on the device */
    const double f = 1.0 - 2.0 / r;
    pl.Mx[0] = -1.0 / sqrt(f);
    pl.Mx[5] = sqrt(f);
    pl.Mx[10] = 1.0 / r;
    pl.Mx[15] = 1.0 / (r * sin(th));
    pl.alpha0 = -9.5; pl.alpha1 = 9.5; pl.beta0 = -9.5; pl.beta1 = 9.5;
    pl.width = 20; pl.height = 20; pl.offset = 1e-6;

    gr_pointfunction pf;
    memset(&pf, 0, sizeof pf);
    pf.pf_id = GR_PF_AFFINE_TIME;
    pf.filter_id = GR_FILTER_EARLY_TERM;     /* ConstPointFunctions.shadow = affine_time ∘ filter_early_term */
    pf.fill = NAN;
    gr_range rg = { 0, 400, 400, 1 };
    double img[400];
    gr_stats st;
    rc = render(ctx, &cfg, &pl, &pf, &rg, img, &st);
    if (rc != GR_OK) { fprintf(stderr, "gr_render: %d %s\n", rc, last_error()); destroy(ctx); return 1; }
    double sum = 0.0;
    for (int i = 0; i < 400; ++i) if (img[i] == img[i]) sum += img[i];
    printf("rays %lld, steps/ray %.1f, fingerprint %.9f (reference 9009.452876610)\n", (long long)st.rays,
           (double)st.accepted_steps / (double)st.rays, sum);
    destroy(ctx);
    return fabs(sum / 9009.452876609641 - 1.0) < 1e-6 ? 0 : 1;
}

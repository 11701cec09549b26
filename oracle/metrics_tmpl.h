/*
 * metrics_tmpl.h -- ORACLE (test infrastructure only; see gradus_oracle.h).
 *
 * metric_components of the reference written once over an abstract number type so the
 * oracle can evaluate it on forward-mode dual numbers exactly as the reference does with
 * ForwardDiff (src/tracing/method-implementations/auto-diff.jl:206-211).
 *
 * Include with these macros defined:
 *   NUM                 number type
 *   FN(name)            name mangler
 *   N_CONST(x)          lift a double
 *   N_ADD N_SUB N_MUL N_DIV (NUM,NUM)->NUM ; N_SCALE(double,NUM) ; N_NEG(NUM)
 *   N_SIN(NUM) N_COS(NUM) N_ATAN(NUM)
 *   N_VAL(NUM)          the value as a double (for the piecewise definitions)
 */

/* __BoyerLindquistAD.metric_components, src/metrics/kerr-metric.jl:11-28 */
static void FN(kerr_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1];
    const double R = 2.0 * M;
    NUM s = N_SIN(th);
    NUM sin2 = N_MUL(s, s);                                /* sinθ2 = sin(θ)^2      :14 */
    NUM cos2 = N_SUB(N_CONST(1.0), sin2);                  /* cosθ2 = 1 - sinθ2     :15 */
    NUM r2 = N_MUL(r, r);
    NUM Sig = N_ADD(r2, N_SCALE(a * a, cos2));             /* Σ₀                    :17 */
    NUM iSig = N_DIV(N_CONST(1.0), Sig);                   /* Σ₀⁻¹                  :18 */
    NUM gam = N_SCALE(a, N_MUL(N_SCALE(R, sin2), r));      /* γ = sinθ2*R*r*a       :19 */
    NUM Rr = N_SCALE(R, r);
    NUM Delta = N_SUB(N_ADD(r2, N_CONST(a * a)), Rr);      /* Δ(r,R,a)              :7  */

    g[0] = N_NEG(N_SUB(N_CONST(1.0), N_MUL(Rr, iSig)));    /* tt                    :21 */
    g[1] = N_DIV(Sig, Delta);                              /* rr                    :22 */
    g[2] = Sig;                                            /* θθ                    :23 */
    g[3] = N_MUL(sin2, N_ADD(N_ADD(r2, N_CONST(a * a)),
                             N_MUL(N_SCALE(a, gam), iSig))); /* ϕϕ                  :24 */
    g[4] = N_NEG(N_MUL(gam, iSig));                        /* tϕ                    :26 */
}

/* __JohannsenAD.metric_components, src/metrics/johannsen-ad.jl:4-34 */
static void FN(johannsen_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], a13 = p[2], a22 = p[3], a52 = p[4], e3 = p[5];
    NUM Mr = N_DIV(N_CONST(M), r);
    NUM Mr2 = N_MUL(Mr, Mr);
    NUM Mr3 = N_MUL(Mr2, Mr);
    NUM A1 = N_ADD(N_CONST(1.0), N_SCALE(a13, Mr3));       /* A₁ :8  */
    NUM A2 = N_ADD(N_CONST(1.0), N_SCALE(a22, Mr2));       /* A₂ :9  */
    NUM A5 = N_ADD(N_CONST(1.0), N_SCALE(a52, Mr2));       /* A₅ :10 */
    NUM c = N_COS(th);
    NUM r2 = N_MUL(r, r);
    NUM f = N_DIV(N_CONST(e3 * M * M * M), r);             /* f  :4  */
    NUM Sig = N_ADD(N_ADD(r2, N_SCALE(a * a, N_MUL(c, c))), f);   /* Σ :5 */
    NUM Delta = N_ADD(N_SUB(r2, N_SCALE(2.0 * M, r)), N_CONST(a * a)); /* Δ :6 */
    NUM r2a2 = N_ADD(r2, N_CONST(a * a));
    NUM s = N_SIN(th);
    NUM s2 = N_MUL(s, s);
    NUM dn = N_SUB(N_MUL(r2a2, A1), N_SCALE(a * a, N_MUL(A2, s2)));
    NUM denom = N_MUL(dn, dn);                             /* :25 */
    NUM tt = N_NEG(N_MUL(Sig, N_SUB(Delta, N_SCALE(a * a, N_MUL(N_MUL(A2, A2), s2)))));   /* :27 */
    NUM rr = N_DIV(Sig, N_MUL(Delta, A5));                 /* :28 */
    NUM pp = N_MUL(N_MUL(Sig, s2),
                   N_SUB(N_MUL(N_MUL(r2a2, r2a2), N_MUL(A1, A1)),
                         N_SCALE(a * a, N_MUL(Delta, s2))));                               /* :30 */
    NUM tp = N_NEG(N_SCALE(a, N_MUL(N_MUL(Sig, s2),
                                    N_SUB(N_MUL(N_MUL(r2a2, A1), A2), Delta))));           /* :32 */
    g[0] = N_DIV(tt, denom);
    g[1] = rr;
    g[2] = Sig;
    g[3] = N_DIV(pp, denom);
    g[4] = N_DIV(tp, denom);
}

/* __MorrisThorneAD.metric_components, src/metrics/morris-thorne-ad.jl:4-15 (ϕϕ has sinθ to the
 * first power in the reference; reproduced as is) */
static void FN(morris_thorne_components)(const double* p, NUM l, NUM th, NUM g[5])
{
    const double b = p[0];
    NUM w = N_ADD(N_CONST(b * b), N_MUL(l, l));
    g[0] = N_CONST(-1.0);
    g[1] = N_CONST(1.0);
    g[2] = w;
    g[3] = N_MUL(w, N_SIN(th));
    g[4] = N_CONST(0.0);
}

/* __BumblebeeAD.metric_components, src/metrics/bumblebee-ad.jl:6-21 */
static void FN(bumblebee_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], l = p[2];
    NUM s = N_SIN(th);
    NUM s2 = N_MUL(s, s);
    NUM r2 = N_MUL(r, r);
    NUM Delta = N_SCALE(1.0 / (l + 1.0), N_SUB(r2, N_SCALE(2.0 * M, r)));     /* Δ :6 */
    g[0] = N_NEG(N_SUB(N_CONST(1.0), N_DIV(N_CONST(2.0 * M), r)));
    g[1] = N_DIV(r2, Delta);
    g[2] = r2;
    g[3] = N_MUL(r2, s2);
    g[4] = N_DIV(N_SCALE(-2.0 * M * a, s2), r);
}

/* __KerrNewmanAD.metric_components, src/metrics/kerr-newman-ad.jl:6-27 */
static void FN(kerr_newman_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], Q = p[2];
    const double R = 2.0 * M;
    NUM c = N_COS(th);
    NUM ac = N_SCALE(a, c);
    NUM r2 = N_MUL(r, r);
    NUM Sig = N_ADD(r2, N_MUL(ac, ac));
    NUM s = N_SIN(th);
    NUM s2 = N_MUL(s, s);
    NUM Delta = N_ADD(N_SUB(r2, N_SCALE(R, r)), N_CONST(a * a + Q * Q));
    NUM r2a2 = N_ADD(r2, N_CONST(a * a));
    g[0] = N_DIV(N_SUB(N_SCALE(a * a, s2), Delta), Sig);
    g[1] = N_DIV(Sig, Delta);
    g[2] = Sig;
    g[3] = N_MUL(N_DIV(s2, Sig), N_SUB(N_MUL(r2a2, r2a2), N_SCALE(a * a, N_MUL(s2, Delta))));
    g[4] = N_MUL(N_DIV(N_SCALE(a, s2), Sig), N_SUB(Delta, r2a2));
}

/* __JohannsenPsaltisAD.metric_components, src/metrics/johannsen-psaltis-ad.jl:4-27 */
static void FN(johannsen_psaltis_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], e3 = p[2];
    NUM c = N_COS(th);
    NUM r2 = N_MUL(r, r);
    NUM Sig = N_ADD(r2, N_SCALE(a * a, N_MUL(c, c)));
    NUM h = N_DIV(N_SCALE(e3 * M * M * M, r), N_MUL(Sig, Sig));                 /* h :4 */
    NUM s = N_SIN(th);
    NUM s2 = N_MUL(s, s);
    NUM Delta = N_ADD(N_SUB(r2, N_SCALE(2.0 * M, r)), N_CONST(a * a));
    NUM hp1 = N_ADD(N_CONST(1.0), h);
    NUM tMr = N_SCALE(2.0 * M, r);
    g[0] = N_NEG(N_MUL(hp1, N_SUB(N_CONST(1.0), N_DIV(tMr, Sig))));
    g[1] = N_DIV(N_MUL(Sig, hp1), N_ADD(Delta, N_SCALE(a * a, N_MUL(s2, h))));
    g[2] = Sig;
    NUM term1 = N_MUL(s2, N_ADD(N_ADD(r2, N_CONST(a * a)), N_DIV(N_SCALE(a * a, N_MUL(tMr, s2)), Sig)));
    NUM term2 = N_DIV(N_MUL(N_SCALE(a * a, h), N_MUL(N_ADD(Sig, tMr), N_MUL(s2, s2))), Sig);
    g[3] = N_ADD(term1, term2);
    g[4] = N_NEG(N_DIV(N_MUL(N_SCALE(a, tMr), N_MUL(s2, hp1)), Sig));
}

/* __DilatonAxionAD.metric_components, src/metrics/dilaton-axion-ad.jl:8-46 ; p = M, a, β, b */
static void FN(dilaton_axion_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], be = p[2], b = p[3];
    const double R = M;
    const double bb = (be == 0.0) ? 0.0 : be / b;             /* βb  :24 */
    const double ba = (be == 0.0) ? 0.0 : be / a;             /* βa  :25 */
    const double bab = (be == 0.0) ? 0.0 : be / (a * b);      /* βab :26 */
    NUM c = N_COS(th), s = N_SIN(th);
    NUM s2 = N_MUL(s, s);
    NUM r2 = N_MUL(r, r);
    NUM Sig = N_ADD(r2, N_SCALE(a * a, N_MUL(c, c)));                                   /* Σ(r,a,θ)   :9  */
    NUM Del = N_SUB(N_ADD(r2, N_CONST(a * a)), N_SCALE(2.0 * R, r));                    /* Δ(r,2R,a)  :10 */
    NUM bt = N_ADD(N_CONST(be * be), N_SCALE(2.0 * b, r));                              /* β² + 2br        */
    NUM Delh = N_SUB(N_SUB(Del, bt), N_CONST(R * (R + 2.0 * b) * bb * bb));             /* Δhat       :13 */
    NUM Sigh = N_ADD(N_SUB(Sig, bt), N_SCALE(R * R * bb, N_SUB(N_CONST(bb), N_SCALE(2.0 * a, c))));  /* Σhat :12 */
    NUM del = N_ADD(N_SUB(r2, N_SCALE(2.0 * b, r)), N_CONST(a * a));                    /* δ          :15 */
    NUM W = N_ADD(N_CONST(1.0),
                  N_DIV(N_ADD(N_SCALE(bab, N_SUB(N_SCALE(2.0, c), N_CONST(bab))), N_CONST(ba * ba)), s2));  /* W :17 */
    NUM Was = N_MUL(W, N_SCALE(a, s));
    NUM A = N_SUB(N_MUL(del, del), N_MUL(Delh, N_MUL(Was, Was)));                       /* A          :18 */
    g[0] = N_NEG(N_DIV(N_SUB(Delh, N_SCALE(a * a, s2)), Sigh));
    g[1] = N_DIV(Sigh, Delh);
    g[2] = Sigh;
    g[3] = N_DIV(N_MUL(A, s2), Sigh);
    g[4] = N_NEG(N_DIV(N_MUL(N_SCALE(a, N_SUB(del, N_MUL(Delh, W))), s2), Sigh));
}

/* SphericalMetric (Minkowski in spherical coordinates), src/metrics/minkowski.jl:4-13 */
static void FN(spherical_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    (void)p;
    NUM s = N_SIN(th);
    NUM r2 = N_MUL(r, r);
    g[0] = N_CONST(-1.0);
    g[1] = N_CONST(1.0);
    g[2] = r2;
    g[3] = N_MUL(r2, N_MUL(s, s));
    g[4] = N_CONST(0.0);
}

/* __KerrDarkMatter.metric_components, src/metrics/kerr-dark-matter.jl:6-49 ; p = M_bh, a, M_dm, Δr, rₛ */
static void FN(kerr_dark_matter_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double Mbh = p[0], a = p[1], Mdm = p[2], dR = p[3], rs = p[4];
    NUM M = N_CONST(Mbh);
    if (N_VAL(r) < rs) {                                   /* dark_matter_mass :15-23 */
    } else if (N_VAL(r) < rs + dR) {
        NUM dr = N_DIV(N_SUB(r, N_CONST(rs)), N_CONST(dR));                 /* G :10-13 */
        NUM G = N_MUL(N_SUB(N_CONST(3.0), N_SCALE(2.0, dr)), N_MUL(dr, dr));
        M = N_ADD(M, N_SCALE(Mdm, G));
    } else {
        M = N_ADD(M, N_CONST(Mdm));
    }
    NUM R = N_SCALE(2.0, M);
    NUM s = N_SIN(th);
    NUM sin2 = N_MUL(s, s);
    NUM cos2 = N_SUB(N_CONST(1.0), sin2);
    NUM r2 = N_MUL(r, r);
    NUM Sig = N_ADD(r2, N_SCALE(a * a, cos2));
    NUM Rr = N_MUL(R, r);
    NUM Delta = N_SUB(N_ADD(r2, N_CONST(a * a)), Rr);
    g[0] = N_NEG(N_SUB(N_CONST(1.0), N_DIV(Rr, Sig)));
    g[1] = N_DIV(Sig, Delta);
    g[2] = Sig;
    g[3] = N_MUL(sin2, N_ADD(N_ADD(r2, N_CONST(a * a)), N_DIV(N_SCALE(a * a, N_MUL(sin2, Rr)), Sig)));
    g[4] = N_DIV(N_NEG(N_SCALE(a, N_MUL(Rr, sin2))), Sig);
}

/* _smooth_interpolate(x, x₀; δx = 2.5, smoothing_offset = 1e4), src/utils.jl:158-168 */
static NUM FN(smooth_interpolate)(NUM x, double x0)
{
    const double dx = 2.5, off = 1e4;
    if (N_VAL(x) <= x0 - dx / 2) return N_CONST(1.0);
    if (N_VAL(x) <= x0 + dx / 2) {
        NUM t = N_DIV(N_SUB(x, N_CONST(x0)), N_CONST(dx));
        NUM v = N_ADD(N_SCALE(1.0 / M_PI, N_ATAN(N_SCALE(off, t))), N_CONST(0.5));
        return N_SUB(N_CONST(1.0), v);
    }
    return N_CONST(0.0);
}

/* __KerrRefractiveAD.metric_components, src/metrics/kerr-refractive-ad.jl:8-33 ; p = M, a, n, corona_radius */
static void FN(kerr_refractive_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], n0 = p[2], rc = p[3];
    const double R = 2.0 * M;
    NUM c = N_COS(th);
    NUM r2 = N_MUL(r, r);
    NUM Sig = N_ADD(r2, N_SCALE(a * a, N_MUL(c, c)));
    NUM s = N_SIN(th);
    NUM sin2 = N_MUL(s, s);
    NUM Rr = N_SCALE(R, r);
    NUM Delta = N_ADD(N_SUB(r2, Rr), N_CONST(a * a));
    NUM tt = N_NEG(N_SUB(N_CONST(1.0), N_DIV(Rr, Sig)));
    NUM pp = N_MUL(sin2, N_ADD(N_ADD(r2, N_CONST(a * a)), N_DIV(N_SCALE(a * a, N_MUL(sin2, Rr)), Sig)));
    NUM tp = N_DIV(N_NEG(N_SCALE(a, N_MUL(Rr, sin2))), Sig);
    NUM t = FN(smooth_interpolate)(r, rc);
    NUM n = N_ADD(t, N_SCALE(n0, N_SUB(N_CONST(1.0), t)));               /* n = t + (1 - t) n */
    g[0] = N_DIV(tt, N_MUL(n, n));
    g[1] = N_DIV(Sig, Delta);
    g[2] = Sig;
    g[3] = pp;
    g[4] = N_DIV(tp, n);
}

/* __NoZMetric.metric_components, src/metrics/noz-metric.jl:7-47 ; p = M, a, ϵ */
static void FN(noz_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    const double M = p[0], a = p[1], e = p[2];
    NUM s = N_SIN(th);
    NUM sin2 = N_MUL(s, s);
    NUM y = N_COS(th);
    NUM y2 = N_MUL(y, y);
    NUM eps = N_SCALE(e * M * a, y);                                       /* epsilon :7 */
    NUM r2 = N_MUL(r, r);
    NUM a2y2 = N_SCALE(a * a, y2);
    NUM S = N_ADD(r2, a2y2);                                               /* r² + a²y² */
    NUM tMr = N_SCALE(2.0 * M, r);
    NUM D = N_ADD(N_MUL(S, S), N_MUL(N_ADD(N_SUB(r2, tMr), a2y2), eps));   /* common denominator */
    NUM Se = N_ADD(S, eps);
    NUM omy2 = N_SUB(N_CONST(1.0), y2);
    NUM big = N_ADD(N_ADD(N_ADD(N_ADD(N_MUL(r2, r2), N_SCALE(a * a * a * a, y2)),
                                N_MUL(r2, N_ADD(N_ADD(N_CONST(a * a), a2y2), eps))),
                          N_SCALE(a * a, eps)),
                    N_MUL(tMr, N_SUB(N_SUB(N_CONST(a * a), a2y2), eps)));
    NUM yy = N_DIV(Se, omy2);
    g[0] = N_ADD(N_CONST(-1.0), N_DIV(N_MUL(tMr, S), D));
    g[1] = N_DIV(Se, N_ADD(N_SUB(r2, tMr), N_CONST(a * a)));
    g[2] = N_MUL(yy, sin2);
    g[3] = N_DIV(N_MUL(N_MUL(omy2, Se), big), D);
    g[4] = N_NEG(N_DIV(N_MUL(N_SCALE(a, tMr), N_MUL(omy2, Se)), D));
}

#ifdef ORC_WITH_TEST_METRIC
/* (compiled into libgradus_oracle_usermetric.so only -- oracle/Makefile `usermetric` -- so that the oracle every other test
 * uses stays bit for bit the build its thresholds were measured with: gcc contracts FMAs differently once a switch gains a case)
 * NOT a metric of the reference: a stand-in for a USER-DEFINED AbstractStaticAxisSymmetric metric (the plugin contract of
 * src/Gradus.jl:78-86 -- "define metric_components(m, rθ)" -- exercised by tests of GR_METRIC_TABULATED).  Kerr with
 * g_tt scaled by 1 + ϵ sin²θ / (1 + ((r - r_b) / w)²): smooth, asymptotically Kerr, in no catalogue.  The oracle pushes it
 * through the same dual numbers as every other metric, i.e. what the reference would do with it.  p = M, a, ϵ, r_b, w */
static void FN(test_bump_components)(const double* p, NUM r, NUM th, NUM g[5])
{
    FN(kerr_components)(p, r, th, g);
    const double e = p[2], rb = p[3], w = p[4];
    NUM s = N_SIN(th);
    NUM x = N_SCALE(1.0 / w, N_SUB(r, N_CONST(rb)));
    NUM bump = N_DIV(N_SCALE(e, N_MUL(s, s)), N_ADD(N_CONST(1.0), N_MUL(x, x)));
    g[0] = N_MUL(g[0], N_ADD(N_CONST(1.0), bump));
}
#endif

"""Disc-rim classification against the oracle for several builds of libgradus_mi355x.so (VERDICT r1 item 7:
does the single-precision PI controller cost parity?).

    python scripts/controller_ab.py [--size 1024] [--config C2|C4] name=path.so [name=path.so ...]

Every build renders BASELINE config C2 (or C4) at full size in its own child process (GRADUS_MI355X_LIB selects the
library); the parent never touches the GPU, runs the oracle once and reports per build: pixels whose class differs
from the oracle's, worst / median relative error of the redshift on common hits, kernel ms (median of 7), and
the flips BETWEEN the builds.  Output: one JSON line (also written to gpurun_out/controller_ab.json).
"""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)
JOH = (1.0, 0.7, 2.0, 0.0, 0.0, 1.0)


def child(config, size, out):
    import gradus_jl_amd as G

    ens = G.EnsembleMI355X(0)
    if config == "C2":
        m = G.KerrMetric(1.0, 0.998)
        x = np.array([0.0, 1000.0, math.radians(75), 0.0])
        pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
    else:
        m = G.JohannsenMetric(*JOH)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.redshift(m, x, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
    ms = []
    for _ in range(8):
        _, _, img, st = G.rendergeodesics(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=size, image_height=size,
                                          alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf, ensemble=ens, stats=True)
        ms.append(st["kernel_ms"])
    extra = {}
    if config == "C4":
        extra = {f"plunge{i}": a for i, a in enumerate(pf.extra["plunge"])}
    np.savez(out, img=img, ms=np.array(ms[1:]), steps=st["accepted_steps"] / st["rays"], rej=st["rejected_steps"] / st["rays"], **extra)


def oracle_child(size, out):
    """The C2 reference image from whichever oracle build GRADUS_ORACLE_LIB selects."""
    from oracle import oracle as O

    isco = 1.2369706551751847
    cfg = O.make_config("kerr", (1.0, 0.998), disc=(isco, 50.0), lambda_max=2000.0)
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    np.save(out, O.rendergeodesics(cfg, x, ALIMS, BLIMS, size, size, pf_id=O.PF_REDSHIFT, filter_id=O.FILTER_INTERSECTED,
                                   r_isco=isco))


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        return child(args[1], int(args[2]), args[3])
    if args and args[0] == "--oracle-child":
        return oracle_child(int(args[1]), args[2])
    size, config, libs = 1024, "C2", []
    it = iter(args)
    for a in it:
        if a == "--size":
            size = int(next(it))
        elif a == "--config":
            config = next(it)
        else:
            name, path = a.split("=")
            libs.append((name, os.path.abspath(path)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    imgs = {}
    res = {"config": config, "size": size, "builds": {}}
    for name, path in libs:
        out = os.path.join(ROOT, "gpurun_out", f"ctlab_{name}.npz")
        env = dict(os.environ, GRADUS_MI355X_LIB=path)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", config, str(size), out], env=env)
        z = np.load(out)
        imgs[name] = z
        res["builds"][name] = {"kernel_ms_median": float(np.median(z["ms"])), "kernel_ms_min": float(z["ms"].min()),
                               "steps_per_ray": float(z["steps"]), "rejected_per_ray": float(z["rej"])}
    from oracle import oracle as O

    if config == "C2":
        isco = 1.2369706551751847
        cfg = O.make_config("kerr", (1.0, 0.998), disc=(isco, 50.0), lambda_max=2000.0)
        x = np.array([0.0, 1000.0, math.radians(75), 0.0])
        ref = O.rendergeodesics(cfg, x, ALIMS, BLIMS, size, size, pf_id=O.PF_REDSHIFT, filter_id=O.FILTER_INTERSECTED, r_isco=isco)
        refs = {name: ref for name, _ in libs}
    else:
        cfg0 = O.make_config("johannsen", JOH)
        isco = O.isco(cfg0)
        cfg = O.make_config("johannsen", JOH, disc=(isco, 50.0), lambda_max=2000.0)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        refs = {}
        pts = None
        for name, _ in libs:
            z = imgs[name]
            plunge = tuple(z[f"plunge{i}"] for i in range(4))
            if pts is None:
                pts = O.trace(cfg, x, O.render_velocities(cfg, x, ALIMS, BLIMS, size, size))
            refs[name] = O.apply_pf(cfg, pts, 2000.0, pf_id=O.PF_REDSHIFT, filter_id=O.FILTER_INTERSECTED, r_isco=isco,
                                    plunge=plunge).reshape(size, size).T
    for name, _ in libs:
        img, ref = imgs[name]["img"], refs[name]
        flips = np.isnan(img) != np.isnan(ref)
        both = ~np.isnan(img) & ~np.isnan(ref)
        rel = np.abs(img[both] / ref[both] - 1)
        b = res["builds"][name]
        b.update({"status_flips_vs_oracle": int(flips.sum()), "device_only_hits": int((flips & np.isnan(ref)).sum()),
                  "oracle_only_hits": int((flips & np.isnan(img)).sum()), "max_rel_err": float(rel.max()),
                  "median_rel_err": float(np.median(rel)), "p99_rel_err": float(np.percentile(rel, 99)),
                  "flips_per_ms": float(flips.sum() / b["kernel_ms_median"])})
    nofma = os.path.join(ROOT, "oracle", "libgradus_oracle_nofma.so")
    if config == "C2" and os.path.exists(nofma):
        # the floor: the SAME algorithm (the oracle) rounded differently (FMA contraction off)
        out = os.path.join(ROOT, "gpurun_out", "ctlab_oracle_nofma.npy")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--oracle-child", str(size), out],
                              env=dict(os.environ, GRADUS_ORACLE_LIB=nofma))
        alt = np.load(out)
        flips = np.isnan(alt) != np.isnan(ref)
        both = ~np.isnan(alt) & ~np.isnan(ref)
        rel = np.abs(alt[both] / ref[both] - 1)
        res["oracle_nofma_vs_oracle"] = {"status_flips": int(flips.sum()), "max_rel_err": float(rel.max()),
                                         "median_rel_err": float(np.median(rel)), "p99_rel_err": float(np.percentile(rel, 99))}
        for name, _ in libs:
            res["builds"][name]["status_flips_vs_oracle_nofma"] = int((np.isnan(imgs[name]["img"]) != np.isnan(alt)).sum())
    names = [n for n, _ in libs]
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a, b = imgs[names[i]]["img"], imgs[names[j]]["img"]
            res[f"flips_{names[i]}_vs_{names[j]}"] = int((np.isnan(a) != np.isnan(b)).sum())
    print(json.dumps(res))
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"controller_ab_{config}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

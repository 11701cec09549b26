#!/usr/bin/env python3
"""Static instruction mix of a gfx950 kernel's loop: splits the .s file into basic blocks, finds the innermost
loop with the most FP64 work (the Tsit5 step), and prints per-block and whole-loop counts.  The blocks that the
step executes unconditionally are those without a skipping branch around them; blocks guarded by
s_cbranch_execz (sincos fallbacks, event sampling) are listed separately.

    python scripts/asm_blocks.py file.s [min block size to list]
"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 30
blocks, cur = [], None
for l in lines:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
        continue
    if cur is None:
        cur = ["entry", []]
        blocks.append(cur)
    t = l.strip()
    if re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)", t):
        cur[1].append(t)
idx = {b[0]: i for i, b in enumerate(blocks)}


def stats(ins):
    c = collections.Counter(x.split()[0] for x in ins)
    return {
        "n": len(ins), "valu": sum(v for k, v in c.items() if k.startswith("v_")),
        "f64": sum(v for k, v in c.items() if "f64" in k), "fma": c["v_fma_f64"] + c["v_fmac_f64_e32"] + c["v_fmac_f64_e64"],
        "mul": c["v_mul_f64"], "add": c["v_add_f64"], "rcp": c["v_rcp_f64_e32"], "mov": c["v_mov_b32_e32"] + c["v_mov_b64_e32"],
        "cnd": c["v_cndmask_b32_e32"] + c["v_cndmask_b32_e64"], "salu": sum(v for k, v in c.items() if k.startswith("s_")),
        "mem": sum(v for k, v in c.items() if k.split("_")[0] in ("global", "flat", "scratch", "buffer", "ds")),
    }


# back edges -> loops
loops = []
for i, (name, ins) in enumerate(blocks):
    for x in ins:
        if x.startswith("s_cbranch") or x.startswith("s_branch"):
            tgt = x.split()[-1]
            if tgt in idx and idx[tgt] <= i:
                loops.append((idx[tgt], i))
best = None
for a, b in loops:
    f = sum(stats(blocks[k][1])["f64"] for k in range(a, b + 1))
    if best is None or f > best[0]:
        best = (f, a, b)
if best is None:
    print("no loop found")
    sys.exit(0)
_, a, b = best
print(f"step loop: blocks {blocks[a][0]} .. {blocks[b][0]}")
tot = collections.Counter()
for k in range(a, b + 1):
    name, ins = blocks[k]
    s = stats(ins)
    for kk, v in s.items():
        tot[kk] += v
    if s["n"] >= lo:
        print(f"  {name:12s} n {s['n']:4d} valu {s['valu']:4d} f64 {s['f64']:4d} fma {s['fma']:4d} mul {s['mul']:4d} add {s['add']:3d} "
              f"rcp {s['rcp']:2d} mov {s['mov']:3d} cnd {s['cnd']:3d} salu {s['salu']:3d} mem {s['mem']:2d}")
print("  whole loop (static, incl. rarely executed blocks):", dict(tot))

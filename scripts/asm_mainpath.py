#!/usr/bin/env python3
"""Static VALU count of the Tsit5 step's COMMON path in a kernel_probe.sh assembly file: all blocks of the step loop
except those that hold a full sincos evaluation (v_rndne_f64: the rotation fall-backs and the event-sampling block, taken
in 1-2 % of the wave-steps).  A relative yardstick for instruction-trimming work (the dynamic number is rocprofv3's
SQ_INSTS_VALU / waves / steps).

    python scripts/asm_mainpath.py /tmp/probe/probe.s
"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
blocks, cur = [], None
for l in lines:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
        continue
    if cur is None:
        continue
    t = l.strip()
    if re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)", t):
        cur[1].append(t)
idx = {b[0]: i for i, b in enumerate(blocks)}
loops = []
for i, (name, ins) in enumerate(blocks):
    for x in ins:
        if x.startswith("s_cbranch") or x.startswith("s_branch"):
            tgt = x.split()[-1]
            if tgt in idx and idx[tgt] <= i:
                loops.append((idx[tgt], i))
best = max(loops, key=lambda ab: sum(sum("f64" in x.split()[0] for x in blocks[k][1]) for k in range(ab[0], ab[1] + 1)))
a, b = best
tot = collections.Counter()
kinds = collections.Counter()
for k in range(a, b + 1):
    name, ins = blocks[k]
    if any(x.startswith("v_rndne_f64") for x in ins):
        # keep the instructions in front of the first branch of such a block (they run before the guard)
        cut = next((i for i, x in enumerate(ins) if x.startswith("s_cbranch")), 0)
        ins = ins[:cut] if len(ins) > 200 else []
    for x in ins:
        op = x.split()[0]
        if op.startswith("v_"):
            tot["valu"] += 1
            if "f64" in op and not op.startswith("v_cmp") and not op.startswith("v_cvt"):
                tot["f64"] += 1
            else:
                kinds[re.sub(r"_e(32|64)$", "", op)] += 1
        elif op.startswith("s_"):
            tot["salu"] += 1
        else:
            tot["mem"] += 1
print(f"common path (static): valu {tot['valu']}  fp64-arith {tot['f64']}  other-valu {tot['valu'] - tot['f64']}  salu {tot['salu']}  mem {tot['mem']}")
print("  other VALU:", ", ".join(f"{k} {v}" for k, v in kinds.most_common(14)))

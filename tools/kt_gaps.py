#!/usr/bin/env python3
"""Per-kernel time and inter-kernel gaps of one timed solve from a rocprofv3 --kernel-trace CSV.
    python tools/kt_gaps.py <run_kernel_trace.csv> [solve-index-from-end] [--timeline]
Prints: per kernel name launches / total / avg, the GPU-busy sum, the gap sum and the span of the selected solve
(solves are delimited by random_fill/memcpy-free heuristics: the `synth_apply_kernel` that follows a long idle)."""
import csv, sys, re
from collections import defaultdict
rows = list(csv.DictReader(open([a for a in sys.argv[1:] if a != "--timeline"][0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\([^()]*\)$", "", nm)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
# split into solves at gaps > 200 us that precede a matvec chain (host work between solves: copy of the guess, python)
solves, cur = [], []
for e in ev:
    if "fillBuffer" in e[2] and e[1] - e[0] < 4_000:     # bench.py's 8-byte marker in front of every solve
        if cur:
            solves.append(cur)
        cur = []
        continue
    cur.append(e)
solves.append(cur)
solves = [s for s in solves if len(s) > 100]
timeline = "--timeline" in sys.argv
args = [a for a in sys.argv[1:] if a != "--timeline"]
k = int(args[1]) if len(args) > 1 else 1
s = solves[-k]
tot = defaultdict(lambda: [0, 0])
for st, en, nm in s:
    tot[nm][0] += 1; tot[nm][1] += en - st
busy = sum(v[1] for v in tot.values())
span = s[-1][1] - s[0][0]
gaps = sorted(((b[0] - a[1]), a[2], b[2]) for a, b in zip(s, s[1:]))
print(f"solves found {len(solves)}; selected has {len(s)} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, gaps {(span-busy)/1e6:.3f} ms")
for nm, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/1e3:10.1f} us {c:5d} x {t/c/1e3:8.2f} us  {nm}")
print("largest gaps (us):")
for g, a, b in gaps[-12:]:
    print(f"  {g/1e3:8.1f}  after {a[:40]:40s} before {b[:40]}")
import statistics
print("median gap", statistics.median(g for g, _, _ in gaps) / 1e3, "us; gaps > 5us:", sum(1 for g, _, _ in gaps if g > 5000))
if timeline:
    # every launch of the selected solve: start offset, duration, idle time before it
    print("timeline (us): start  dur  gap-before  kernel")
    t0, prev = s[0][0], s[0][0]
    for st, en, nm in s:
        print(f"  {(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {(st - prev) / 1e3:8.1f}  {nm[:70]}")
        prev = en

#!/usr/bin/env python3
"""Stall counters per kernel symbol from rocprofv3 PMC passes (derived metrics of the installed counter set):
    python tools/pmc_stalls.py <dir> [<dir> ...]        # every <dir> holds one pass: run_counter_collection.csv (+ run_kernel_trace.csv)
Per kernel: launches, average duration (from the kernel trace of the first pass that has one), and the dispatch-weighted mean of
every counter collected for it.  Counters are collected in separate passes (tools/profile_stalls.sh): a kernel's values of different
passes belong to different runs of the same workload."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    for f in cc:
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if kt and not dur:
        for r in csv.DictReader(open(kt[0])):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for k in per for c in per[k]})
order = sorted(per, key=lambda k: -sum(dur.get(k, [0.0])))
print("kernel | launches | avg us | " + " | ".join(names))
for k in order:
    if sum(dur.get(k, [0.0])) < 200.0:       # (kernels below 0.2 ms in total are not listed)
        continue
    n = len(dur.get(k, [])) or max(len(v) for v in per[k].values())
    avg = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    vals = []
    for c in names:
        v = per[k].get(c)
        vals.append(f"{sum(v) / len(v):.3g}" if v else "-")
    print(f"{k} | {n} | {avg:.1f} | " + " | ".join(vals))

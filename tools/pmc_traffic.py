#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md section HBM prescribes) into per-launch HBM traffic per kernel class.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Units / gfx950 corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read on gfx950, so it is doubled; WRITE_SIZE
is exact for 16-byte-per-lane streaming stores.
"""
import csv
import json
import re
import sys
from collections import defaultdict

CLASS = [("gram_kernel", "gram"), ("gram_lds_kernel", "gram"), ("gram_reduce", "gram"), ("ritz_kernel", "ritz"), ("ritz_reduce", "ritz")]


def klass(name, grid):
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+),", name)
    if m:
        return "trmm" if m.group(3) == "2" else "gemm"       # MODE 2 = in-place triangular update
    for key, c in CLASS:
        if key in name:
            return c
    return None


def load(path, counter):
    """Per class: [sum of counter, launches].  The benchmark operator's own small Gram (W^T x, followed in
    dispatch order by synth_apply_kernel) belongs to the matvec class, as in the engine's statistics."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    cls = [klass(r["Kernel_Name"], None) for r in rows]
    for i, r in enumerate(rows):
        if "synth_apply_kernel" in r["Kernel_Name"]:
            j, seen = i - 1, 0
            while j >= 0 and seen < 2:
                if "gram_kernel" in rows[j]["Kernel_Name"] or "gram_lds_kernel" in rows[j]["Kernel_Name"] or "gram_reduce" in rows[j]["Kernel_Name"]:
                    cls[j] = "matvec"; seen += 1
                j -= 1
    per = defaultdict(lambda: [0.0, 0])
    for r, c in zip(rows, cls):
        if c is None:
            continue
        per[c][0] += float(r["Counter_Value"])
        if "reduce" not in r["Kernel_Name"]:
            per[c][1] += 1
        if c != "matvec":
            # per kernel symbol as well (name as bench.py reports it)
            m = re.search(r"((gram_lds|gram|gemm|ritz)_kernel<[^>]*>)", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
            if m:
                per[m.group(1)][0] += float(r["Counter_Value"])
                per[m.group(1)][1] += 1
    return per


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for c in sorted(set(fetch) | set(write)):
        launches = max(fetch[c][1], write[c][1], 1)
        rd = 2.0 * fetch[c][0] * 1024.0
        wr = write[c][0] * 1024.0
        out[c] = round((rd + wr) / launches, 1)
        print(f"{c:6s} launches {launches:4d}  read {rd / launches / 1e6:9.1f} MB  write {wr / launches / 1e6:8.1f} MB  "
              f"total/launch {out[c] / 1e6:9.1f} MB")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

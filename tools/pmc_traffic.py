#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md section HBM
prescribes) into HBM bytes per launch per KERNEL SYMBOL for ONE workload, merged into a table keyed by the workload.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <table.json> "<workload key>"

The workload key is the one bench.py builds (`<solver> n=... roots=... n_max=... max_dav=... guess=...`); bench.py
quotes `roofline.traffic` only from the entry of its own workload and kernel, never from another configuration.

Units / gfx950 corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read on gfx950, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming
stores.  Dispatches of a device-driven chain that found it was not their turn (predicated launches, a few hundred bytes)
are left out of the averages: they are the ones below 5 % of the kernel's largest dispatch.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def kname(full):
    m = re.search(r"((gram_lds|gram|gemm|ritz)_kernel<[^>]*>)", full.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else None


def load(path, counter):
    per = defaultdict(dict)     # kernel -> dispatch id -> value
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = kname(r["Kernel_Name"])
        if k:
            d = per[k]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return per


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    table_path, key = sys.argv[3], sys.argv[4]
    out = {}
    for k in sorted(set(fetch) | set(write)):
        def mean_real(d):
            if not d:
                return 0.0, 0
            top = max(d.values())
            real = [v for v in d.values() if v >= 0.05 * top] if top > 0 else list(d.values())
            return sum(real) / len(real), len(real)
        f, nf = mean_real(fetch.get(k, {}))
        w, nw = mean_real(write.get(k, {}))
        out[k] = round(2.0 * f * 1024.0 + w * 1024.0, 1)
        print(f"{k:52s} launches {max(nf, nw):4d}  read {2 * f * 1024 / 1e6:9.1f} MB  write {w * 1024 / 1e6:8.1f} MB  "
              f"total/launch {out[k] / 1e6:9.1f} MB")
    table = json.load(open(table_path)) if os.path.exists(table_path) else {}
    table = {k: v for k, v in table.items() if isinstance(v, dict)}      # drop the flat round-1 layout
    table[key] = out
    json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

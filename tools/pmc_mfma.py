#!/usr/bin/env python3
"""MFMA utilisation per kernel symbol from one rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
              --kernel-trace --output-format csv -d <dir> -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_mfma.py <dir>/run_counter_collection.csv <dir>/run_kernel_trace.csv [out.json]

Reading of the counters on gfx950 (checked against the instruction counts of the kernels):
  SQ_INSTS_VALU_MFMA_MOPS_F64  = 4 per v_mfma_f64_16x16x4 (512 flop units)
  SQ_VALU_MFMA_BUSY_CYCLES     = 64 per v_mfma_f64_16x16x4, summed over all SIMDs
                                 (64 cycles x 1024 SIMDs x 2.4 GHz x 2048 flop = the 78.6 TFLOP/s FP64 matrix peak)
  GRBM_GUI_ACTIVE              = busy cycles summed over the 8 XCDs (128 SIMDs each)
  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128)
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(sys.argv[2]))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            per[k]["launches"] += 1
            t = trace.get(r["Dispatch_Id"])
            if t:
                per[k]["ns"] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    out = {}
    for k, c in sorted(per.items(), key=lambda kv: -kv[1]["ns"]):
        if c["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
            continue
        n = max(1.0, c["launches"])
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)
        flops = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
        out[k] = {"launches": int(n), "avg_us": round(c["ns"] / n / 1e3, 1), "mfma_util": round(util, 4),
                  "mfma_TFLOPs": round(flops / max(c["ns"], 1.0) / 1e3, 2),
                  "mfma_per_launch": round(c["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0 / n)}
        print(f"{k:52s} x{int(n):4d} {out[k]['avg_us']:8.1f} us  MfmaUtil {100 * util:5.1f} %  {out[k]['mfma_TFLOPs']:6.2f} TFLOP/s issued")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condensed host-side timeline of ONE drop-in (host-callback) Davidson iteration from a rocprofv3 --hip-trace run of
tools/host_mode_timeline.py:  python3 tools/host_mode_timeline_report.py <dir with *_hip_api_trace.csv> [iteration]
Consecutive polls (hipEventQuery) are merged into one entry; time between two API calls that the library spends in the
CALLER's routine (or in its own host logic) shows up as a gap.  Copy and kernel durations come from the *_memory_copy_trace /
*_kernel_trace files of the same run where the profiler delivered them."""
import csv
import glob
import os
import sys

d = sys.argv[1]
want_it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
api = glob.glob(os.path.join(d, "**", "*_hip_api_trace.csv"), recursive=True)[0]
rows = []
with open(api) as f:
    for r in csv.DictReader(f):
        if r["Domain"] != "HIP_RUNTIME_API_EXT" and r["Domain"] != "HIP_RUNTIME_API":
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
rows.sort()
# the solves: separated by the big host-to-device upload of the guess (hipMemcpy of n x 13 doubles happens once per solve);
# simpler: take the LAST 40 % of the trace's D2H staging pattern -- locate iterations by the 4-copy pattern below
ev = []
for s, e, fn in rows:
    if ev and fn == "hipEventQuery" and ev[-1][2] == "hipEventQuery (polling)" and s - ev[-1][1] < 400000:
        ev[-1] = (ev[-1][0], e, ev[-1][2])
        continue
    ev.append((s, e, "hipEventQuery (polling)" if fn == "hipEventQuery" else fn))
# iterations of the second solve: every host-mode callback is hipMemcpyAsync(D2H) .. hipEventSynchronize .. [caller] .. hipMemcpyAsync(H2D)
syncs = [i for i, x in enumerate(ev) if x[2] == "hipEventSynchronize" and x[1] - x[0] > 1_000_000]
# two callbacks (precnd, matvec) per iteration, 17 per solve, two solves traced + the reference run of the operator alone
per_solve = 17
if len(syncs) < 2 * per_solve:
    print("unexpected trace: %d long event waits" % len(syncs)); sys.exit(1)
first = syncs[per_solve + 1 + 2 * (want_it - 1)]      # precnd download of iteration want_it + 1 of the second solve
last = syncs[per_solve + 1 + 2 * want_it]
# start a little earlier: at the Ritz sweep's launch before that wait
i0 = first
while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 2_000_000 and first - i0 < 60:
    i0 -= 1
t0 = ev[i0][0]
print(f"host-side timeline of Davidson iteration {want_it + 1} (second solve), times in ms from its first call")
print(f"{'start':>9} {'dur':>9}  call")
prev_end = None
for s, e, fn in ev[i0:last]:
    if prev_end is not None and s - prev_end > 200_000:
        print(f"{(prev_end - t0) / 1e6:9.3f} {(s - prev_end) / 1e6:9.3f}  -- no HIP call: the caller's routine / library host logic")
    if e - s > 50_000 or fn.startswith("hipMemcpy"):
        print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e6:9.3f}  {fn}")
    prev_end = e
print(f"{(ev[last][0] - t0) / 1e6:9.3f}            (next iteration's first download wait)")

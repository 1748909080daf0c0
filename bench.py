#!/usr/bin/env python3
"""bench.py -- headline benchmark of the diaglib hot path on MI355X.

Workload (BASELINE.json metric / SURVEY.md 8d): block Davidson-Liu on the matrix-free dense
symmetric operator A = diag(i+1) + 0.5 W W^T (rank 4), n = 2e6 rows, 8 wanted roots, block
n_max = 13, max_dav = 20, unit-vector guess, device-resident callbacks and eigenvector block.  Tolerance
2e-13 in the reference's sense (rms < tol and max < 10 tol): the loosest one for which rms * sqrt(n) / |lambda_min|
guarantees the north star's ||A x - lambda x||_2 / |lambda| <= 1e-10 (measured: 2-3e-12).
One "step" = one complete solve (guess -> converged eigenpairs).  With --gpus N the n rows are
sharded row-wise over N ranks (strong scaling, as the metric is quoted: same n on 1/2/4/8 GPUs);
the only cross-rank traffic is the RCCL all-reduce of the small m x m products.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = reference-schedule GFLOP (the flops of the
BLAS calls the reference would issue for the iterations performed -- counted per logical operation
at the library's entry points, dla_stats.ref_flops, so that a fused / pending / skipped sweep raises
it; `value_launched` is the per-launch count of what the engine ran) / wall time of the solves,
inputs resident in HBM.  `roofline` is the dominant kernel class measured with HIP events on the
engine's stream during the timed region; `cpu_baseline` times the UNMODIFIED reference
(oracle/_ref, flang+MKL) -- or the oracle's C port when that library cannot be loaded -- on this
box's host cores on the same workload (one solve at n = 2e6 rows, about 10 s of CPU work).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

def _usable_cpus() -> int:
    """Host cores this process may use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


HOST_CPUS = os.cpu_count() or 1
CPU_THREADS = int(os.environ.get("DIAGLIB_BENCH_CPU_THREADS", str(_usable_cpus())))
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))
os.environ.setdefault("MKL_NUM_THREADS", str(CPU_THREADS))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F64_PEAK_TFLOPS = 78.6  # dense FP64 matrix peak (256 CUs x 4 SIMDs x one 16x16x4 MFMA = 2048 flop per 64 cycles
                             # at 2.4 GHz; tools/mfma_probe.hip measures the 64 cycles); balance = 9.8 flop per HBM byte


def shard_rows(n: int, nranks: int, rank: int):
    """Contiguous row blocks, multiples of 64 rows except the last."""
    per = ((n + nranks - 1) // nranks + 63) // 64 * 64
    r0 = min(n, rank * per)
    r1 = min(n, r0 + per)
    return r0, r1 - r0


class _CaptureStdout:
    """The reference prints its timing table on Fortran unit 6: catch file descriptor 1 for the duration of the call."""

    def __enter__(self):
        import tempfile
        sys.stdout.flush()
        self.tmp = tempfile.TemporaryFile(mode="w+b")
        self.saved = os.dup(1)
        os.dup2(self.tmp.fileno(), 1)
        return self

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode(errors="replace")
        self.tmp.close()
        return False


def _reference_buckets(text: str):
    """(cpu, wall) pairs of the reference's four timers (diaglib.f90:1835-1841), wall seconds returned."""
    out = {}
    for key, pat in (("matvec", r"matrix-vector multiplications:\s+([\d.]+)\s+([\d.]+)"),
                     ("diag", r"diagonalization:\s+([\d.]+)\s+([\d.]+)"),
                     ("ortho", r"orthogonalization:\s+([\d.]+)\s+([\d.]+)"),
                     ("total", r"total:\s+([\d.]+)\s+([\d.]+)")):
        m = re.search(pat, text)
        if m:
            out[key] = float(m.group(2))
    return out or None


def _reference_iterations(text: str):
    """iterations of the reference's verbose table (format 1040 of diaglib.f90: `iter root eigenvalue rms max ok` rows): the largest
    iteration number printed"""
    its = [int(m.group(1)) for m in re.finditer(r"^\s+(\d+)\s+\d+\s+-?\d+\.\d{12}\s+\S+\s+\S+\s+[TF]\s*$", text, re.M)]
    return max(its) if its else None


def cpu_baseline(n_sample: int, n_targ: int, n_max: int, max_dav: int, tol: float, flops_per_row: float):
    """Reference (or port) on the host cores, bounded sample of the same workload."""
    from oracle.pyoracle import Oracle, Reference
    o = Oracle()
    o.synth_setup(n_sample, 0, n_sample)
    guess = np.zeros((n_sample, n_max), order="F")
    guess[np.arange(n_max), np.arange(n_max)] = 1.0
    mv, pc = o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd")
    kind, buckets, ref_iters = "port", None, None
    t0 = time.perf_counter()
    o.synth_counters(reset=True)
    try:
        ref = Reference()
        kind = "reference"
        with _CaptureStdout() as cap:
            t0 = time.perf_counter()
            _, _, ok = ref.davidson(n_sample, n_targ, n_max, 100, tol, max_dav, 0.0, mv, pc, guess, verbose=True)
            dt = time.perf_counter() - t0
            if hasattr(ref.lib, "ref_flush"):
                ref.lib.ref_flush()                    # the table sits in the Fortran runtime's buffer of unit 6
        buckets = _reference_buckets(cap.text)
        ref_iters = _reference_iterations(cap.text)
    except (OSError, FileNotFoundError):
        o.synth_counters(reset=True)
        t0 = time.perf_counter()
        _, _, ok, tr_ = o.davidson(n_sample, n_targ, n_max, 100, tol, max_dav, 0.0, mv, pc, guess)
        dt = time.perf_counter() - t0
        ref_iters = tr_.iters
    counters = o.synth_counters()
    flops = flops_per_row * n_sample
    # what the CPU run itself did (the shared flop numerator is the GPU run's reference-schedule count per row: it is the CPU run's
    # too when both took the same iterations over the same block widths -- checkable from these two numbers against config.iters /
    # config.matvec_cols of the same line)
    res = {"value": flops / dt / 1e9, "unit": "GFLOP/s", "cores": CPU_THREADS, "host_cpus": HOST_CPUS, "kind": kind,
           "iters": ref_iters, "matvec_cols": counters["matvec_cols"], "matvec_calls": counters["matvec_calls"],
           "precnd_cols": counters["precnd_cols"],
           "seconds_whole_call": round(dt, 3),
           "sample": f"same Davidson-Liu solve (synthetic operator, {n_targ} roots, n_max={n_max}, tol={tol:g}) at "
                     f"n={n_sample} rows, whole call incl. the reference's allocation + zero-fill of its panels, "
                     f"{dt:.2f} s on {CPU_THREADS} threads (MKL + OpenMP callbacks; {CPU_THREADS} = the cores this process may use: "
                     f"affinity mask / cgroup CPU quota of the box, which has {HOST_CPUS}), converged={bool(ok)}; "
                     "flops = the GPU run's reference-schedule flops per row (dla_stats.ref_flops: the BLAS calls of the reference "
                     "for the iterations performed, independent of what the engine launched) x n_sample"}
    if buckets and buckets.get("total"):
        # the reference's own timers (diaglib.f90:1835-1841): in-loop wall time and its three buckets; the rest of the
        # loop (projection, Ritz vectors, residuals) is un-bucketed in the reference
        res["seconds_in_loop"] = buckets["total"]
        res["value_in_loop"] = flops / buckets["total"] / 1e9
        res["buckets_s"] = {"matvec": buckets.get("matvec"), "diagonalization": buckets.get("diag"),
                            "orthogonalization": buckets.get("ortho"),
                            "unbucketed": round(buckets["total"] - sum(buckets.get(k, 0.0) for k in ("matvec", "diag", "ortho")), 4)}
    return res


def _launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` started without a launcher: run the N ranks under torch.distributed.run as a child
    process of this one (which has initialised neither torch nor the GPU, and is never replaced by exec), pass the child's
    stdout / stderr through -- rank 0 prints the JSON line -- and return its exit code (non-zero when any rank failed:
    torch.distributed.run ends all ranks and fails when one does)."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    for a in sys.argv[1:]:                           # "--n" is an abbreviation of the launcher's own "--nnodes": spell it "--rows"
        cmd.append("--rows" if a == "--n" else "--rows=" + a[4:] if a.startswith("--n=") else a)
    child = subprocess.Popen(cmd, cwd=os.getcwd())
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        try:
            return child.wait(timeout=30)
        except subprocess.TimeoutExpired:
            child.kill()
            return child.wait()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=2_000_000,
                    help="global rows (--rows: the spelling that survives torch.distributed.run's own option parser)")
    ap.add_argument("--roots", type=int, default=8)
    ap.add_argument("--solver", default="davidson", choices=["davidson", "lobpcg"])
    ap.add_argument("--tol", type=float, default=2e-13)
    ap.add_argument("--max-dav", type=int, default=20)
    ap.add_argument("--event-steps", type=int, default=1,
                    help="timed steps during which per-kernel HIP events are recorded (roofline figures)")
    ap.add_argument("--guess", default="unit", choices=["unit", "seed2"],
                    help="unit: e_1..e_M (SURVEY 8d guess (a), the headline); seed2: guess (b), uniform [-0.5,0.5) from the "
                         "documented generator on the leading --guess-rows rows (restart + locking)")
    ap.add_argument("--guess-rows", type=int, default=2000)
    ap.add_argument("--no-random-leg", action="store_true", help="skip the untimed-region seed-2 leg reported under config")
    ap.add_argument("--allreduce", default="p2p", choices=["p2p", "rccl"],
                    help="transport of the small cross-rank sums with --gpus > 1: one-shot peer-to-peer mailboxes over "
                         "hipIpc (falls back to RCCL when the mailboxes cannot be shared) or RCCL all-reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=2_000_000)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has touched neither torch nor the GPU; it starts
        # the ranks as a CHILD (one process per GPU under torch.distributed.run), relays their output and exit code
        raise SystemExit(_launch_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # rehearsal of the N-rank glue on a box with fewer GPUs than ranks: ranks share devices and the small products
    # are reduced through gloo (dla_set_allreduce_hook) instead of RCCL, which refuses two ranks per device
    rehearsal = world > 1 and bool(os.environ.get("DIAGLIB_BENCH_HOOK"))
    shared = world > 1 and bool(os.environ.get("DIAGLIB_BENCH_SHARE_GPU"))     # N ranks on fewer GPUs, peer-to-peer mailboxes
    if rehearsal or shared:
        local_rank = local_rank % torch.cuda.device_count()
        os.environ["LOCAL_RANK"] = str(local_rank)
        os.environ["DIAGLIB_AMD_SHARE_DEVICES"] = "1"
    torch.cuda.set_device(local_rank)
    if world > 1:
        # control plane (rendezvous, id broadcast, barriers, max-over-ranks of the time) on gloo;
        # the data path's collectives run on the engine's own RCCL communicator (ncclCommInitRank below)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from diaglib_amd import capi
    ctx = capi.Context()
    assert ctx.backend.startswith("hip:"), ctx.backend
    for kv in filter(None, os.environ.get("DIAGLIB_BENCH_TUNE", "").split(",")):     # A/B of engine knobs: "6=4,7=1"
        knob, val = kv.split("=")
        ctx.set_option(100 + int(knob), int(val))

    transport = None
    p2p_selftest = None                            # outcome of the mailbox self-test before anything is timed (N > 1)
    n, n_targ = args.n, args.roots
    n_max = min(2 * n_targ, n_targ + 5)            # harness convention, reference main.f90:354
    row0, n_loc = shard_rows(n, world, rank)
    if rehearsal:
        def hook(buf, op):
            tt = torch.from_numpy(buf)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        ctx.set_allreduce_hook(hook, world, rank)
        ctx.set_shard(n, row0)
    elif world > 1:
        transport = "rccl"
        if not shared:
            # RCCL communicator: the transport of --allreduce rccl, and of anything larger than a mailbox slot otherwise
            uid = [ctx.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(world, rank, uid[0])
        if args.allreduce == "p2p" or shared:
            # one-shot peer-to-peer all-reduce: export the mailbox, gather the handles on the control plane, attach;
            # all ranks take the same decision (a failure anywhere means RCCL everywhere)
            try:
                mine, okf = ctx.p2p_export(world), 1.0
            except capi.DlaError:
                mine, okf = b"\0" * 128, 0.0
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            flag = torch.tensor([okf], dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag[0]) > 0:
                try:
                    ctx.p2p_attach(world, rank, everyone)
                    okf = 1.0
                except capi.DlaError:
                    okf = 0.0
                flag = torch.tensor([okf], dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag[0]) > 0:
                    # self-test of the transport before anything is timed: a 2 x 2 product whose sum over ranks is known
                    try:
                        probe = ctx.panel(np.full((64, 2), float(rank + 1), order="F"))
                        got = ctx.gram(probe, probe)
                        probe.free()
                        okf = 1.0 if abs(got[0, 0] - 64.0 * sum((r + 1) ** 2 for r in range(world))) < 1e-9 else 0.0
                    except capi.DlaError:
                        okf = 0.0
                    flag = torch.tensor([okf], dtype=torch.float64)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                p2p_selftest = "passed" if float(flag[0]) > 0 else "failed on some rank: RCCL everywhere"
                if float(flag[0]) > 0:
                    transport = "p2p"
                elif shared:
                    raise SystemExit("peer-to-peer mailboxes could not be attached")
                else:
                    ctx.p2p_detach()              # RCCL everywhere
        ctx.set_shard(n, row0)
    elif os.environ.get("DIAGLIB_BENCH_FORCE_COMM"):
        # latency rehearsal of one shard of an N-GPU run: every small product goes through a 1-rank RCCL all-reduce
        ctx.comm_init(1, 0, ctx.unique_id())
        ctx.set_shard(n, 0)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ctx.synth_setup(n, row0, n_loc)
    mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")

    guess = np.zeros((n_loc, n_max), order="F")
    for j in range(n_max):                          # unit vectors e_1..e_M of the GLOBAL problem
        if row0 <= j < row0 + n_loc:
            guess[j - row0, j] = 1.0
    g_dev = ctx.panel(guess)
    ev = ctx.panel(n_loc, n_max)

    workload_key = f"{args.solver} n={n} roots={n_targ} n_max={n_max} max_dav={args.max_dav} guess={args.guess}"

    marker = ctx.panel(1, 1)

    def solve(max_iter=400):
        marker.zero()            # an 8-byte fill: separates the solves in a kernel trace (tools/kt_gaps.py)
        if args.guess == "unit":
            ctx.lib.dla_copy(ctx.h, ev.ptr, g_dev.ptr, 8 * n_loc * n_max)
        else:
            ctx.fill_guess(ev, 2, args.guess_rows)
        if args.solver == "davidson":
            return ctx.davidson_driver(n_loc, n_targ, n_max, max_iter, args.tol, args.max_dav, 0.0, mv, pc, ev)
        return ctx.lobpcg_driver(n_loc, n_targ, n_max, max_iter, args.tol, 0.0, mv, pc, ev)

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solve()
    # Kernel durations come from HIP events recorded on the engine's stream around every launch.  Each event is a
    # barrier packet (about 3.6 us; ~500 of them per solve = 1.8 ms of a 20 ms solve, measured), so they are recorded
    # during the first --event-steps timed steps only; the remaining timed steps run the same work without them.
    ev_steps = 0 if os.environ.get("DIAGLIB_BENCH_NOPROFILE") else max(1, min(args.event_steps, args.steps))
    ctx.reset_stats()
    ctx.set_option(capi.OPT_PROFILE, 1 if ev_steps else 0)
    ev_stats, ctx_kernel_stats = None, {}
    barrier()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        eig, _, ok, info = solve()
        if s_ + 1 == ev_steps:
            ev_stats, ctx_kernel_stats = ctx.stats(), ctx.kernel_stats()
            ctx.set_option(capi.OPT_PROFILE, 0)
    barrier()
    dt = time.perf_counter() - t0
    stats = ctx.stats()                             # flops / launch counts of ALL timed steps
    ctx.set_option(capi.OPT_PROFILE, 0)
    if ev_stats is None:
        ev_stats = stats
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    # ---- residual check of the returned pairs: ||A x - lambda x||_2 / |lambda| (north star <= 1e-10)
    ax = ctx.panel(n_loc, n_max)
    ctx.synth_matvec(ev, ax)
    x_h, ax_h = ev.download(), ax.download()
    r2 = ((ax_h - x_h * eig[None, :]) ** 2).sum(0)
    if world > 1:
        rt = torch.from_numpy(r2.copy())
        dist.all_reduce(rt)
        r2 = rt.numpy()
    rel_res = float((np.sqrt(r2[:n_targ]) / np.abs(eig[:n_targ])).max())

    rows_per_rank = [n_loc]
    if world > 1:
        rows_per_rank = [None] * world
        dist.all_gather_object(rows_per_rank, n_loc)
    # ---- flops / bytes (each rank counted its local rows; shards are equal up to 64 rows)
    classes = ["gram", "gemm", "trmm", "ritz", "elem"]
    # "GFLOP/s = reference-schedule flops of the iterations performed / wall time", the numerator from the solve's own
    # record of LOGICAL operations (dla_stats.ref_flops: per entry point the flops of the BLAS calls the reference issues there --
    # projection 2nLk, Ritz + residual 4nLM + 5nT, ortho_vs_x 2n(4Lk) + 15nk^2, ...), so that a sweep that is fused, left pending or
    # skipped RAISES the figure.  `value_launched` keeps the per-launch count of what the engine ran (it shrinks with every fusion).
    flops_total = stats["ref_flops"] * (n / n_loc)
    value = flops_total / dt / 1e9
    flops_launched = sum(stats[c]["flops"] for c in classes) * (n / n_loc)
    kst = ctx_kernel_stats
    # dominant kernel = the single kernel symbol with the largest HIP-event time in the timed region
    main = {k: v for k, v in kst.items() if v["alg_bytes"] > 0 and v["ms"] > 0}
    if not main:                                    # DIAGLIB_BENCH_NOPROFILE: wall time only
        if rank == 0:
            print(json.dumps({"ms_per_step": round(dt / args.steps * 1e3, 3), "value": round(value, 2), "n_gpus": world,
                              "iters": info["iters"], "allreduces": stats["allreduces"], "host_syncs": stats["host_syncs"],
                              "steps": args.steps, "allreduce_transport": transport, "p2p_selftest": p2p_selftest,
                              "rows_per_rank": rows_per_rank, "note": "kernel events disabled"}), flush=True)
        if os.environ.get("DIAGLIB_AMD_HOSTTIME") and world == 1:
            ctx.destroy()                # prints the host time spent inside every entry point
        return
    dom = max(main, key=lambda k: main[k]["ms"])
    dk = main[dom]
    cls_of = "gram" if dom.startswith("gram") else "ritz" if dom.startswith("ritz") else \
             ("trmm" if re.match(r"gemm_kernel<\d+, \d+, 2,", dom) else "gemm")
    # (kernel names carry every template argument, exactly as rocprofv3 prints them in profiles/r01/kernel_stats_*.csv)
    kern = {c: {"launches": ev_stats[c]["launches"], "ms": round(ev_stats[c]["ms"], 3),
                "GBps": round(ev_stats[c]["alg_bytes"] / max(ev_stats[c]["ms"], 1e-9) / 1e6, 1)}
            for c in classes + ["matvec", "precnd"]}
    per_kernel = {k: {"launches": v["launches"], "avg_us": round(v["ms"] / max(1, v["launches"]) * 1e3, 1),
                      "GBps": round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["alg_bytes"] > 0 else None}
                  for k, v in sorted(kst.items(), key=lambda kv: -kv[1]["ms"])[:10]}
    ach = dk["alg_bytes"] / max(dk["ms"], 1e-9) / 1e6
    # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py): only a figure collected for THIS workload and THIS
    # kernel symbol is quoted -- profiles/pmc_traffic.json is keyed by the workload string; anything else stays null
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic_source = None
    if os.path.exists(tfile) and world == 1:      # the PMC passes were taken at N = 1 (whole problem on one GPU)
        try:
            tj = json.load(open(tfile))
            traffic = tj.get(workload_key, {}).get(dom)
            if traffic is not None:       # (a separate rocprofv3 --pmc pass of this workload, not a counter of THIS run)
                traffic_source = "profiles/pmc_traffic.json (" + str(tj.get("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.py")) + ")"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launches": dk["launches"], "event_steps": ev_steps,
                "avg_launch_ms": round(dk["ms"] / max(1, dk["launches"]), 4),
                "alg_bytes_per_launch": round(dk["alg_bytes"] / max(1, dk["launches"]), 1)}
    # the wide-block sweeps of the other BASELINE shapes (n_max = 37: 74 output columns in the Ritz + P sweep) do more than the
    # machine balance of flops per byte: their roofline is the dense FP64 MFMA peak, not HBM
    intensity = dk.get("flops", 0.0) / max(dk["alg_bytes"], 1.0)
    roofline["flop_per_byte"] = round(intensity, 2)
    if intensity > MFMA_F64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        tf = dk["flops"] / max(dk["ms"], 1e-9) / 1e9
        roofline.update({"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_F64_PEAK_TFLOPS, 4), "hbm_GBps": round(ach, 1)})
    # the north star's "ortho/matvec step": every O(n) launch of ortho_vs_x (Gram, projection update, triangular update,
    # their reductions and k x k tail kernels, which move no panel bytes but take time) plus the operator; and the
    # whole solve: all algorithmic bytes over the wall time of the solve, host work included
    step_cls = ["gram", "gemm", "trmm", "matvec"]
    step_b = sum(ev_stats[c]["alg_bytes"] for c in step_cls)
    step_ms = sum(ev_stats[c]["ms"] for c in step_cls)
    all_cls = classes + ["matvec", "precnd"]
    all_b = sum(stats[c]["alg_bytes"] for c in all_cls) / args.steps
    # ... and the same classes without the launches that move no panel bytes (reductions of per-block partials, the k x k
    # steps of the device-driven chains, the cross-rank exchange): the rate of the sweeps themselves
    small_ms = sum(v["ms"] for k, v in kst.items()
                   if k.startswith(("gram_reduce_kernel", "ortho_tail_kernel", "ortho_tail16_kernel", "p2p_allreduce_kernel")))
    roofline["step"] = {"what": "ortho/matvec step = classes " + "+".join(step_cls) + " (HIP-event time, reductions included)",
                        "achieved": round(step_b / max(step_ms, 1e-9) / 1e6, 1),
                        "frac": round(step_b / max(step_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                        "sweeps_only": {"what": "the same without the reduction / k x k / exchange launches (no panel bytes)",
                                        "ms_excluded": round(small_ms, 3),
                                        "achieved": round(step_b / max(step_ms - small_ms, 1e-9) / 1e6, 1),
                                        "frac": round(step_b / max(step_ms - small_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}}
    roofline["solve"] = {"what": "all algorithmic bytes of one solve / wall time of the solve",
                         "alg_GB": round(all_b / 1e9, 2),
                         "achieved": round(all_b / (dt / args.steps) / 1e9, 1),
                         "frac": round(all_b / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    if rank == 0 and world == 1:
        try:
            triad = ctx.stream_triad(1 << 27, 5)    # 3 x 1 GiB, far beyond the 256 MiB Infinity Cache
            roofline["triad_GBps"] = round(triad, 1)
            roofline["frac_of_triad"] = round(ach / max(triad, 1e-9), 4)
        except capi.DlaError:
            roofline["triad_GBps"] = None

    out = {
        "metric": "eigensolver GFLOP/s + iters-to-converge, n=2e6 m=8 Davidson, 1/2/4/8 GPU",
        "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "value_launched": round(flops_launched / dt / 1e9, 2), "gflop_per_solve": round(flops_total / args.steps / 1e9, 3),
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.solver} n={n} roots={n_targ} n_max={n_max} max_dav={args.max_dav} tol={args.tol:g} "
                               f"operator=diag(i+1)+0.5*W*W^T(rank 4) guess={args.guess} callbacks=device",
                   "workload_key": workload_key,
                   "iters": info["iters"], "matvec_cols": info["matvec_cols"], "restarts": info["restarts"],
                   "converged": bool(ok), "max_rel_residual": rel_res, "rows_per_gpu": n_loc,
                   "eig": [round(float(e), 9) for e in eig[:n_targ]]},
        "roofline": roofline,
        "kernel_classes": kern,
        "kernels": per_kernel,
        "host": {"allreduces": stats["allreduces"], "host_syncs": stats["host_syncs"], "nproc": HOST_CPUS,
                 "allreduce_transport": (transport if world > 1 and not rehearsal else ("gloo hook (rehearsal)" if rehearsal else None)),
                 "p2p_selftest": p2p_selftest, "rows_per_rank": rows_per_rank,
                 "usable_cpus": CPU_THREADS},
    }
    if args.guess == "unit" and not args.no_random_leg:
        # second leg (SURVEY 8d guess (b)): the seed-2 random guess confined to the leading rows -- the run that restarts
        # and locks roots one by one.  Timed on its own (outside the K timed steps), reported beside the headline.
        args.guess = "seed2"
        solve()                                         # warm-up (allocator, speculation history)
        barrier()
        t1 = time.perf_counter()
        eig2, _, ok2, info2 = solve()
        barrier()
        dt2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt2], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = float(tt[0])
        ctx.synth_matvec(ev, ax)
        x_h, ax_h = ev.download(), ax.download()
        r2 = ((ax_h - x_h * eig2[None, :]) ** 2).sum(0)
        if world > 1:
            rt = torch.from_numpy(r2.copy())
            dist.all_reduce(rt)
            r2 = rt.numpy()
        out["config"]["random_guess_leg"] = {
            "guess": f"seed 2, uniform [-0.5,0.5) on the leading {args.guess_rows} rows (dla_fill_guess)",
            "ms": round(dt2 * 1e3, 3), "iters": info2["iters"], "matvec_cols": info2["matvec_cols"],
            "restarts": info2["restarts"], "converged": bool(ok2),
            "max_rel_residual": float((np.sqrt(r2[:n_targ]) / np.abs(eig2[:n_targ])).max()),
            "max_eig_diff_vs_unit_guess": float(np.abs(eig2[:n_targ] - eig[:n_targ]).max())}
        args.guess = "unit"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, n_targ, n_max, args.max_dav, args.tol,
                                               flops_total / args.steps / n)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": CPU_THREADS, "kind": "port",
                                   "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if (world > 1 and not rehearsal) or os.environ.get("DIAGLIB_BENCH_FORCE_COMM"):
        barrier()
        ctx.comm_finalize()
    if world > 1:
        dist.destroy_process_group()
    if os.environ.get("DIAGLIB_AMD_HOSTTIME"):
        for p_ in (ax, ev, g_dev):
            p_.free()
        ctx.destroy()                    # prints the engine's host-wait totals


if __name__ == "__main__":
    main()

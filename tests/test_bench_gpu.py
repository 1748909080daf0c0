"""The benchmark's own contract, on a small row count so that it runs in seconds: one JSON line, BASELINE.json's metric
string, the roofline / config objects the driver and the judge read, and the N > 1 launch exactly as the driver starts it
(torch.distributed.run, one rank per process) -- here with the ranks sharing the one GPU of the test box
(DIAGLIB_BENCH_SHARE_GPU, the rehearsal mode of tools/shard_rehearsal.sh), which takes the peer-to-peer transport through
its export / attach / self-test path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check(d: dict, n_gpus: int, steps: int, warmup: int, n: int) -> None:
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "GFLOP/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    cfg = d["config"]
    assert f"n={n}" in cfg["workload"] and "model" not in cfg
    assert cfg["converged"] is True and cfg["iters"] > 0 and cfg["max_rel_residual"] <= 1e-10
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0      # only the headline workload has PMC counters beside it


def test_bench_one_gpu_small():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "200000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-random-leg"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    _check(d, 1, 2, 1, 200000)
    assert d["roofline"]["traffic"] is None               # not the workload the committed counters belong to
    assert d["roofline"]["triad_GBps"] > 1000


def test_bench_two_ranks_as_the_driver_launches_them():
    env = dict(os.environ, DIAGLIB_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-random-leg"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["host"]["allreduce_transport"] == "p2p" and d["host"]["allreduces"] > 0
    assert d["host"]["p2p_selftest"] == "passed" and sum(d["host"]["rows_per_rank"]) == 200000       # the line describes its own run
    assert d["config"]["converged"] is True and d["config"]["rows_per_gpu"] in (100000, 100032, 99968)


def test_bench_two_ranks_started_without_a_launcher():
    """`python bench.py --gpus 2 ...` exactly as the driver starts the 1-GPU run: bench.py itself starts the ranks (a child
    torch.distributed.run; the parent touches neither torch nor the GPU), relays rank 0's line and the exit code."""
    env = dict(os.environ, DIAGLIB_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "200000", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-random-leg"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    _check(d, 2, 2, 1, 200000)
    assert d["host"]["allreduce_transport"] == "p2p" and d["host"]["p2p_selftest"] == "passed"
    assert sum(d["host"]["rows_per_rank"]) == 200000 and len(d["host"]["rows_per_rank"]) == 2
    # a rank that fails takes the exit code with it: an argument the ranks reject
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000", "--solver", "nope"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_value_counts_the_reference_schedule_not_the_launches():
    """`value` = reference-schedule flops of the iterations performed / wall time.  Two schedules of the orthogonalisation
    that launch different sweeps (tune knob 6: 12 = the five-sweep schedule, 13 = the three-pass one) converge in the same number
    of iterations and must report the SAME flops per solve -- while the per-launch count (`value_launched` x time) differs."""
    outs = []
    for knob in ("12", "13", "14"):                  # (14: dla_expand_project mode 5 behaves like mode 4 -- the same logical operations again)
        env = dict(os.environ, DIAGLIB_BENCH_TUNE="6=" + knob)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--no-cpu-baseline",
                            "--no-random-leg"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        outs.append(_line(p.stdout))
    a, b, c4 = outs
    assert a["config"]["iters"] == b["config"]["iters"] == c4["config"]["iters"] == 9
    assert c4["gflop_per_solve"] == pytest.approx(b["gflop_per_solve"], rel=2e-2)
    # (the same logical operations; the two schedules round differently, and a root whose residual sits at its threshold may leave
    #  the active block one iteration earlier or later: a column of one iteration is 1 % of the solve)
    assert a["gflop_per_solve"] == pytest.approx(b["gflop_per_solve"], rel=2e-2)
    la = a["value_launched"] * a["ms_per_step"]; lb = b["value_launched"] * b["ms_per_step"]
    assert abs(la - lb) > 1e-3 * la                  # the engine did launch different work
    assert a["value"] * a["ms_per_step"] == pytest.approx(a["gflop_per_solve"] * 1e3, rel=1e-3)


def test_headline_line_finds_its_counters():
    """On the headline workload the dominant kernel's HBM traffic must come from profiles/pmc_traffic.json: the file is keyed by
    workload and by the kernel name with every template argument, so a kernel whose template list changed without a new PMC
    pass shows up here as traffic = null."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-random-leg"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    _check(d, 1, 2, 1, 2000000)
    r = d["roofline"]
    assert r["traffic"] is not None, r["kernel"]
    assert 0.9 < r["traffic"] / r["alg_bytes_per_launch"] < 1.1          # no wasted re-reads
    assert d["config"]["iters"] == 9 and d["host"]["host_syncs"] / d["steps"] <= 19       # (the statistics are reset after the warm-up)
    assert r["step"]["sweeps_only"]["frac"] > r["step"]["frac"]


def test_shard_rehearsal_exchange_cost_is_bounded():
    """the latency floor of an N-GPU run, rehearsed on one GPU -- FOUR ranks share the device, every rank
    holds n / 4 rows, every small product crosses ranks through the peer-to-peer mailboxes (the transport of the real run).
    The ranks share one HBM, so the sweeps take what they take on one rank; what the 4-rank solve costs on top is the
    exchanges (66 per solve, each inside a reduction kernel) and the skew of four processes that time-share one device.
    Bound, derived from the exchange count: (exchanges per solve) x 20 us + 0.5 ms.  Per exchange: 17 us measured with four
    processes time-sharing one device (r03, events off: 9.50 -> 10.62 ms at n = 1e6 and 17.06 -> 18.20 ms at n = 2e6, 66 exchanges
    per solve either way; 23.6 us on the slowest box seen, 11.05 -> 12.61 ms) -- the exchange itself is one block writing four
    mailboxes and polling four flags, the rest is the device switching between the four processes' queues.  The 0.5 ms cover
    the skew of the ranks' host threads over a solve's 19 host waits.  tools/shard_rehearsal.sh prints the same figures."""
    n = 1_000_000
    env = dict(os.environ, DIAGLIB_BENCH_NOPROFILE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--rows", str(n), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-random-leg"]
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600,
                        cwd=ROOT, env=env)
    assert p1.returncode == 0, p1.stderr[-3000:]
    one = _line(p1.stdout)
    env4 = dict(env, DIAGLIB_BENCH_SHARE_GPU="1")
    p4 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                         "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "4"] + common,
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env4)
    assert p4.returncode == 0, p4.stderr[-3000:]
    four = _line(p4.stdout)
    assert four["n_gpus"] == 4 and four["allreduce_transport"] == "p2p" and four["p2p_selftest"] == "passed"
    assert sum(four["rows_per_rank"]) == n and four["iters"] == one["iters"]
    per_solve_exchanges = four["allreduces"] / four["steps"]
    assert per_solve_exchanges > 20
    allowed_ms = per_solve_exchanges * 0.020 + 0.5
    assert four["ms_per_step"] - one["ms_per_step"] <= allowed_ms, (four["ms_per_step"], one["ms_per_step"], per_solve_exchanges, allowed_ms)


def test_bench_four_ranks_full_row_count_with_a_short_last_shard():
    """The command the driver issues on an 8-GPU node, `bench.py --gpus N --rows 2000000`, with as many ranks as ONE card of this
    pool takes inside the suite: its process guard ends a run with more than 6 processes on the card (measured r06: five ranks +
    their launcher + this test process = 7, killed), so four ranks + launcher + this process.  Eight mailboxes cannot be rehearsed
    here; the 8-rank shard arithmetic, offsets and reduction points run in
    tests/test_hostsim.py::test_eight_ranks_gloo_at_the_cfg4_block_width_match_the_oracle, and tools/shard_rehearsal.sh 5 runs five
    ranks outside pytest.  The row count leaves the last shard short, as 2e6 rows over 8 ranks do (249 664 instead of 250 048):
    four mailboxes, four shard offsets of the generator and of the operator, unequal shards, the full-size panels -- converged
    to the benchmark's residual bound in the iteration count of the 1-rank run."""
    n = 2_000_000 - 64
    env = dict(os.environ, DIAGLIB_BENCH_SHARE_GPU="1", DIAGLIB_BENCH_NOPROFILE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                        "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rows", str(n),
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-random-leg"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 4 and d["allreduce_transport"] == "p2p" and d["p2p_selftest"] == "passed"
    rows = d["rows_per_rank"]
    assert sum(rows) == n and rows[:3] == [500032] * 3 and rows[3] == n - 3 * 500032
    assert d["iters"] in (9, 10) and d["allreduces"] > 0    # the 1-rank headline solve takes 9 (BENCH_r05); summation order differs

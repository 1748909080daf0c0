#!/usr/bin/env python3
"""Are the linear-response drivers row-shardable as they stand?  1 rank against 2 ranks (peer-to-peer mailboxes, one GPU) on the
built-in LR operators.    python tools/multirank_lr_probe.py"""
import json, os, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from diaglib_amd import capi
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from bench import shard_rows
n, t, m, trad = %(n)d, 4, 8, %(trad)r
row0, n_loc = shard_rows(n, world, rank)
ctx = capi.Context()
if world > 1:
    mine = ctx.p2p_export(world); everyone = [None] * world
    dist.all_gather_object(everyone, mine); ctx.p2p_attach(world, rank, everyone); ctx.set_shard(n, row0)
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, row0, n_loc)
g = np.zeros((2 * n_loc, m), order="F")
for j in range(m):
    if row0 <= j < row0 + n_loc: g[j - row0, j] = 1.0
ev = ctx.panel(g)
fns = [capi.fn_address(k) for k in ("dla_synth_apbmul", "dla_synth_ambmul", "dla_synth_spdmul", "dla_synth_smdmul", "dla_synth_lrprec1" if trad else "dla_synth_lrprec2")]
solve = ctx.caslr_driver if trad else ctx.caslr_eff_driver
eig, _, ok, info = solve(n_loc, t, m, 200, 1e-9, 10, *fns, ev)
print("RESULT", json.dumps(dict(rank=rank, ok=bool(ok), iters=info["iters"], eig=[float(x) for x in eig[:t]])), flush=True)
dist.barrier()
if world > 1: ctx.comm_finalize()
dist.destroy_process_group()
'''
def run(world, n, trad):
    td = tempfile.mkdtemp()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = os.path.join(td, "w.py"); open(script, "w").write(WORKER % dict(root=ROOT, n=n, trad=trad))
    procs = [subprocess.Popen([sys.executable, script], env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    got = []
    for p, (o, e) in zip(procs, outs):
        res = [l for l in o.splitlines() if l.startswith("RESULT")]
        print(world, "ranks:", res[0] if res else ("rc %d " % p.returncode) + " | ".join((o + e).splitlines()[-4:]), flush=True)
        got.append(json.loads(res[0][7:]) if res else None)
    return got
bad = 0
for trad in (False, True):
    for n, worlds in ((200000, (2, 4)), (200001, (3,))):
        one = run(1, n, trad)[0]
        for w in worlds:
            many = run(w, n, trad)
            good = one is not None and all(r is not None and r["ok"] and r["eig"] == many[0]["eig"] and r["iters"] == many[0]["iters"] for r in many) and \
                max(abs(a / b - 1.0) for a, b in zip(many[0]["eig"], one["eig"])) < 1e-11 and many[0]["iters"] == one["iters"]
            bad += 0 if good else 1
print("linear-response drivers on row shards:", "ok" if not bad else f"{bad} FAILURES", flush=True)
sys.exit(1 if bad else 0)

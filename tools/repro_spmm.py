import os, sys, json, subprocess, tempfile, socket
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_spmm_sharded as tm
spec = dict(backend="hip", transport=sys.argv[1] if len(sys.argv) > 1 else "hook", n=200_000, n_targ=6, n_max=10, half_band=6, tol=1e-9)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    script = td / "worker.py"
    script.write_text(tm.WORKER.format(root=ROOT, spec=json.dumps(spec), out=str(td)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=600)
        print("rank", r, "rc", p.returncode); print(o[-2500:]); print(e[-800:])

import os, sys, json, subprocess, tempfile, socket
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_multirank_gpu as tm
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
spec = {'n': 235047, 'n_targ': 29, 'n_max': 36, 'tol': 1e-08, 'solver': 'lobpcg', 'guess': 'unit', 'transport': 'hook'}
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    script = td / "worker.py"
    script.write_text(tm.WORKER.format(root=ROOT, spec=json.dumps(spec), out=str(td)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True)
    print("rc", p.returncode); print(p.stdout[-3000:]); print(p.stderr[-1500:])

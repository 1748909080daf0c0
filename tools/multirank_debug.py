import os, sys, tempfile, json
from pathlib import Path
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
import test_multirank_gpu as tm
import subprocess, socket
spec = json.loads(sys.argv[1]); world = int(sys.argv[2])
td = tempfile.mkdtemp()
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
script = Path(td) / "worker.py"
script.write_text(tm.WORKER.format(root=ROOT, spec=json.dumps(spec), out=td))
procs = []
for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
outs = [p.communicate(timeout=600) for p in procs]
for r, (p, (o, e)) in enumerate(zip(procs, outs)):
    print("== rank", r, "rc", p.returncode)
    print("\n".join([l for l in (o + e).splitlines() if "amdgpu.ids" not in l and "socket.cpp" not in l][-12:]))

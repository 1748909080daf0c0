#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference.

Run in the build container (needs /root/reference -> oracle/_ref via `make -C oracle ref`):

    python tests/golden/make_golden.py

Every fixture is data only: seeded inputs (or the recipe to rebuild them) and the outputs the
reference (Molecolab-Pisa/diaglib compiled with flang 22 + MKL LP64 from /opt/conda/lib, see
oracle/Makefile) produced for them.  The reference's verbose convergence table (its own trace
format, diaglib.f90:1671,1752) is captured from stdout and parsed into arrays.
"""
import io
import json
import os
import re
import subprocess
import sys

os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("MKL_NUM_THREADS", "8")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

ROW = re.compile(r"^\s+(\d+)\s+(\d+)\s+(-?\d+\.\d+)\s+([0-9.DE+-]+)\s+([0-9.DE+-]+)\s+([TF])\s*$")


def fnum(s):
    return float(s.replace("D", "E"))


def parse_trace(text, n_targ):
    its = {}
    for line in text.splitlines():
        m = ROW.match(line)
        if m:
            it, root = int(m.group(1)), int(m.group(2))
            its.setdefault(it, {})[root] = (float(m.group(3)), fnum(m.group(4)), fnum(m.group(5)), m.group(6) == "T")
    n_it = max(its) if its else 0
    eig = np.zeros((n_it, n_targ)); rms = np.zeros((n_it, n_targ)); rmx = np.zeros((n_it, n_targ))
    done = np.zeros((n_it, n_targ), np.int32)
    for it, rows in its.items():
        for root, (e, a, b, d) in rows.items():
            eig[it - 1, root - 1] = e; rms[it - 1, root - 1] = a; rmx[it - 1, root - 1] = b; done[it - 1, root - 1] = d
    n_act = [int(x) for x in re.findall(r"# new vectors added:\s+(\d+)", text)]
    return dict(iters=n_it, eig=eig, rms=rms, rmax=rmx, done=done, n_act_added=np.array(n_act, np.int32),
                restarts=text.count("Restarting davidson."))


def run_driver_child(spec):
    """Run one reference driver call in a child process (so that its Fortran stdout can be captured)."""
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, Reference
spec = json.loads(%r)
o = Oracle(); r = Reference()
n, T, M = spec['n'], spec['n_targ'], spec['n_max']
if spec['op'] == 'dense':
    o.dense_setup(n); mv, pc = o.fn('orc_dense_matvec'), o.fn('orc_dense_precnd')
else:
    o.synth_setup(n, 0, n); mv, pc = o.fn('orc_synth_matvec'), o.fn('orc_synth_precnd')
g = np.load(spec['guess'])
if spec.get('gen'):
    o.metric_setup(n); bv = o.fn('orc_metric_matvec')
if spec['solver'] == 'gen_davidson':
    e, v, ok = r.gen_davidson(n, T, M, spec['max_iter'], spec['tol'], spec['max_dav'], spec['shift'], mv, pc, bv, g, verbose=True)
elif spec['solver'] == 'lobpcg' and spec.get('gen'):
    e, v, ok = r.lobpcg(n, T, M, spec['max_iter'], spec['tol'], spec['shift'], mv, pc, g, verbose=True, bvec=bv)
elif spec['solver'] == 'davidson':
    e, v, ok = r.davidson(n, T, M, spec['max_iter'], spec['tol'], spec['max_dav'], spec['shift'], mv, pc, g, verbose=True)
else:
    e, v, ok = r.lobpcg(n, T, M, spec['max_iter'], spec['tol'], spec['shift'], mv, pc, g, verbose=True)
sys.stdout.flush()
np.savez(spec['out'], eig=e, evec=v[:, :T], ok=ok)
""" % (ROOT, json.dumps(spec))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(p.stderr)
    return p.stdout


def guess_array(kind, n, m, seed):
    if kind == "unit":
        g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        return g
    return np.asfortranarray(np.random.default_rng(seed).random((n, m)) - 0.5)


def main():
    from oracle.pyoracle import Reference
    ref = Reference()
    rng = np.random.default_rng(2024)
    out = {}

    # ---- F1 ortho_cd: well / ill conditioned / rank deficient (shift ladder)
    cases = []
    for (n, k, cond) in [(257, 1, 1.0), (257, 5, 1e2), (257, 13, 1e6), (300, 13, 1e12)]:
        q, _ = np.linalg.qr(rng.standard_normal((n, k)))
        s = np.logspace(0, -np.log10(cond), k) if k > 1 else np.ones(1)
        w, _ = np.linalg.qr(rng.standard_normal((k, k)))
        u = np.asfortranarray((q * s) @ w.T)
        uo, g, ok = ref.ortho_cd(u)
        cases.append((u, uo, g, ok))
    u = np.asfortranarray(rng.standard_normal((257, 6))); u[:, 5] = u[:, 0] + u[:, 1]
    uo, g, ok = ref.ortho_cd(u)
    cases.append((u, uo, g, ok))
    for i, (u, uo, g, ok) in enumerate(cases):
        out[f"ocd{i}_in"] = u; out[f"ocd{i}_out"] = uo; out[f"ocd{i}_growth"] = g; out[f"ocd{i}_ok"] = ok
    out["ocd_count"] = len(cases)

    # ---- F2 ortho_vs_x
    cases = []
    for (n, m, k, mix) in [(257, 13, 13, 5.0), (257, 39, 13, 0.0), (300, 26, 5, 50.0)]:
        x = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, m)))[0])
        u = np.asfortranarray(rng.standard_normal((n, k)) + mix * x[:, :k] @ rng.standard_normal((k, k)))
        cases.append((x, u, ref.ortho_vs_x(x, u)))
    for i, (x, u, uo) in enumerate(cases):
        out[f"ovx{i}_x"] = x; out[f"ovx{i}_u"] = u; out[f"ovx{i}_out"] = uo
    out["ovx_count"] = len(cases)

    # ---- F3 b_ortho / b_ortho_vs_x
    n, m, k = 200, 8, 5
    a = rng.standard_normal((n, n)) * 0.02
    b = a @ a.T + np.eye(n)
    x = rng.standard_normal((n, m)); x = x @ np.linalg.inv(np.linalg.cholesky(x.T @ b @ x)).T
    x = np.asfortranarray(x); bx = np.asfortranarray(b @ x)
    u = np.asfortranarray(rng.standard_normal((n, k)))
    uo = ref.b_ortho_vs_x(x, bx, u)
    bu = np.asfortranarray(b @ uo)
    u2, bu2 = ref.b_ortho(uo, bu)
    out.update(bo_x=x, bo_bx=bx, bo_u=u, bo_vsx_out=uo, bo_bu=bu, bo_u_out=u2, bo_bu_out=bu2)

    # ---- private helpers: norm_est, get_coeffs, check_guess (non-zero guess branch)
    lmat = np.asfortranarray(np.tril(rng.standard_normal((9, 9))) + 2 * np.eye(9) + np.triu(rng.standard_normal((9, 9)), 1))
    out["ne_in"] = lmat; out["ne_out"] = ref.norm_est(lmat)
    n_max, n_act = 6, 4
    len_u, len_a = n_max + 2 * n_act, 3 * n_max
    q, _ = np.linalg.qr(rng.standard_normal((len_u, len_u)))
    q = q * np.sign(np.diag(q))        # positive diagonal, like the eigenvector blocks get_coeffs sees
    a_red = np.zeros((len_a, len_a), order="F"); a_red[:len_u, :len_u] = q
    ux, up = ref.get_coeffs(a_red, len_u, n_max, n_act)
    out.update(gc_a_red=a_red, gc_len_u=len_u, gc_n_max=n_max, gc_n_act=n_act, gc_ux=ux, gc_up=up)
    g = np.asfortranarray(rng.random((257, 6)) - 0.5)
    out["cg_in"] = g; out["cg_out"] = ref.check_guess(g)
    e = np.zeros((257, 6), order="F"); e[np.arange(6), np.arange(6)] = 1.0
    out["cg_unit_unchanged"] = bool(np.array_equal(ref.check_guess(e), e))

    # ---- F4/F5/F6 drivers on the reference's dense test matrix a_ii=i+1, a_ij=1/(i+j)
    drv = []
    tmp = os.path.join(HERE, "_tmp"); os.makedirs(tmp, exist_ok=True)
    specs = [
        dict(name="dav_n1000_unit", solver="davidson", op="dense", n=1000, n_targ=10, n_max=15, max_dav=20, guess="unit"),
        dict(name="dav_n2000_unit", solver="davidson", op="dense", n=2000, n_targ=4, n_max=8, max_dav=20, guess="unit"),
        dict(name="dav_n2000_rand", solver="davidson", op="dense", n=2000, n_targ=4, n_max=8, max_dav=20, guess="rand"),
        dict(name="dav_n600_rand_dav10", solver="davidson", op="dense", n=600, n_targ=6, n_max=11, max_dav=10, guess="rand"),
        dict(name="lob_n1000_unit", solver="lobpcg", op="dense", n=1000, n_targ=10, n_max=15, max_dav=0, guess="unit"),
        dict(name="lob_n2000_unit", solver="lobpcg", op="dense", n=2000, n_targ=4, n_max=8, max_dav=0, guess="unit"),
        dict(name="lob_n2000_rand", solver="lobpcg", op="dense", n=2000, n_targ=4, n_max=8, max_dav=0, guess="rand"),
        dict(name="lob_n800_shift", solver="lobpcg", op="dense", n=800, n_targ=3, n_max=6, max_dav=0, guess="rand", shift=0.75),
        dict(name="gdav_n600_unit", solver="gen_davidson", op="dense", gen=True, n=600, n_targ=4, n_max=8, max_dav=20, guess="unit"),
        dict(name="gdav_n600_rand", solver="gen_davidson", op="dense", gen=True, n=600, n_targ=4, n_max=8, max_dav=20, guess="rand"),
        dict(name="glob_n600_unit", solver="lobpcg", op="dense", gen=True, n=600, n_targ=4, n_max=8, max_dav=0, guess="unit"),
        dict(name="dav_synth_n100000", solver="davidson", op="synth", n=100000, n_targ=8, n_max=13, max_dav=20, guess="unit"),
        dict(name="lob_synth_n100000", solver="lobpcg", op="synth", n=100000, n_targ=8, n_max=13, max_dav=0, guess="unit"),
    ]
    for i, sp in enumerate(specs):
        sp.setdefault("shift", 0.0); sp["tol"] = 1e-8; sp["max_iter"] = 200
        seed = 1000 + i
        g = guess_array(sp["guess"], sp["n"], sp["n_max"], seed)
        gpath = os.path.join(tmp, "g.npy"); np.save(gpath, g)
        opath = os.path.join(tmp, "o.npz")
        text = run_driver_child(dict(sp, guess=gpath, out=opath))
        res = np.load(opath)
        tr = parse_trace(text, sp["n_targ"])
        nm = sp["name"]
        out[nm + "_eig"] = res["eig"]; out[nm + "_ok"] = bool(res["ok"])
        # eigenvectors are stored for the small dense cases only (fixture size)
        if sp["op"] == "dense" and sp["n"] <= 2000:
            ev = res["evec"]; ev = ev * np.sign(ev[np.abs(ev).argmax(0), np.arange(ev.shape[1])])
            out[nm + "_evec"] = ev.astype(np.float64)
        for k2 in ("iters", "eig", "rms", "rmax", "done", "n_act_added", "restarts"):
            out[nm + "_tr_" + k2] = tr[k2]
        drv.append(dict(sp, seed=seed))
        print(nm, "iters", tr["iters"], "restarts", tr["restarts"], "ok", bool(res["ok"]))
    # dense LAPACK cross-check values (main.f90:321-342 writes these to lapack.txt)
    for n in (1000, 2000):
        idx = np.arange(1, n + 1, dtype=np.float64)
        a = 1.0 / (idx[:, None] + idx[None, :]); np.fill_diagonal(a, idx + 1.0)
        out[f"dense_eigs_n{n}"] = np.linalg.eigvalsh(a)[:15]
    out["driver_specs"] = json.dumps(drv)
    out["provenance"] = ("reference: Molecolab-Pisa/diaglib @ /root/reference (unmodified), flang 22 (ROCm 7.2) -O2 -fopenmp, "
                         "MKL LP64 /opt/conda/lib (mkl_gf_lp64 + mkl_gnu_thread + mkl_core), 8 threads")
    np.savez_compressed(os.path.join(HERE, "reference_fixtures.npz"), **out)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    print("wrote", os.path.join(HERE, "reference_fixtures.npz"))


if __name__ == "__main__":
    main()

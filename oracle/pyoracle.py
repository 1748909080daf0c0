"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes doors to (a) ``liboracle.so``, the plain-C restatement of the reference's
Davidson-Liu / LOBPCG hot path (oracle.c) and (b) ``_ref/libdiaglib_ref.so``, the
UNMODIFIED reference compiled by oracle/Makefile (flang + MKL).  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this
module; the product package ``diaglib_amd`` never does.

All arrays are Fortran-ordered float64 (column-major, ld = n) like the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libdiaglib_ref.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def build(verbose: bool = False) -> None:
    """Compile the C restatement and, when /root/reference is present, the reference."""
    out = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout, out.stderr)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed")


def _f(a: np.ndarray) -> np.ndarray:
    assert a.dtype == np.float64 and (a.flags.f_contiguous or a.ndim == 1), "need Fortran-ordered float64"
    return a


def _p(a: np.ndarray):
    return a.ctypes.data_as(c_dp)


class _Trace(C.Structure):
    _fields_ = [
        ("iters", C.c_int), ("matvec_cols", C.c_int), ("restarts", C.c_int),
        ("n_act", c_ip), ("ldu", c_ip), ("eig", c_dp), ("rms", c_dp), ("rmax", c_dp), ("done", c_ip),
    ]


@dataclass
class Trace:
    iters: int = 0
    matvec_cols: int = 0
    restarts: int = 0
    n_act: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    ldu: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    eig: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    rms: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    rmax: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    done: np.ndarray = field(default_factory=lambda: np.zeros((0, 0), np.int32))


class Oracle:
    """The C restatement (oracle.c / oracle_ops.c)."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.orc_norm_est.restype = C.c_double
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_ulonglong] * 3
        L.orc_synth_setup.argtypes = [C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_double]
        L.orc_synth_w.restype = c_dp
        L.orc_synth_diag.restype = c_dp
        L.orc_davidson.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                                   c_dp, c_dp, c_ip, C.c_void_p]
        L.orc_lobpcg.argtypes = [C.c_int] * 5 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                                 c_dp, c_dp, c_ip, C.c_void_p]
        L.orc_lobpcg_gen.argtypes = [C.c_int] * 5 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     c_dp, c_dp, c_ip, C.c_void_p]
        L.orc_gen_davidson.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       c_dp, c_dp, c_ip, C.c_void_p]
        L.orc_metric.restype = c_dp
        L.orc_gemm_nn.argtypes = [C.c_int] * 3 + [C.c_double, c_dp, C.c_int, c_dp, C.c_int, C.c_double, c_dp, C.c_int]

    # ---- callbacks as raw addresses (usable by oracle, reference and product alike)
    def fn(self, name: str) -> int:
        return C.cast(getattr(self.lib, name), C.c_void_p).value

    # ---- small dense
    def potrf_lower(self, a):
        a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
        info = self.lib.orc_potrf_lower(a.shape[0], _p(a), a.shape[0])
        return a, info

    def trtri_lower(self, a):
        a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
        info = self.lib.orc_trtri_lower(a.shape[0], _p(a), a.shape[0])
        return a, info

    def syev(self, a, uplo="l"):
        a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
        n = a.shape[0]
        w = np.zeros(n)
        self.lib.orc_syev(C.c_char(uplo.encode()), n, _p(a), n, _p(w))
        return w, a

    def norm_est(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        return float(self.lib.orc_norm_est(a.shape[0], _p(a), a.shape[0]))

    # ---- panels
    def gemm_tn(self, x, u):
        x = np.asfortranarray(x); u = np.asfortranarray(u)
        n, l = x.shape; k = u.shape[1]
        c = np.zeros((l, k), order="F")
        self.lib.orc_gemm_tn(n, l, k, _p(x), n, _p(u), n, _p(c), l)
        return c

    def gemm_nn(self, x, c, alpha=1.0, beta=0.0, z=None):
        x = np.asfortranarray(x); c = np.asfortranarray(c)
        n, l = x.shape; k = c.shape[1]
        z = np.zeros((n, k), order="F") if z is None else np.asfortranarray(z).copy(order="F")
        self.lib.orc_gemm_nn(n, l, k, alpha, _p(x), n, _p(c), c.shape[0], beta, _p(z), n)
        return z

    # ---- ortho
    def ortho_cd(self, u):
        u = np.asfortranarray(u, dtype=np.float64).copy(order="F")
        g = C.c_double(0.0); ok = C.c_int(0); nm = C.c_int(0)
        self.lib.orc_ortho_cd(u.shape[0], u.shape[1], _p(u), C.byref(g), C.byref(ok), C.byref(nm))
        return u, g.value, bool(ok.value), nm.value

    def ortho_qr(self, u):
        u = np.asfortranarray(u, dtype=np.float64).copy(order="F")
        self.lib.orc_ortho_qr(u.shape[0], u.shape[1], _p(u))
        return u

    def ortho_vs_x(self, x, u):
        x = np.asfortranarray(x, dtype=np.float64)
        u = np.asfortranarray(u, dtype=np.float64).copy(order="F")
        no = C.c_int(0)
        st = self.lib.orc_ortho_vs_x(x.shape[0], x.shape[1], u.shape[1], _p(x), _p(u), C.byref(no))
        return u, no.value, st

    def b_ortho(self, u, bu):
        u = np.asfortranarray(u).copy(order="F"); bu = np.asfortranarray(bu).copy(order="F")
        self.lib.orc_b_ortho(u.shape[0], u.shape[1], _p(u), _p(bu))
        return u, bu

    def b_ortho_vs_x(self, x, bx, u):
        x = np.asfortranarray(x); bx = np.asfortranarray(bx)
        u = np.asfortranarray(u).copy(order="F")
        st = self.lib.orc_b_ortho_vs_x(x.shape[0], x.shape[1], u.shape[1], _p(x), _p(bx), _p(u))
        return u, st

    def check_guess(self, evec):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        self.lib.orc_check_guess(evec.shape[0], evec.shape[1], _p(evec))
        return evec

    def get_coeffs(self, a_red, len_u, n_max, n_act):
        a_red = np.asfortranarray(a_red)
        len_a = a_red.shape[0]
        u_x = np.zeros((len_u, n_max), order="F"); u_p = np.zeros((len_u, max(n_act, 1)), order="F")
        self.lib.orc_get_coeffs(len_a, len_u, n_max, n_act, _p(a_red), _p(u_x), _p(u_p))
        return u_x, u_p[:, :n_act]

    # ---- operators
    def dense_setup(self, n):
        self.lib.orc_dense_setup(n)

    def metric_setup(self, n):
        self.lib.orc_metric_setup(n)
        return np.ctypeslib.as_array(self.lib.orc_metric(), (n, n)).copy(order="F")

    def lr_setup(self, n):
        """linear-response test problem; returns the dense matrices (A+B, A-B, S+D, S-D)"""
        self.lib.orc_lr_setup(n)
        self.lib.orc_lr_matrix.restype = c_dp
        return tuple(np.ctypeslib.as_array(self.lib.orc_lr_matrix(w), (n, n)).T.copy(order="F") for w in range(4))

    def synth_setup(self, n_global, row0, n_local, rank_w=4, sigma=0.5):
        self.lib.orc_synth_setup(n_global, row0, n_local, rank_w, sigma)
        self._synth = (n_local, rank_w)

    def synth_counters(self, reset=False):
        """calls / block columns the synthetic operator's matvec and precnd have seen since the last reset"""
        out = (C.c_longlong * 4)()
        self.lib.orc_synth_counters(int(reset), out)
        return dict(matvec_calls=int(out[0]), matvec_cols=int(out[1]), precnd_calls=int(out[2]), precnd_cols=int(out[3]))

    def synth_w(self):
        n, r = self._synth
        return np.ctypeslib.as_array(self.lib.orc_synth_w(), (r, n)).T.copy(order="F")

    def synth_diag(self):
        n, _ = self._synth
        return np.ctypeslib.as_array(self.lib.orc_synth_diag(), (n,)).copy()

    def u01(self, seed, i, j):
        return float(self.lib.orc_u01(seed, i, j))

    @staticmethod
    def guess_u01(seed, n, m, row0=0, offset=-0.5):
        """n x m block of orc_u01(seed, row0+i+1, j+1) + offset, vectorised restatement of oracle.c:orc_u01
        (64-bit wrap-around arithmetic); SURVEY 8d guess (b) is seed 2, offset -0.5."""
        with np.errstate(over="ignore"):
            i = (np.arange(n, dtype=np.uint64) + np.uint64(row0 + 1))[:, None]
            j = (np.arange(m, dtype=np.uint64) + np.uint64(1))[None, :]
            z = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + i * np.uint64(0xBF58476D1CE4E5B9) \
                + j * np.uint64(0x94D049BB133111EB)
            z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
            z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
            z ^= z >> np.uint64(31)
        return np.asfortranarray((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) + offset)

    # ---- drivers
    def _mk_trace(self, max_iter, n_targ):
        t = Trace(n_act=np.zeros(max_iter, np.int32), ldu=np.zeros(max_iter, np.int32),
                  eig=np.zeros((max_iter, n_targ)), rms=np.zeros((max_iter, n_targ)),
                  rmax=np.zeros((max_iter, n_targ)), done=np.zeros((max_iter, n_targ), np.int32))
        ct = _Trace(0, 0, 0, t.n_act.ctypes.data_as(c_ip), t.ldu.ctypes.data_as(c_ip), _p(t.eig), _p(t.rms),
                    _p(t.rmax), t.done.ctypes.data_as(c_ip))
        return t, ct

    @staticmethod
    def _fin_trace(t, ct):
        t.iters, t.matvec_cols, t.restarts = ct.iters, ct.matvec_cols, ct.restarts
        for name in ("n_act", "ldu", "eig", "rms", "rmax", "done"):
            setattr(t, name, getattr(t, name)[: t.iters])
        return t

    def davidson(self, n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd, evec, verbose=False):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        t, ct = self._mk_trace(max_iter, n_targ)
        self.lib.orc_davidson(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd,
                              _p(eig), _p(evec), C.byref(ok), C.addressof(ct))
        return eig, evec, bool(ok.value), self._fin_trace(t, ct)

    def lobpcg(self, n, n_targ, n_max, max_iter, tol, shift, matvec, precnd, evec, verbose=False):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        t, ct = self._mk_trace(max_iter, n_targ)
        self.lib.orc_lobpcg(int(verbose), n, n_targ, n_max, max_iter, tol, shift, matvec, precnd,
                            _p(eig), _p(evec), C.byref(ok), C.addressof(ct))
        return eig, evec, bool(ok.value), self._fin_trace(t, ct)


def _gen_methods():
    def lobpcg_gen(self, n, n_targ, n_max, max_iter, tol, shift, matvec, precnd, bvec, evec, verbose=False):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        t, ct = self._mk_trace(max_iter, n_targ)
        self.lib.orc_lobpcg_gen(int(verbose), n, n_targ, n_max, max_iter, tol, shift, matvec, precnd, bvec,
                                _p(eig), _p(evec), C.byref(ok), C.addressof(ct))
        return eig, evec, bool(ok.value), self._fin_trace(t, ct)

    def gen_davidson(self, n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd, bvec, evec, verbose=False):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        t, ct = self._mk_trace(max_iter, n_targ)
        self.lib.orc_gen_davidson(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd, bvec,
                                  _p(eig), _p(evec), C.byref(ok), C.addressof(ct))
        return eig, evec, bool(ok.value), self._fin_trace(t, ct)

    Oracle.lobpcg_gen = lobpcg_gen
    Oracle.gen_davidson = gen_davidson


_gen_methods()


class Reference:
    """The unmodified reference, compiled by oracle/Makefile into _ref/ (flang + MKL)."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.ref_davidson.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                                   c_dp, c_dp, c_ip]
        L.ref_lobpcg.argtypes = [C.c_int] * 6 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 c_dp, c_dp, c_ip]
        if hasattr(L, "ref_gen_david"):
            L.ref_gen_david.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        c_dp, c_dp, c_ip]
        # private module procedures are still global symbols in the object (flang mangling)
        self._norm_est = getattr(L, "_QMdiaglibPnorm_est")
        self._norm_est.restype = C.c_double
        self._get_coeffs = getattr(L, "_QMdiaglibPget_coeffs")
        self._check_guess = getattr(L, "_QMdiaglibPcheck_guess")

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def ortho_cd(self, u):
        u = np.asfortranarray(u, dtype=np.float64).copy(order="F")
        g = C.c_double(0.0); ok = C.c_int(0)
        self.lib.ref_ortho_cd(u.shape[0], u.shape[1], _p(u), C.byref(g), C.byref(ok))
        return u, g.value, bool(ok.value)

    def ortho_vs_x(self, x, u):
        x = np.asfortranarray(x, dtype=np.float64)
        u = np.asfortranarray(u, dtype=np.float64).copy(order="F")
        self.lib.ref_ortho_vs_x(x.shape[0], x.shape[1], u.shape[1], _p(x), _p(u))
        return u

    def b_ortho(self, u, bu):
        u = np.asfortranarray(u).copy(order="F"); bu = np.asfortranarray(bu).copy(order="F")
        self.lib.ref_b_ortho(u.shape[0], u.shape[1], _p(u), _p(bu))
        return u, bu

    def b_ortho_vs_x(self, x, bx, u):
        x = np.asfortranarray(x); bx = np.asfortranarray(bx)
        u = np.asfortranarray(u).copy(order="F")
        self.lib.ref_b_ortho_vs_x(x.shape[0], x.shape[1], u.shape[1], _p(x), _p(bx), _p(u))
        return u

    def norm_est(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        m = C.c_int(a.shape[0])
        return float(self._norm_est(C.byref(m), _p(a)))

    def ortho(self, u):
        """the reference's Householder fallback (diaglib.f90:3052-3092), through its module symbol"""
        u = np.asfortranarray(u, dtype=np.float64).copy(order="F")
        n, m = (C.c_int(v) for v in u.shape)
        w = np.zeros(1)
        getattr(self.lib, "_QMdiaglibPortho")(C.byref(n), C.byref(m), _p(u), _p(w))
        return u

    def set_i_alg(self, value):
        """the harness switch of caslr_driver (module utils, reference utils.f90:7; read at diaglib.f90:675)"""
        C.c_int.in_dll(self.lib, "_QMutilsEi_alg").value = int(value)

    def get_coeffs(self, a_red, len_u, n_max, n_act):
        a_red = np.asfortranarray(a_red)
        la, lu, nm, na = (C.c_int(v) for v in (a_red.shape[0], len_u, n_max, n_act))
        u_x = np.zeros((len_u, n_max), order="F"); u_p = np.zeros((len_u, max(n_act, 1)), order="F")
        self._get_coeffs(C.byref(la), C.byref(lu), C.byref(nm), C.byref(na), _p(a_red), _p(u_x), _p(u_p))
        return u_x, u_p[:, :n_act]

    def check_guess(self, evec):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        n, m = (C.c_int(v) for v in evec.shape)
        self._check_guess(C.byref(n), C.byref(m), _p(evec))
        return evec

    def davidson(self, n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd, evec, verbose=False):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        self.lib.ref_davidson(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd,
                              _p(eig), _p(evec), C.byref(ok))
        return eig, evec, bool(ok.value)

    def gen_davidson(self, n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd, bvec, evec, verbose=False):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        self.lib.ref_gen_david(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, shift, matvec, precnd, bvec,
                               _p(eig), _p(evec), C.byref(ok))
        return eig, evec, bool(ok.value)

    def caslr(self, n, n_targ, n_max, max_iter, tol, max_dav, apb, amb, spd, smd, lrprec, evec, verbose=False):
        """reference caslr_driver (diaglib.f90:558-1022, i_alg = 0); evec is 2n x n_max"""
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        self.lib.ref_caslr.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int] + [C.c_void_p] * 5 + [c_dp, c_dp, c_ip]
        self.lib.ref_caslr(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, apb, amb, spd, smd, lrprec,
                           _p(eig), _p(evec), C.byref(ok))
        return eig, evec, bool(ok.value)

    def caslr_eff(self, n, n_targ, n_max, max_iter, tol, max_dav, apb, amb, spd, smd, lrprec, evec, verbose=False):
        """reference caslr_eff_driver (diaglib.f90:1024-1481); evec is 2n x n_max"""
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        assert evec.shape == (2 * n, n_max)
        eig = np.zeros(n_max); ok = C.c_int(0)
        self.lib.ref_caslr_eff.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int] + [C.c_void_p] * 5 + [c_dp, c_dp, c_ip]
        self.lib.ref_caslr_eff(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, apb, amb, spd, smd, lrprec,
                               _p(eig), _p(evec), C.byref(ok))
        return eig, evec, bool(ok.value)

    def lobpcg(self, n, n_targ, n_max, max_iter, tol, shift, matvec, precnd, evec, verbose=False, bvec=None):
        evec = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
        eig = np.zeros(n_max); ok = C.c_int(0)
        self.lib.ref_lobpcg(int(verbose), 0 if bvec is None else 1, n, n_targ, n_max, max_iter, tol, shift, matvec, precnd,
                            bvec if bvec is not None else matvec, _p(eig), _p(evec), C.byref(ok))
        return eig, evec, bool(ok.value)

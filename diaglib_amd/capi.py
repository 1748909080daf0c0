"""ctypes binding of the C-ABI in ``include/diaglib_amd.h``.

This is plumbing for tests and ``bench.py``: it loads ``diaglib_amd/lib/libdiaglib_amd.so`` (HIP
engine + Fortran drivers) and mirrors the reference's interface -- ``davidson_driver`` /
``lobpcg_driver`` with ``matvec(n,m,x,ax)`` / ``precnd(n,m,fac,x,px)`` callbacks (reference
diaglib.f90:1483-1539, 171-228, README.md:34-35).  There is no CPU path: if the library or a
GPU is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Union

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libdiaglib_amd.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
MATVEC_T = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp)
PRECND_T = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp, c_dp)
ALLREDUCE_T = C.CFUNCTYPE(None, C.c_void_p, c_dp, C.c_int, C.c_int)

OPT_CALLBACKS_ON_DEVICE, OPT_EVEC_ON_DEVICE, OPT_PROFILE, OPT_VERBOSE_ORTHO, OPT_CALLBACK_ORDER, OPT_ORTHO_MAXIT, OPT_CASLR_ALGORITHM, OPT_STAGE_CHUNKS = 1, 2, 3, 4, 5, 6, 7, 8
OPT_P2P_TIMEOUT_MS = 9
OPT_RUN_AHEAD = 10
OPT_PENDING_BLOCKS = 11
ERR_NO_DEVICE, ERR_ALLOC, ERR_ARG, ERR_RUNTIME, ERR_ORTHO, ERR_LAPACK, ERR_COMM = 1, 2, 3, 4, 5, 6, 7
OP_NAMES = ["gram", "gemm", "trmm", "ritz", "elem", "matvec", "precnd"]

# every symbol include/diaglib_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "dla_create", "dla_destroy", "dla_default_ctx", "dla_set_option", "dla_get_option", "dla_begin_solve", "dla_class_times", "dla_last_error",
    "dla_backend_name", "dla_get_stats", "dla_reset_stats", "dla_get_kernel_stats", "dla_stream",
    "dla_comm_unique_id", "dla_comm_init", "dla_comm_finalize", "dla_comm_info", "dla_p2p_export", "dla_p2p_attach", "dla_p2p_detach", "dla_set_allreduce_hook", "dla_set_shard",
    "dla_alloc", "dla_free", "dla_trim", "dla_zero", "dla_upload", "dla_download", "dla_copy", "dla_sync",
    "dla_gram", "dla_gram_lower", "dla_panel_gemm", "dla_panel_update", "dla_trmm_linvt", "dla_trmm_gram", "dla_update_gram", "dla_combo_gram", "dla_ritz_residual", "dla_ritz_residual_p", "dla_ritz_residual2", "dla_axpy",
    "dla_nrm2", "dla_stream_triad", "dla_random_fill", "dla_fill_guess",
    "dla_ortho_cd", "dla_ortho_qr", "dla_ortho_vs_x", "dla_b_ortho", "dla_b_ortho_vs_x", "dla_check_guess", "dla_get_coeffs",
    "dla_call_matvec", "dla_call_precnd", "dla_expand_project", "dla_expand_project_metric",
    "dla_syev", "dla_syev_lowest", "dla_potrf_lower", "dla_trtri_lower", "dla_norm_est",
    "dla_synth_setup", "dla_synth_matvec", "dla_synth_precnd", "dla_synth_apbmul", "dla_synth_ambmul", "dla_synth_spdmul", "dla_synth_smdmul",
    "dla_synth_metric", "dla_synth_lrprec1", "dla_synth_lrprec2", "dla_pending_factor", "dla_pending_block", "dla_basis_admit", "dla_basis_fold", "dla_basis_sync", "dla_spmm_setup_csr", "dla_spmm_setup_csr_sharded", "dla_spmm_matvec", "dla_spmm_precnd",
    "dla_davidson_driver", "dla_gen_david_driver", "dla_lobpcg_driver", "dla_caslr_eff_driver", "dla_caslr_driver", "dla_call_lrprec",
    "dla_last_solve_info", "dla_set_solve_info",
]


class Stats(C.Structure):
    _fields_ = [("launches", C.c_longlong * 7), ("alg_bytes", C.c_double * 7), ("flops", C.c_double * 7),
                ("ms", C.c_double * 7), ("allreduces", C.c_longlong), ("host_syncs", C.c_longlong), ("ref_flops", C.c_double)]

    def as_dict(self) -> dict:
        d = {"allreduces": int(self.allreduces), "host_syncs": int(self.host_syncs), "ref_flops": float(self.ref_flops)}
        for i, nm in enumerate(OP_NAMES):
            d[nm] = {"launches": int(self.launches[i]), "alg_bytes": float(self.alg_bytes[i]),
                     "flops": float(self.flops[i]), "ms": float(self.ms[i])}
        return d


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_longlong), ("alg_bytes", C.c_double), ("ms", C.c_double),
                ("flops", C.c_double)]


class DlaError(RuntimeError):
    pass


_lib = None


def load(path: str = LIB_PATH) -> C.CDLL:
    """Load the native library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise DlaError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    # The PyTorch wheel ships its own libamdhip64.so (soname libamdhip64.so.7, like /opt/rocm's).  Loaded first, it also
    # serves this library's DT_NEEDED entry, so the process has ONE HIP runtime; loaded second, the process ends up
    # with two runtimes that share neither devices, streams nor events ("No HIP GPUs are available").  A Python
    # process that uses both must therefore import torch first -- do it here, once, when torch is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path, mode=C.RTLD_LOCAL)
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    sig = {
        "dla_create": (i, [C.POINTER(vp), i]), "dla_destroy": (i, [vp]), "dla_default_ctx": (vp, []),
        "dla_trim": (i, [vp, C.POINTER(sz)]),
        "dla_set_option": (i, [vp, i, i]), "dla_get_option": (i, [vp, i]), "dla_begin_solve": (i, [vp]), "dla_class_times": (i, [vp, vp]),
        "dla_last_error": (C.c_char_p, [vp]), "dla_backend_name": (C.c_char_p, [vp]),
        "dla_get_stats": (i, [vp, C.POINTER(Stats)]), "dla_reset_stats": (i, [vp]), "dla_stream": (vp, [vp]),
        "dla_get_kernel_stats": (i, [vp, C.POINTER(KernelStat), i]),
        "dla_comm_unique_id": (i, [C.c_char_p]), "dla_comm_init": (i, [vp, i, i, C.c_char_p]),
        "dla_comm_info": (i, [vp, c_ip, c_ip]), "dla_comm_finalize": (i, [vp]),
        "dla_p2p_export": (i, [vp, i, C.c_char_p]), "dla_p2p_attach": (i, [vp, i, i, C.c_char_p]), "dla_p2p_detach": (i, [vp]),
        "dla_set_allreduce_hook": (i, [vp, vp, vp, i, i]), "dla_set_shard": (i, [vp, C.c_longlong, C.c_longlong]),
        "dla_alloc": (i, [vp, sz, C.POINTER(vp)]), "dla_free": (i, [vp, vp]), "dla_zero": (i, [vp, vp, sz]),
        "dla_upload": (i, [vp, vp, vp, sz]), "dla_download": (i, [vp, vp, vp, sz]), "dla_copy": (i, [vp, vp, vp, sz]),
        "dla_sync": (i, [vp]),
        "dla_gram": (i, [vp, i, i, vp, i, vp, c_dp, i]),
        "dla_gram_lower": (i, [vp, i, i, vp, vp, c_dp, i]),
        "dla_panel_gemm": (i, [vp, i, i, vp, i, c_dp, i, vp]),
        "dla_panel_update": (i, [vp, i, i, vp, i, c_dp, i, vp]),
        "dla_trmm_linvt": (i, [vp, i, i, vp, c_dp, i]),
        "dla_trmm_gram": (i, [vp, i, i, vp, c_dp, i, c_dp, i]),
        "dla_update_gram": (i, [vp, i, i, vp, i, c_dp, i, vp, c_dp, i]),
        "dla_combo_gram": (i, [vp, i, i, vp, i, c_dp, i, vp, c_dp, i]),
        "dla_ritz_residual": (i, [vp, i, i, i, vp, vp, c_dp, i, c_dp, i, c_ip, vp, vp, vp, c_dp]),
        "dla_ritz_residual_p": (i, [vp, i, i, i, vp, vp, c_dp, i, c_dp, i, c_ip, vp, vp, vp, c_dp, i, c_dp, i, vp, vp]),
        "dla_ritz_residual2": (i, [vp, i, i, i, vp, vp, c_dp, i, c_dp, i, c_dp, i, c_ip, vp, vp, vp, vp, c_dp]),
        "dla_axpy": (i, [vp, sz, d, vp, vp]), "dla_nrm2": (i, [vp, sz, vp, c_dp]),
        "dla_stream_triad": (i, [vp, sz, i, c_dp]),
        "dla_random_fill": (i, [vp, i, i, vp]), "dla_fill_guess": (i, [vp, i, i, vp, C.c_ulonglong, C.c_longlong]),
        "dla_ortho_cd": (i, [vp, i, i, vp, c_dp, c_ip]), "dla_ortho_qr": (i, [vp, i, i, vp]), "dla_ortho_vs_x": (i, [vp, i, i, i, vp, vp]),
        "dla_b_ortho": (i, [vp, i, i, vp, vp]), "dla_b_ortho_vs_x": (i, [vp, i, i, i, vp, vp, vp]),
        "dla_check_guess": (i, [vp, i, i, vp]),
        "dla_get_coeffs": (i, [vp, i, i, i, i, c_dp, c_dp, c_dp]),
        "dla_expand_project": (i, [vp, i, i, i, i, vp, vp, vp, d, c_dp, i]),
        "dla_expand_project_metric": (i, [vp, i, i, i, i, vp, vp, vp, vp, vp, d, c_dp, i]),
        "dla_call_matvec": (i, [vp, vp, i, i, vp, vp]), "dla_call_precnd": (i, [vp, vp, i, i, d, vp, vp]),
        "dla_syev": (i, [C.c_char, i, c_dp, i, c_dp]), "dla_syev_lowest": (i, [C.c_char, i, c_dp, i, c_dp, i]),
        "dla_potrf_lower": (i, [i, c_dp, i]),
        "dla_trtri_lower": (i, [i, c_dp, i]), "dla_norm_est": (d, [i, c_dp, i]),
        "dla_synth_setup": (i, [vp, C.c_longlong, C.c_longlong, i, i, d]),
        "dla_pending_factor": (i, [vp, i, vp, i]),
        "dla_pending_block": (i, [vp, i, i, vp, i, vp]),
        "dla_basis_admit": (i, [i, i, vp, i, i, vp, vp, vp, i]),
        "dla_basis_fold": (i, [i, i, vp, i, vp, i]),
        "dla_basis_sync": (i, [vp, i, i, vp, i]),
        "dla_spmm_setup_csr": (i, [vp, i, vp, vp, vp]),
        "dla_spmm_setup_csr_sharded": (i, [vp, i, C.c_longlong, C.c_longlong, vp, vp, vp]),
        "dla_davidson_driver": (None, [i, i, i, i, i, d, i, d, vp, vp, vp, vp, c_ip]),
        "dla_lobpcg_driver": (None, [i, i, i, i, i, i, d, d, vp, vp, vp, vp, vp, c_ip]),
        "dla_caslr_eff_driver": (None, [i, i, i, i, i, d, i, vp, vp, vp, vp, vp, vp, vp, c_ip]),
        "dla_caslr_driver": (None, [i, i, i, i, i, d, i, vp, vp, vp, vp, vp, vp, vp, c_ip]),
        "dla_call_lrprec": (i, [vp, vp, i, i, d, vp, vp, vp, vp]),
        "dla_gen_david_driver": (None, [i, i, i, i, i, d, i, d, vp, vp, vp, vp, vp, c_ip]),
        "dla_last_solve_info": (None, [c_ip, c_ip, c_ip]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def fn_address(name: str) -> int:
    """Address of an exported function (e.g. the built-in ``dla_synth_matvec`` callback)."""
    return C.cast(getattr(load(), name), C.c_void_p).value


def _dp(a: np.ndarray):
    return a.ctypes.data_as(c_dp)


class DevPanel:
    """A column-major float64 n x m panel in HBM (ld = n)."""

    def __init__(self, ctx: "Context", n: int, m: int, ptr: Optional[int] = None, owner: bool = True):
        self.ctx, self.n, self.m, self.owner = ctx, int(n), int(m), owner
        if ptr is None:
            p = C.c_void_p()
            ctx._chk(ctx.lib.dla_alloc(ctx.h, max(8, 8 * self.n * self.m), C.byref(p)))
            ptr = p.value
        self.ptr = ptr

    def col(self, j: int, m: Optional[int] = None) -> "DevPanel":
        """View of columns j .. j+m-1 (0-based)."""
        m = self.m - j if m is None else m
        return DevPanel(self.ctx, self.n, m, self.ptr + 8 * self.n * j, owner=False)

    def upload(self, a: np.ndarray) -> "DevPanel":
        a = np.asfortranarray(a, dtype=np.float64)
        assert a.size == self.n * self.m, (a.shape, self.n, self.m)
        self.ctx._chk(self.ctx.lib.dla_upload(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self) -> np.ndarray:
        out = np.empty((self.n, self.m), dtype=np.float64, order="F")
        self.ctx._chk(self.ctx.lib.dla_download(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self) -> "DevPanel":
        self.ctx._chk(self.ctx.lib.dla_zero(self.ctx.h, self.ptr, 8 * self.n * self.m))
        return self

    def free(self) -> None:
        # (a panel that outlives its context must not call into the destroyed engine)
        if self.owner and self.ptr and getattr(self.ctx, "h", None):
            self.ctx.lib.dla_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


Callback = Union[int, Callable]


class Context:
    """The library's default context (the one the Fortran drivers use)."""

    def __init__(self):
        self.lib = load()
        self.h = self.lib.dla_default_ctx()
        if not self.h:
            raise DlaError("no HIP device")
        self._keep = []
        # device-mode Python callbacks: True = the trampolines drain the engine's stream before the call and torch's
        # after it; False = rely on the engine's own ordering (OPT_CALLBACK_ORDER), as a compiled caller would
        self.sync_python_callbacks = True

    def destroy(self) -> None:
        """dla_destroy; panels still alive afterwards no longer call into the engine"""
        if self.h:
            h, self.h = self.h, None
            self.lib.dla_destroy(h)

    # ---- plumbing
    def _chk(self, st: int) -> None:
        if st != 0:
            raise DlaError(f"status {st}: {self.lib.dla_last_error(self.h).decode()}")

    @property
    def backend(self) -> str:
        return self.lib.dla_backend_name(self.h).decode()

    def set_option(self, opt: int, val: int) -> None:
        self._chk(self.lib.dla_set_option(self.h, opt, int(val)))

    def get_option(self, opt: int) -> int:
        return int(self.lib.dla_get_option(self.h, opt))

    def sync(self) -> None:
        self._chk(self.lib.dla_sync(self.h))

    def trim(self) -> int:
        """release the allocator's cached (freed) panels; returns the bytes handed back"""
        rel = C.c_size_t(0)
        self._chk(self.lib.dla_trim(self.h, C.byref(rel)))
        return rel.value

    def stats(self) -> dict:
        s = Stats()
        self._chk(self.lib.dla_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def kernel_stats(self) -> dict:
        """Per kernel: launches, algorithmic bytes, HIP-event ms (with OPT_PROFILE)."""
        buf = (KernelStat * 128)()
        n = self.lib.dla_get_kernel_stats(self.h, buf, 128)
        return {buf[j].name.decode(): {"launches": int(buf[j].launches), "alg_bytes": float(buf[j].alg_bytes),
                                       "ms": float(buf[j].ms), "flops": float(buf[j].flops)} for j in range(n)}

    def reset_stats(self) -> None:
        self._chk(self.lib.dla_reset_stats(self.h))

    def panel(self, a_or_n, m: Optional[int] = None) -> DevPanel:
        if isinstance(a_or_n, np.ndarray):
            a = np.asfortranarray(a_or_n, dtype=np.float64)
            if a.ndim == 1:
                a = a.reshape(-1, 1, order="F")
            return DevPanel(self, a.shape[0], a.shape[1]).upload(a)
        return DevPanel(self, a_or_n, m)

    # ---- multi-GPU
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._chk(self.lib.dla_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, uid: bytes) -> None:
        self._chk(self.lib.dla_comm_init(self.h, nranks, rank, uid))

    def p2p_export(self, nranks: int) -> bytes:
        """this rank's mailbox handles (128 bytes) for the one-shot peer-to-peer all-reduce"""
        buf = C.create_string_buffer(128)
        self._chk(self.lib.dla_p2p_export(self.h, nranks, buf))
        return buf.raw

    def p2p_attach(self, nranks: int, rank: int, handles) -> None:
        """handles: the 128-byte exports of all ranks, in rank order"""
        blob = b"".join(handles)
        assert len(blob) == 128 * nranks
        self._chk(self.lib.dla_p2p_attach(self.h, nranks, rank, blob))

    def p2p_detach(self) -> None:
        """close the mailboxes; the small products go back to the RCCL communicator / the hook"""
        self._chk(self.lib.dla_p2p_detach(self.h))

    def comm_finalize(self) -> None:
        self._chk(self.lib.dla_comm_finalize(self.h))

    def comm_info(self):
        """(ranks, this rank) of the attached reduction transport; (1, 0) without one"""
        nr, rk = C.c_int(0), C.c_int(0)
        self._chk(self.lib.dla_comm_info(self.h, C.byref(nr), C.byref(rk)))
        return int(nr.value), int(rk.value)

    def set_shard(self, n_global: int, row0: int) -> None:
        self._chk(self.lib.dla_set_shard(self.h, n_global, row0))

    def set_allreduce_hook(self, fn: Optional[Callable], nranks: int, rank: int) -> None:
        """fn(buf: np.ndarray, op: int) reduces buf in place over ranks (op 0 sum, 1 max)."""
        if fn is None:
            self._chk(self.lib.dla_set_allreduce_hook(self.h, None, None, 1, 0))
            return

        def tramp(_user, buf, count, op):
            fn(np.ctypeslib.as_array(buf, (count,)), op)

        cb = ALLREDUCE_T(tramp)
        self._keep.append(cb)
        self._chk(self.lib.dla_set_allreduce_hook(self.h, C.cast(cb, C.c_void_p), None, nranks, rank))

    # ---- block algebra
    def gram(self, x: DevPanel, u: DevPanel) -> np.ndarray:
        c = np.zeros((x.m, u.m), order="F")
        self._chk(self.lib.dla_gram(self.h, x.n, x.m, x.ptr, u.m, u.ptr, _dp(c), max(1, x.m)))
        return c

    def gram_lower(self, x: DevPanel, u: DevPanel) -> np.ndarray:
        c = np.zeros((x.m, x.m), order="F")
        self._chk(self.lib.dla_gram_lower(self.h, x.n, x.m, x.ptr, u.ptr, _dp(c), max(1, x.m)))
        return c

    def panel_gemm(self, x: DevPanel, c: np.ndarray, z: DevPanel) -> None:
        c = np.asfortranarray(c, dtype=np.float64)
        self._chk(self.lib.dla_panel_gemm(self.h, x.n, x.m, x.ptr, c.shape[1], _dp(c), max(1, c.shape[0]), z.ptr))

    def panel_update(self, x: DevPanel, c: np.ndarray, u: DevPanel) -> None:
        c = np.asfortranarray(c, dtype=np.float64)
        self._chk(self.lib.dla_panel_update(self.h, x.n, x.m, x.ptr, c.shape[1], _dp(c), max(1, c.shape[0]), u.ptr))

    def trmm_linvt(self, u: DevPanel, linv: np.ndarray) -> None:
        linv = np.asfortranarray(linv, dtype=np.float64)
        self._chk(self.lib.dla_trmm_linvt(self.h, u.n, u.m, u.ptr, _dp(linv), linv.shape[0]))

    def trmm_gram(self, u: DevPanel, w: np.ndarray) -> np.ndarray:
        w = np.asfortranarray(w, dtype=np.float64); g = np.zeros((u.m, u.m), order="F")
        self._chk(self.lib.dla_trmm_gram(self.h, u.n, u.m, u.ptr, _dp(w), w.shape[0], _dp(g), u.m))
        return g

    def update_gram(self, x: DevPanel, c: np.ndarray, u: DevPanel) -> np.ndarray:
        c = np.asfortranarray(c, dtype=np.float64); g = np.zeros((u.m, u.m), order="F")
        self._chk(self.lib.dla_update_gram(self.h, x.n, x.m, x.ptr, u.m, _dp(c), c.shape[0], u.ptr, _dp(g), u.m))
        return g

    def combo_gram(self, x: DevPanel, c: np.ndarray, u: DevPanel) -> np.ndarray:
        c = np.asfortranarray(c, dtype=np.float64); g = np.zeros((u.m, u.m), order="F")
        self._chk(self.lib.dla_combo_gram(self.h, x.n, x.m, x.ptr, u.m, _dp(c), c.shape[0], u.ptr, _dp(g), u.m))
        return g

    def ritz_residual(self, v: DevPanel, av: DevPanel, y: np.ndarray, eig: np.ndarray, n_res: int,
                      skip: np.ndarray, evec: DevPanel, r: DevPanel, avy: Optional[DevPanel] = None) -> np.ndarray:
        y = np.asfortranarray(y, dtype=np.float64)
        eig = np.ascontiguousarray(eig, dtype=np.float64)
        skip = np.ascontiguousarray(skip, dtype=np.int32)
        rn = np.zeros((2, max(1, n_res)), order="F")
        self._chk(self.lib.dla_ritz_residual(self.h, v.n, v.m, evec.m, v.ptr, av.ptr, _dp(y), y.shape[0], _dp(eig),
                                             n_res, skip.ctypes.data_as(c_ip), evec.ptr, r.ptr,
                                             avy.ptr if avy is not None else None, _dp(rn)))
        return rn

    def ritz_residual_p(self, v: DevPanel, av: DevPanel, y: np.ndarray, eig: np.ndarray, n_res: int, skip: np.ndarray,
                        evec: DevPanel, r: DevPanel, avy: Optional[DevPanel], c2: np.ndarray, p: DevPanel,
                        ap: DevPanel) -> np.ndarray:
        """the Ritz sweep with the extra products p = V c2, ap = AV c2 formed from the same pass over V and AV"""
        y = np.asfortranarray(y, dtype=np.float64); c2 = np.asfortranarray(c2, dtype=np.float64)
        eig = np.ascontiguousarray(eig, dtype=np.float64)
        skip = np.ascontiguousarray(skip, dtype=np.int32)
        rn = np.zeros((2, max(1, n_res)), order="F")
        self._chk(self.lib.dla_ritz_residual_p(self.h, v.n, v.m, evec.m, v.ptr, av.ptr, _dp(y), y.shape[0], _dp(eig),
                                               n_res, skip.ctypes.data_as(c_ip), evec.ptr, r.ptr,
                                               avy.ptr if avy is not None else None, _dp(rn),
                                               c2.shape[1], _dp(c2), c2.shape[0], p.ptr, ap.ptr))
        return rn

    def ritz_residual2(self, v: DevPanel, av: DevPanel, y1: np.ndarray, y2: np.ndarray, eig: np.ndarray, n_res: int,
                       skip: np.ndarray, e: DevPanel, r: DevPanel) -> np.ndarray:
        """e = V y1, r = AV y2 - eig e with the norms of r, one sweep (the residual blocks of the linear-response drivers)"""
        y1 = np.asfortranarray(y1, dtype=np.float64); y2 = np.asfortranarray(y2, dtype=np.float64)
        eig = np.ascontiguousarray(eig, dtype=np.float64)
        skip = np.ascontiguousarray(skip, dtype=np.int32)
        rn = np.zeros((2, max(1, n_res)), order="F")
        tw, jk = self.panel(v.n, e.m), self.panel(v.n, e.m)
        try:
            self._chk(self.lib.dla_ritz_residual2(self.h, v.n, v.m, e.m, v.ptr, av.ptr, _dp(y1), y1.shape[0], _dp(y2), y2.shape[0],
                                                  _dp(eig), n_res, skip.ctypes.data_as(c_ip), e.ptr, r.ptr, tw.ptr, jk.ptr, _dp(rn)))
        finally:
            tw.free(); jk.free()
        return rn

    def axpy(self, alpha: float, x: DevPanel, y: DevPanel) -> None:
        self._chk(self.lib.dla_axpy(self.h, x.n * x.m, alpha, x.ptr, y.ptr))

    def nrm2(self, x: DevPanel) -> float:
        out = C.c_double(0.0)
        self._chk(self.lib.dla_nrm2(self.h, x.n * x.m, x.ptr, C.byref(out)))
        return out.value

    def stream_triad(self, length: int, reps: int = 5) -> float:
        """device STREAM triad GB/s on three scratch arrays of `length` doubles (best of reps)"""
        out = C.c_double(0.0)
        self._chk(self.lib.dla_stream_triad(self.h, length, reps, C.byref(out)))
        return out.value

    def random_fill(self, x: DevPanel) -> None:
        self._chk(self.lib.dla_random_fill(self.h, x.n, x.m, x.ptr))

    def fill_guess(self, x: DevPanel, seed: int = 2, support_rows: int = 0) -> None:
        """uniform [-0.5, 0.5) guess from the documented counter-based generator (SURVEY 8d guess (b)); with
        support_rows > 0 only the leading support_rows global rows are random, the rest zero"""
        self._chk(self.lib.dla_fill_guess(self.h, x.n, x.m, x.ptr, seed, support_rows))

    # ---- orthogonalisation
    def ortho_cd(self, u: DevPanel):
        g = C.c_double(0.0); ok = C.c_int(0)
        self._chk(self.lib.dla_ortho_cd(self.h, u.n, u.m, u.ptr, C.byref(g), C.byref(ok)))
        return g.value, bool(ok.value)

    def ortho_qr(self, u: DevPanel) -> None:
        """the reference's Householder fallback `ortho` (diaglib.f90:3052-3092): U <- U R^-1, LAPACK signs"""
        self._chk(self.lib.dla_ortho_qr(self.h, u.n, u.m, u.ptr))

    def ortho_vs_x(self, x: DevPanel, u: DevPanel, m: Optional[int] = None) -> None:
        self._chk(self.lib.dla_ortho_vs_x(self.h, x.n, x.m if m is None else m, u.m, x.ptr, u.ptr))

    def pending_factor(self, k: int) -> np.ndarray:
        """the triangular factor the last expand_project(mode 3) left pending (identity when none)"""
        t = np.zeros((k, k), order="F")
        self._chk(self.lib.dla_pending_factor(self.h, k, _dp(t), k))
        return t

    def pending_block(self, m: int, k: int) -> np.ndarray:
        """[E ; T] ((m + k) x k): what the last expand_project(mode 3 / 4) left pending -- the finished block is
        [X | U_stored] p ([0 ; I] when nothing stayed pending)"""
        p = np.zeros((m + k, k), order="F")
        applied = C.c_int(0)
        self._chk(self.lib.dla_pending_block(self.h, m, k, _dp(p), m + k, C.byref(applied)))
        self.pending_applied = bool(applied.value)       # the chain's closing sweep has applied p in memory (see the header)
        return p

    def basis_admit(self, m: int, k: int, p: np.ndarray, hraw: np.ndarray, dmat: np.ndarray, h: np.ndarray, applied: bool = False) -> None:
        """dla_basis_admit on square Fortran-ordered arrays of one leading dimension (in place; p is completed in place)"""
        ld = h.shape[0]
        assert hraw.shape == dmat.shape == h.shape and all(a.flags.f_contiguous for a in (p, hraw, dmat, h))
        self._chk(self.lib.dla_basis_admit(m, k, _dp(p), p.shape[0], 1 if applied else 0, _dp(hraw), _dp(dmat), _dp(h), ld))

    def basis_fold(self, rows: int, dmat: np.ndarray, c: np.ndarray) -> None:
        """c[:rows] <- dmat[:rows, :rows] c[:rows] (in place)"""
        assert dmat.flags.f_contiguous and c.flags.f_contiguous
        self._chk(self.lib.dla_basis_fold(rows, c.shape[1], _dp(dmat), dmat.shape[0], _dp(c), c.shape[0]))

    def basis_sync(self, m: int, k: int, dmat: np.ndarray = None) -> None:
        """dla_basis_sync: columns m .. m+k-1 of the caller's D to the device (k <= 0: forget everything)"""
        if k <= 0:
            self._chk(self.lib.dla_basis_sync(self.h, 0, 0, None, 0))
            return
        assert dmat.flags.f_contiguous
        self._chk(self.lib.dla_basis_sync(self.h, m, k, _dp(dmat), dmat.shape[0]))

    def expand_project(self, mode: int, basis: DevPanel, abasis: DevPanel, m: int, k: int, matvec: int, shift: float = 0.0) -> np.ndarray:
        """dla_expand_project on the leading m + k columns of the two panels: ortho_vs_x(X, U), AU = A U + shift U, then the
        projection -- mode 0: [X | U]^T AU ((m+k) x k), mode 1: lower triangle of [X | U]^T [AX | AU]"""
        h = np.zeros((m + k, k if mode in (0, 4, 5, 6) else m + k), order="F")    # (modes 3 / 4 / 5: 1 / 0 / 0 with the last factor pending; 6: 0 against the finished basis X D)
        self._chk(self.lib.dla_expand_project(self.h, mode, basis.n, m, k, basis.ptr, abasis.ptr, matvec, shift, _dp(h), m + k))
        return h

    def expand_project_metric(self, mode: int, basis: DevPanel, bbasis: DevPanel, abasis: DevPanel, m: int, k: int, matvec: int,
                              bvec: int, shift: float = 0.0) -> np.ndarray:
        """dla_expand_project_metric on the leading m + k columns of the three panels"""
        h = np.zeros((m + k, k if mode == 0 else m + k), order="F")
        self._chk(self.lib.dla_expand_project_metric(self.h, mode, basis.n, m, k, basis.ptr, bbasis.ptr, abasis.ptr, matvec, bvec, shift,
                                                     _dp(h), m + k))
        return h

    def b_ortho(self, u: DevPanel, bu: DevPanel) -> None:
        self._chk(self.lib.dla_b_ortho(self.h, u.n, u.m, u.ptr, bu.ptr))

    def b_ortho_vs_x(self, x: DevPanel, bx: DevPanel, u: DevPanel) -> None:
        self._chk(self.lib.dla_b_ortho_vs_x(self.h, x.n, x.m, u.m, x.ptr, bx.ptr, u.ptr))

    def check_guess(self, evec: DevPanel) -> None:
        self._chk(self.lib.dla_check_guess(self.h, evec.n, evec.m, evec.ptr))

    def get_coeffs(self, a_red: np.ndarray, len_u: int, n_max: int, n_act: int):
        a_red = np.asfortranarray(a_red, dtype=np.float64)
        u_x = np.zeros((len_u, n_max), order="F"); u_p = np.zeros((len_u, max(1, n_act)), order="F")
        self._chk(self.lib.dla_get_coeffs(self.h, a_red.shape[0], len_u, n_max, n_act, _dp(a_red), _dp(u_x), _dp(u_p)))
        return u_x, u_p[:, :n_act]

    # ---- built-in operator
    def synth_setup(self, n_global: int, row0: int, n_local: int, rank_w: int = 4, sigma: float = 0.5) -> None:
        self._chk(self.lib.dla_synth_setup(self.h, n_global, row0, n_local, rank_w, sigma))

    def spmm_setup(self, a) -> None:
        """hand a scipy.sparse matrix (symmetric, square) to the sample ELLPACK operator of this thread's context"""
        a = a.tocsr()
        rp = np.ascontiguousarray(a.indptr, dtype=np.int64)
        ci = np.ascontiguousarray(a.indices, dtype=np.int32)
        va = np.ascontiguousarray(a.data, dtype=np.float64)
        self._chk(self.lib.dla_spmm_setup_csr(self.h, a.shape[0], rp.ctypes.data, ci.ctypes.data, va.ctypes.data))

    def spmm_setup_sharded(self, a_rows, row0: int, n_global: int) -> None:
        """hand THIS rank's rows (a scipy.sparse matrix of shape n_local x n_global, global column indices) of a banded
        symmetric matrix to the sample operator; collective over the ranks of the context's transport"""
        a = a_rows.tocsr()
        rp = np.ascontiguousarray(a.indptr, dtype=np.int64)
        ci = np.ascontiguousarray(a.indices, dtype=np.int64)
        va = np.ascontiguousarray(a.data, dtype=np.float64)
        self._chk(self.lib.dla_spmm_setup_csr_sharded(self.h, a.shape[0], row0, n_global, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))

    def synth_matvec(self, x: DevPanel, ax: DevPanel) -> None:
        self._chk(self.lib.dla_call_matvec(self.h, fn_address("dla_synth_matvec"), x.n, x.m, x.ptr, ax.ptr))

    # ---- callbacks
    # A Python callable receives numpy views of the HOST blocks by default.  With OPT_CALLBACKS_ON_DEVICE = 1 it
    # receives torch tensors that alias the DEVICE blocks (zero copy, n x m, column-major strides), so an operator
    # written with torch (dense, torch.sparse, custom kernels) stays in HBM.  The engine's stream is drained before
    # the call and torch's after it; the result is written into the output block.
    def _device_callbacks(self) -> bool:
        return self.lib.dla_get_option(self.h, OPT_CALLBACKS_ON_DEVICE) == 1

    @staticmethod
    def _dev_tensor(ptr, n: int, m: int):
        import torch

        class _Block:                      # column-major n x m block as an (m, n) row-major array
            __cuda_array_interface__ = {"shape": (m, n), "typestr": "<f8", "data": (C.cast(ptr, C.c_void_p).value, False),
                                        "version": 3, "strides": None}
        return torch.as_tensor(_Block(), device="cuda").T

    def _wrap_mv(self, f: Callback) -> int:
        if isinstance(f, int):
            return f
        if self._device_callbacks():
            import torch

            def tramp(pn, pm, px, pax):
                n, m = pn[0], pm[0]
                if self.sync_python_callbacks:
                    self.sync()
                self._dev_tensor(pax, n, m).copy_(f(self._dev_tensor(px, n, m)))
                if self.sync_python_callbacks:
                    torch.cuda.synchronize()
        else:
            def tramp(pn, pm, px, pax):
                n, m = pn[0], pm[0]
                x = np.ctypeslib.as_array(px, (m, n)).T
                ax = np.ctypeslib.as_array(pax, (m, n)).T
                ax[:, :] = f(x)

        cb = MATVEC_T(tramp)
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p).value

    def _wrap_pc(self, f: Callback) -> int:
        if isinstance(f, int):
            return f
        if self._device_callbacks():
            import torch

            def tramp(pn, pm, pf, px, ppx):
                n, m = pn[0], pm[0]
                if self.sync_python_callbacks:
                    self.sync()
                self._dev_tensor(ppx, n, m).copy_(f(pf[0], self._dev_tensor(px, n, m)))
                if self.sync_python_callbacks:
                    torch.cuda.synchronize()
        else:
            def tramp(pn, pm, pf, px, ppx):
                n, m = pn[0], pm[0]
                x = np.ctypeslib.as_array(px, (m, n)).T
                y = np.ctypeslib.as_array(ppx, (m, n)).T
                y[:, :] = f(pf[0], x)

        cb = PRECND_T(tramp)
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p).value

    # ---- drivers (reference argument order)
    def davidson_driver(self, n: int, n_targ: int, n_max: int, max_iter: int, tol: float, max_dav: int, shift: float,
                        matvec: Callback, precnd: Callback, evec, verbose: bool = False):
        """evec: numpy (n, n_max) guess (host mode) or DevPanel (evec-on-device mode).
        Returns (eig, evec_out, ok, info)."""
        mv, pc = self._wrap_mv(matvec), self._wrap_pc(precnd)
        eig = np.zeros(n_max); ok = C.c_int(0)
        if isinstance(evec, DevPanel):
            self.set_option(OPT_EVEC_ON_DEVICE, 1)
            ev_ptr, out = evec.ptr, evec
        else:
            self.set_option(OPT_EVEC_ON_DEVICE, 0)
            out = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
            ev_ptr = out.ctypes.data
        self.lib.dla_davidson_driver(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, shift, mv, pc,
                                     eig.ctypes.data, ev_ptr, C.byref(ok))
        return eig, out, bool(ok.value), self.last_solve_info()

    def gen_david_driver(self, n: int, n_targ: int, n_max: int, max_iter: int, tol: float, max_dav: int, shift: float,
                         matvec: Callback, precnd: Callback, bvec: Callback, evec, verbose: bool = False):
        mv, pc, bv = self._wrap_mv(matvec), self._wrap_pc(precnd), self._wrap_mv(bvec)
        eig = np.zeros(n_max); ok = C.c_int(0)
        if isinstance(evec, DevPanel):
            self.set_option(OPT_EVEC_ON_DEVICE, 1)
            ev_ptr, out = evec.ptr, evec
        else:
            self.set_option(OPT_EVEC_ON_DEVICE, 0)
            out = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
            ev_ptr = out.ctypes.data
        self.lib.dla_gen_david_driver(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, shift, mv, pc, bv,
                                      eig.ctypes.data, ev_ptr, C.byref(ok))
        return eig, out, bool(ok.value), self.last_solve_info()

    def lobpcg_driver(self, n: int, n_targ: int, n_max: int, max_iter: int, tol: float, shift: float,
                      matvec: Callback, precnd: Callback, evec, verbose: bool = False, bvec: Optional[Callback] = None):
        mv, pc = self._wrap_mv(matvec), self._wrap_pc(precnd)
        bv = self._wrap_mv(bvec) if bvec is not None else mv
        eig = np.zeros(n_max); ok = C.c_int(0)
        if isinstance(evec, DevPanel):
            self.set_option(OPT_EVEC_ON_DEVICE, 1)
            ev_ptr, out = evec.ptr, evec
        else:
            self.set_option(OPT_EVEC_ON_DEVICE, 0)
            out = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
            ev_ptr = out.ctypes.data
        self.lib.dla_lobpcg_driver(int(verbose), 0 if bvec is None else 1, n, n_targ, n_max, max_iter, tol, shift, mv, pc, bv,
                                   eig.ctypes.data, ev_ptr, C.byref(ok))
        return eig, out, bool(ok.value), self.last_solve_info()

    def caslr_driver(self, *args, **kw):
        """traditional linear-response driver (reference diaglib.f90:558-1022); arguments as caslr_eff_driver"""
        return self.caslr_eff_driver(*args, _entry="dla_caslr_driver", **kw)

    def caslr_eff_driver(self, n: int, n_targ: int, n_max: int, max_iter: int, tol: float, max_dav: int,
                         apbmul: Callback, ambmul: Callback, spdmul: Callback, smdmul: Callback, lrprec: Callback,
                         evec, verbose: bool = False, _entry: str = "dla_caslr_eff_driver"):
        """linear-response driver (reference diaglib.f90:1024-1481); evec is 2n x n_max; lrprec as an address or a
        Python callable (fac, xp, xm) -> (yp, ym)"""
        f1, f2, f3, f4 = (self._wrap_mv(f) for f in (apbmul, ambmul, spdmul, smdmul))
        f5 = self._wrap_lrpc(lrprec)
        eig = np.zeros(n_max); ok = C.c_int(0)
        if isinstance(evec, DevPanel):
            self.set_option(OPT_EVEC_ON_DEVICE, 1)
            ev_ptr, out = evec.ptr, evec
        else:
            self.set_option(OPT_EVEC_ON_DEVICE, 0)
            out = np.asfortranarray(evec, dtype=np.float64).copy(order="F")
            assert out.shape == (2 * n, n_max)
            ev_ptr = out.ctypes.data
        getattr(self.lib, _entry)(int(verbose), n, n_targ, n_max, max_iter, tol, max_dav, f1, f2, f3, f4, f5,
                                  eig.ctypes.data, ev_ptr, C.byref(ok))
        return eig, out, bool(ok.value), self.last_solve_info()

    def _wrap_lrpc(self, f: Callback) -> int:
        if isinstance(f, int):
            return f

        def tramp(pn, pm, pf, pxp, pxm, pyp, pym):
            n, m = pn[0], pm[0]
            xp = np.ctypeslib.as_array(pxp, (m, n)).T
            xm = np.ctypeslib.as_array(pxm, (m, n)).T
            yp, ym = f(pf[0], xp, xm)
            np.ctypeslib.as_array(pyp, (m, n)).T[:, :] = yp
            np.ctypeslib.as_array(pym, (m, n)).T[:, :] = ym

        cb = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp, c_dp, c_dp, c_dp)(tramp)
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p).value

    def last_solve_info(self) -> dict:
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self.lib.dla_last_solve_info(C.byref(a), C.byref(b), C.byref(c))
        return {"iters": a.value, "matvec_cols": b.value, "restarts": c.value}


# ---- host-size dense helpers (no GPU needed)
def syev(a: np.ndarray, uplo: str = "l"):
    L = load()
    a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
    n = a.shape[0]
    w = np.zeros(n)
    info = L.dla_syev(uplo.encode(), n, _dp(a), n, _dp(w))
    if info != 0:
        raise DlaError(f"dla_syev info={info}")
    return w, a


def syev_lowest(a: np.ndarray, m: int, uplo: str = "l"):
    L = load()
    a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
    n = a.shape[0]
    w = np.zeros(n)
    info = L.dla_syev_lowest(uplo.encode(), n, _dp(a), n, _dp(w), m)
    if info != 0:
        raise DlaError(f"dla_syev_lowest info={info}")
    return w, a[:, :m]


def potrf_lower(a: np.ndarray):
    L = load()
    a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
    info = L.dla_potrf_lower(a.shape[0], _dp(a), a.shape[0])
    return a, info


def trtri_lower(a: np.ndarray):
    L = load()
    a = np.asfortranarray(a, dtype=np.float64).copy(order="F")
    info = L.dla_trtri_lower(a.shape[0], _dp(a), a.shape[0])
    return a, info


def norm_est(a: np.ndarray) -> float:
    L = load()
    a = np.asfortranarray(a, dtype=np.float64)
    return float(L.dla_norm_est(a.shape[0], _dp(a), a.shape[0]))

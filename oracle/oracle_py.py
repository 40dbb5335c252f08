"""oracle/oracle_py.py — ctypes door onto oracle/libsvnicp_oracle.so and oracle/_ref.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsvnicp_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libknn_cpu_ref.so")

OPTIMIZERS = {"Adam": 0, "RMSprop": 1, "SGD": 2, "Adagrad": 3}
MODE_SVN, MODE_SVGD = 0, 1


class Params(C.Structure):
    _fields_ = [("iterations", C.c_int), ("lr", C.c_double), ("max_dist", C.c_double),
                ("check_early_stop", C.c_int), ("convergence_threshold", C.c_double),
                ("knn_count", C.c_int), ("svn_full_grad", C.c_int), ("optimizer", C.c_int)]


class Trace(C.Structure):
    _fields_ = [("corr", C.c_void_p), ("mask", C.c_void_p), ("H", C.c_void_p), ("b", C.c_void_p),
                ("newton", C.c_void_p), ("phi", C.c_void_p), ("h", C.c_void_p), ("pose", C.c_void_p)]


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "svnicp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libsvnicp_oracle.so"])
    if os.path.isdir("/root/reference") and not os.path.exists(_REF):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        dp, ip64, fp = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_float)
        L.orc_knn_topk.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_int, ip64, dp]
        L.orc_knn_topk_f32.argtypes = [fp, C.c_int64, fp, C.c_int64, C.c_int, ip64, fp]
        L.orc_transform.argtypes = [dp, C.c_int64, dp, dp, dp]
        L.orc_so3_exp.argtypes = [dp, dp, dp]
        L.orc_so3_log.argtypes = [dp, dp]
        L.orc_euler_to_R.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.orc_solve6.argtypes = [dp, dp, dp]
        L.orc_inv6.argtypes = [dp, dp]
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.POINTER(Params), dp, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_add_cloud.argtypes = [C.c_void_p, dp, C.c_int64, dp, C.c_int64, dp, C.c_int]
        L.orc_set_initial_mean.argtypes = [C.c_void_p, dp, dp]
        L.orc_set_k.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_threshold.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_trace.argtypes = [C.c_void_p, C.POINTER(Trace)]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_threads.restype = C.c_int
        L.orc_stein_align.argtypes = [C.c_void_p]
        L.orc_stein_align.restype = C.c_int
        for n in ("transformation", "distribution", "cov_matrix", "particles", "particle_weight"):
            getattr(L, "orc_get_" + n).argtypes = [C.c_void_p, dp]
        L.orc_get_particle_history.argtypes = [C.c_void_p, fp]
        L.orc_get_finish_iter.argtypes = [C.c_void_p]
        L.orc_get_finish_iter.restype = C.c_int
        L.orc_get_iterations_run.argtypes = [C.c_void_p]
        L.orc_get_iterations_run.restype = C.c_int
        L.orc_set_correspondence_full.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_correspondence_full.restype = None
        L.orc_get_candidates.argtypes = [C.c_void_p]
        L.orc_get_candidates.restype = ip64
        L.orc_get_candidate_dist2.argtypes = [C.c_void_p]
        L.orc_get_candidate_dist2.restype = dp
        L.orc_sp_begin.argtypes = [C.c_void_p]
        L.orc_sp_candidate_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.orc_sp_candidates.argtypes = [C.c_void_p]
        L.orc_sp_candidates.restype = ip64
        L.orc_sp_build_table.argtypes = [C.c_void_p]
        L.orc_sp_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]
        L.orc_sp_accumulate_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, C.c_int]
        L.orc_sp_update.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_sp_update.restype = C.c_int
        L.orc_sp_finish.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def set_threads(n: int):
    lib().orc_set_threads(int(n))


def get_threads() -> int:
    return int(lib().orc_get_threads())


def knn_topk(q, tgt, K):
    q, tgt = _d(q), _d(tgt)
    B, M = q.shape[0], tgt.shape[0]
    idx = np.zeros((B, K), np.int64)
    d2 = np.zeros((B, K), np.float64)
    lib().orc_knn_topk(_p(q), B, _p(tgt), M, K, _p(idx, C.c_int64), _p(d2))
    return idx, d2


def knn_topk_f32(q, tgt, K):
    q = np.ascontiguousarray(q, np.float32)
    tgt = np.ascontiguousarray(tgt, np.float32)
    B, M = q.shape[0], tgt.shape[0]
    idx = np.zeros((B, K), np.int64)
    d2 = np.zeros((B, K), np.float32)
    lib().orc_knn_topk_f32(_p(q, C.c_float), B, _p(tgt, C.c_float), M, K, _p(idx, C.c_int64), _p(d2, C.c_float))
    return idx, d2


def ref_available() -> bool:
    return os.path.exists(_REF)


def ref_knn_cpu_f32(q, tgt, K):
    """The reference's own KNearestNeighborIdxCpu (float32) via oracle/_ref."""
    import torch  # noqa: F401  (libtorch must be loaded before the harness)
    R = C.CDLL(_REF)
    q = np.ascontiguousarray(q, np.float32)
    tgt = np.ascontiguousarray(tgt, np.float32)
    B, M = q.shape[0], tgt.shape[0]
    idx = np.zeros((B, K), np.int64)
    d2 = np.zeros((B, K), np.float32)
    R.ref_knn_cpu_f32.argtypes = [C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float), C.c_int64, C.c_int,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    rc = R.ref_knn_cpu_f32(_p(q, C.c_float), B, _p(tgt, C.c_float), M, K, _p(idx, C.c_int64), _p(d2, C.c_float))
    if rc != 0:
        raise RuntimeError("reference KNearestNeighborIdxCpu raised")
    return idx, d2


def transform(src, R, t):
    src = _d(src)
    out = np.empty_like(src)
    lib().orc_transform(_p(src), src.shape[0], _p(_d(R).reshape(9)), _p(_d(t).reshape(3)), _p(out))
    return out


def so3_exp(r):
    R, J = np.zeros(9), np.zeros(9)
    lib().orc_so3_exp(_p(_d(r)), _p(R), _p(J))
    return R.reshape(3, 3), J.reshape(3, 3)


def so3_log(R):
    w = np.zeros(3)
    lib().orc_so3_log(_p(_d(R).reshape(9)), _p(w))
    return w


def solve6(A, b):
    x = np.zeros(6)
    lib().orc_solve6(_p(_d(A).reshape(36)), _p(_d(b)), _p(x))
    return x


def inv6(A):
    x = np.zeros(36)
    lib().orc_inv6(_p(_d(A).reshape(36)), _p(x))
    return x.reshape(6, 6)


class Solver:
    """Mirrors svnicp::SVNICP / svnicp::SVGDICP on top of the C oracle."""

    def __init__(self, init_pose, mode=MODE_SVN, iterations=50, lr=0.02, max_dist=1.0, check_early_stop=False,
                 convergence_threshold=1e-5, knn_count=100, svn_full_grad=True, optimizer="Adam"):
        self.L = lib()
        init_pose = _d(init_pose).reshape(6, -1)
        self.P = init_pose.shape[1]
        self.I = iterations
        self.prm = Params(iterations, lr, max_dist, int(check_early_stop), convergence_threshold, knn_count,
                          int(svn_full_grad), OPTIMIZERS.get(optimizer, -1))
        self.h = self.L.orc_create(mode, C.byref(self.prm), _p(init_pose), self.P)
        self.K = knn_count
        self._trace_bufs = None

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def add_cloud(self, src, tgt, init_pose):
        src, tgt, init_pose = _d(src), _d(tgt), _d(init_pose).reshape(6, -1)
        self.B, self.M = src.shape[0], tgt.shape[0]
        self.L.orc_add_cloud(self.h, _p(src), self.B, _p(tgt), self.M, _p(init_pose), init_pose.shape[1])

    def set_initial_mean(self, R0, t0):
        self.L.orc_set_initial_mean(self.h, _p(_d(R0).reshape(9)), _p(_d(t0).reshape(3)))

    def set_k(self, k):
        self.K = k
        self.L.orc_set_k(self.h, k)

    def set_threshold(self, md):
        self.L.orc_set_threshold(self.h, md)

    def enable_trace(self):
        I, P, B = self.I, self.P, self.B
        bufs = dict(corr=np.full((I, P, B), -1, np.int32), mask=np.zeros((I, P, B), np.uint8),
                    H=np.zeros((I, P, 36)), b=np.zeros((I, P, 6)), newton=np.zeros((I, P, 6)),
                    phi=np.zeros((I, P, 6)), h=np.zeros(I), pose=np.zeros((I, 6, P)))
        t = Trace(*[bufs[k].ctypes.data for k in ("corr", "mask", "H", "b", "newton", "phi", "h", "pose")])
        self.L.orc_set_trace(self.h, C.byref(t))
        self._trace_bufs = bufs
        return bufs

    def stein_align(self):
        return self.L.orc_stein_align(self.h)

    def _get(self, name, n):
        out = np.zeros(n)
        getattr(self.L, "orc_get_" + name)(self.h, _p(out))
        return out

    def get_transformation(self):
        return self._get("transformation", 6)

    def get_distribution(self):
        return self._get("distribution", 6)

    def get_cov_matrix(self):
        return self._get("cov_matrix", 36)

    def get_particles(self):
        return self._get("particles", 6 * self.P)

    def get_particle_weight(self):
        return self._get("particle_weight", self.P)

    def get_particle_history(self):
        out = np.zeros((self.I, 6 * self.P), np.float32)
        self.L.orc_get_particle_history(self.h, _p(out, C.c_float))
        return out

    def finish_iter(self):
        """finish_iter_ as the reference keeps it: constructor value, changed only by an SVGD early stop."""
        return self.L.orc_get_finish_iter(self.h)

    def iterations_run(self):
        return self.L.orc_get_iterations_run(self.h)

    def set_correspondence_full(self, on=True):
        """SVGDICP.cpp:274-298 get_correspondence (K = 1 over the whole target per particle) instead of the fast path."""
        self.L.orc_set_correspondence_full(self.h, 1 if on else 0)

    def candidates(self):
        p = self.L.orc_get_candidates(self.h)
        return np.ctypeslib.as_array(p, shape=(self.B, self.K)).copy()

    def candidate_dist2(self):
        p = self.L.orc_get_candidate_dist2(self.h)
        return np.ctypeslib.as_array(p, shape=(self.B, self.K)).copy()

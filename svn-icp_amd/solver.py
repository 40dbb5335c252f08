"""Host-side mirror of the reference solver classes on top of the C ABI.

Same names, argument meaning and call sequence as ``svnicp::SVGDICP`` / ``svnicp::SVNICP``
(/root/reference/svn-icp/include/core/SVGDICP.h:64-110, SVNICP.h:29-43) as driven by
``OdometryPipeline::ICP_processing`` (src/core/OdometryPipeline.cpp:573-607):

    solver = SVNICP(param, init_pose, ParticleWeightOpt())
    solver.add_cloud(source, target, init_pose)       # [B,3], [M,3] float64, [6,P]
    solver.set_initial_mean(T_4x4)                    # gtsam::Pose3 in the reference
    state  = solver.stein_align()
    mean   = solver.get_transformation()              # [6]  (x,y,z, so(3) log)
    var    = solver.get_distribution()                # [6]
    cov    = solver.get_cov_matrix()                  # [36] row-major
    parts  = solver.get_particles()                   # [6*P]: x.., y.., z.., rx.., ry.., rz..
    w      = solver.get_particle_weight()             # [P]

Clouds may be numpy arrays (host, copied over PCIe) or float64 CUDA torch tensors (device
pointers are handed to the library, no host round trip).  All compute happens in
libsvnicp_hip.so; this file holds no arithmetic of the path.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import binding
from .binding import Params, SvnIcpError

_OPT = {"Adam": 0, "RMSprop": 1, "SGD": 2, "Adagrad": 3}


class SteinICPState(enum.IntEnum):  # include/core/SVGDICP.h:59-62
    ALIGN_SUCCESS = 1
    NO_OPTIMIZER = 2


@dataclass
class SteinICPParam:  # include/core/SVGDICP.h:41-57 (same field names and defaults)
    iterations: int = 50
    use_minibatch: bool = False      # never set by the reference node; ignored by its solver
    batch_size: int = 50             # overwritten with N_src by the reference (SVGDICP.cpp:181)
    lr: float = 0.02
    max_dist: float = 1.0
    normalize_cloud: bool = True     # normalize_factor_ == 1 in the reference (SVGDICP.cpp:32)
    optimizer: str = "Adam"
    check_early_stop: bool = False
    convergence_steps: int = 5       # unused by the reference solver
    convergence_threshold: float = 1e-5
    KNN_count: int = 100
    SVN_full_grad: bool = True
    record_trace: bool = False       # test hook (not in the reference)


@dataclass
class ParticleWeightOpt:  # include/core/SVNICP.h:25-27
    use_weight_mean: bool = False


def initialize_particles(particle_count: int, ub, lb, rng: np.random.Generator | None = None) -> np.ndarray:
    """svnicp::initialize_particles (src/core/ICPUtils.cpp:45-58): uniform in [lb, ub] per row,
    zeros for a single particle.  Returns [6, P] float64."""
    if particle_count == 1:
        return np.zeros((6, 1))
    rng = rng or np.random.default_rng()
    ub, lb = np.asarray(ub, np.float64).reshape(6, 1), np.asarray(lb, np.float64).reshape(6, 1)
    return (ub - lb) * rng.random((6, particle_count)) + lb


def _is_torch_cuda(x) -> bool:
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


class _SolverBase:
    _mode = 0

    def __init__(self, parameters: SteinICPParam, init_pose, opt: ParticleWeightOpt | None = None, device: int = 0):
        self._L = binding.load_library()
        self._h = C.c_void_p()
        self.config = parameters
        self.weight_config = opt or ParticleWeightOpt()
        init = self._pose_arg(init_pose)
        self._P = init.shape[1]
        prm = Params(C.sizeof(Params), self._mode, int(parameters.iterations), int(parameters.KNN_count),
                     float(parameters.lr), float(parameters.max_dist), float(parameters.convergence_threshold),
                     int(parameters.check_early_stop), int(parameters.SVN_full_grad),
                     _OPT.get(parameters.optimizer, -1), int(parameters.record_trace))
        rc = self._L.svnicp_create(C.byref(prm), int(device), init.ctypes.data_as(C.POINTER(C.c_double)), self._P,
                                   C.byref(self._h))
        if rc != 0:
            raise SvnIcpError(f"svnicp_create failed ({rc}): {self._L.svnicp_last_error(None).decode()}")
        self._B = self._M = 0
        self._K = int(parameters.KNN_count)
        self._keep = None

    # -- plumbing -------------------------------------------------------------------------
    @staticmethod
    def _pose_arg(init_pose) -> np.ndarray:
        if hasattr(init_pose, "detach"):
            init_pose = init_pose.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(init_pose, np.float64))
        if a.ndim == 3:
            a = a.reshape(a.shape[0], a.shape[1])
        if a.ndim != 2 or a.shape[0] != 6:
            raise ValueError("init_pose must be [6, P] (or [6, P, 1])")
        return np.ascontiguousarray(a)

    def _check(self, rc: int, what: str):
        if rc < 0:
            raise SvnIcpError(f"{what} failed ({rc}): {self._L.svnicp_last_error(self._h).decode()}")
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.svnicp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- reference interface ----------------------------------------------------------------
    def add_cloud(self, new_cloud, target, init_pose):
        """SVGDICP::add_cloud (src/core/SVGDICP.cpp:46-62)."""
        if _is_torch_cuda(new_cloud) != _is_torch_cuda(target):
            raise ValueError("source and target must live on the same side (both host or both device)")
        if _is_torch_cuda(new_cloud):
            import torch
            src = new_cloud.to(torch.float64).contiguous()
            tgt = target.to(torch.float64).contiguous()
            torch.cuda.current_stream(src.device).synchronize()
            self._keep = (src, tgt)
            B, M = src.shape[0], tgt.shape[0]
            rc = self._L.svnicp_set_clouds(self._h, C.c_void_p(src.data_ptr()), B, C.c_void_p(tgt.data_ptr()), M, 1)
            self._check(rc, "svnicp_set_clouds")      # device-to-device copies queued on the library's stream; self._keep holds the
            #                                           tensors until the next add_cloud, and the caller must not write to them
            #                                           before stein_align has returned (include/svnicp_hip.h, svnicp_set_clouds)
        else:
            src = np.ascontiguousarray(np.asarray(new_cloud, np.float64).reshape(-1, 3))
            tgt = np.ascontiguousarray(np.asarray(target, np.float64).reshape(-1, 3))
            B, M = src.shape[0], tgt.shape[0]
            rc = self._L.svnicp_set_clouds(self._h, src.ctypes.data_as(C.c_void_p), B, tgt.ctypes.data_as(C.c_void_p),
                                           M, 0)
            self._check(rc, "svnicp_set_clouds")
        self._B, self._M = B, M
        init = self._pose_arg(init_pose)
        self._P = init.shape[1]
        self._check(self._L.svnicp_set_particles(self._h, init.ctypes.data_as(C.POINTER(C.c_double)), self._P),
                    "svnicp_set_particles")

    def add_cloud_device_target(self, new_cloud, target_devptr: int, M: int, init_pose):
        """add_cloud with a host source scan and a target that already lives in HBM (DeviceVoxelHashMap.get_map):
        the source goes over PCIe, the target is copied device-to-device."""
        src = np.ascontiguousarray(np.asarray(new_cloud, np.float64).reshape(-1, 3))
        self._check(self._L.svnicp_set_source(self._h, src.ctypes.data_as(C.c_void_p), src.shape[0], 0), "svnicp_set_source")
        self._check(self._L.svnicp_set_target(self._h, C.c_void_p(int(target_devptr)), int(M), 1), "svnicp_set_target")
        self._B, self._M = src.shape[0], int(M)
        init = self._pose_arg(init_pose)
        self._P = init.shape[1]
        self._check(self._L.svnicp_set_particles(self._h, init.ctypes.data_as(C.POINTER(C.c_double)), self._P),
                    "svnicp_set_particles")

    def add_cloud_device(self, source_devptr: int, B: int, target_devptr: int, M: int, init_pose):
        """add_cloud with both clouds already in HBM (float64 rows: DevicePreprocessor.source, DeviceVoxelHashMap.get_map):
        two device-to-device copies, nothing crosses PCIe but the particle prior."""
        self._check(self._L.svnicp_set_source(self._h, C.c_void_p(int(source_devptr)), int(B), 1), "svnicp_set_source")
        self._check(self._L.svnicp_set_target(self._h, C.c_void_p(int(target_devptr)), int(M), 1), "svnicp_set_target")
        self._B, self._M = int(B), int(M)
        init = self._pose_arg(init_pose)
        self._P = init.shape[1]
        self._check(self._L.svnicp_set_particles(self._h, init.ctypes.data_as(C.POINTER(C.c_double)), self._P),
                    "svnicp_set_particles")

    def set_initial_mean(self, pose):
        """SVGDICP::set_initial_mean(gtsam::Pose3) (include/core/SVGDICP.h:102-110).
        ``pose``: 4x4 homogeneous matrix, or a (R[3,3], t[3]) pair."""
        if isinstance(pose, (tuple, list)) and len(pose) == 2:
            R, t = np.asarray(pose[0], np.float64).reshape(3, 3), np.asarray(pose[1], np.float64).reshape(3)
        else:
            T = np.asarray(pose, np.float64).reshape(4, 4)
            R, t = T[:3, :3], T[:3, 3]
        R = np.ascontiguousarray(R).reshape(9)
        t = np.ascontiguousarray(t)
        dp = C.POINTER(C.c_double)
        self._check(self._L.svnicp_set_initial_mean(self._h, R.ctypes.data_as(dp), t.ctypes.data_as(dp)),
                    "svnicp_set_initial_mean")

    def set_k(self, k: int):
        self._K = int(k)
        self._check(self._L.svnicp_set_k(self._h, int(k)), "svnicp_set_k")

    def set_option(self, name: str, value) -> None:
        """Test / profiling knob of this context (include/svnicp_hip.h: svnicp_set_option)."""
        self._check(self._L.svnicp_set_option(self._h, str(name).encode(), str(value).encode()), "svnicp_set_option")

    def set_threshold(self, max_dist: float):
        self._check(self._L.svnicp_set_max_dist(self._h, float(max_dist)), "svnicp_set_max_dist")

    def stein_align(self) -> SteinICPState:
        return SteinICPState(self._check(self._L.svnicp_align(self._h), "svnicp_align"))

    def stein_align_async(self):
        self._check(self._L.svnicp_align_async(self._h), "svnicp_align_async")

    def synchronize(self):
        self._check(self._L.svnicp_synchronize(self._h), "svnicp_synchronize")

    def _getd(self, name: str, n: int) -> np.ndarray:
        out = np.zeros(n, np.float64)
        self._check(getattr(self._L, "svnicp_get_" + name)(self._h, out.ctypes.data_as(C.POINTER(C.c_double))),
                    "svnicp_get_" + name)
        return out

    def get_transformation(self) -> np.ndarray:
        return self._getd("transformation", 6)

    def get_distribution(self) -> np.ndarray:
        return self._getd("distribution", 6)

    def get_cov_matrix(self) -> np.ndarray:
        return self._getd("cov_matrix", 36)

    def get_particles(self) -> np.ndarray:
        return self._getd("particles", 6 * self._P)

    def get_particle_weight(self) -> np.ndarray:
        return self._getd("particle_weight", self._P)

    def get_particle_history(self) -> np.ndarray:
        out = np.zeros((int(self.config.iterations), 6 * self._P), np.float32)
        self._check(self._L.svnicp_get_particle_history(self._h, out.ctypes.data_as(C.POINTER(C.c_float))),
                    "svnicp_get_particle_history")
        return out

    def get_runtime(self) -> np.ndarray:
        """{knn_duration_, update_duration_, finish_iter_} (include/core/SVGDICP.h:94-96), seconds on the GPU.
        finish_iter_ is the reference's: the constructor's ``iterations`` unless an SVGD-mode early stop changed it."""
        return self._getd("runtime", 3)

    # -- test / bench taps ------------------------------------------------------------------
    def get_gpu_ms(self) -> np.ndarray:
        return self._getd("gpu_ms", 3)

    KERNEL_CLASSES = ("stage_a_knn", "k_build_table", "k_stein_search", "k_stein_accumulate", "k_reduce_partials",
                      "k_particle_update")  # include/svnicp_hip.h SVNICP_KERNEL_CLASSES

    def set_profile(self, on, classes=None):
        """Bracket kernel launches with hipEvents: every class (on=True) or only the named ``classes``."""
        v = int(bool(on))
        if on and classes:
            v = 0
            for k in classes:
                v |= 1 << (self.KERNEL_CLASSES.index(k) + 1)
        self._check(self._L.svnicp_set_profile(self._h, v), "svnicp_set_profile")

    def get_kernel_ms(self) -> dict:
        """{kernel class: (total ms in the last align, launches)} — hipEvents on the library's stream."""
        ms = np.zeros(len(self.KERNEL_CLASSES), np.float64)
        n = np.zeros(len(self.KERNEL_CLASSES), np.int32)
        self._check(self._L.svnicp_get_kernel_ms(self._h, ms.ctypes.data_as(C.POINTER(C.c_double)),
                                                 n.ctypes.data_as(C.POINTER(C.c_int32))), "svnicp_get_kernel_ms")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.KERNEL_CLASSES)}

    def get_candidates(self) -> np.ndarray:
        out = np.zeros((self._B, self._K), np.int32)
        self._check(self._L.svnicp_get_candidates(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))),
                    "svnicp_get_candidates")
        return out

    def get_knn_fallbacks(self) -> int:
        """Queries the pre-filtered stage-A kernel handed to the streaming fallback (-1: streaming kernel only)."""
        v = C.c_int(0)
        self._check(self._L.svnicp_get_knn_fallbacks(self._h, C.byref(v)), "svnicp_get_knn_fallbacks")
        return int(v.value)

    def get_knn_fallback_rows(self) -> np.ndarray:
        """Source rows the pruned stage-A kernel handed to the streaming fallback (unordered)."""
        n = C.c_int(0)
        out = np.zeros(max(1, self._B), np.int32)
        self._check(self._L.svnicp_get_knn_fallback_rows(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size,
                                                         C.byref(n)), "svnicp_get_knn_fallback_rows")
        return out[:max(0, min(int(n.value), out.size))].copy()

    def get_knn_survivors(self) -> np.ndarray:
        """Per source point: targets that survived the f32 pre-filter of the pruned stage-A kernel (record_trace)."""
        out = np.zeros(self._B, np.int32)
        self._check(self._L.svnicp_get_knn_survivors(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))),
                    "svnicp_get_knn_survivors")
        return out

    def get_iterations_run(self) -> int:
        """Iterations the last align executed (the early stop may end it before ``iterations``)."""
        v = C.c_int(0)
        self._check(self._L.svnicp_get_iterations_run(self._h, C.byref(v)), "svnicp_get_iterations_run")
        return int(v.value)

    def get_ambiguous_steps(self) -> int:
        """Wave steps whose float32 nearest-of-K search had to be redone in float64 (-1: f64 kernel only)."""
        v = C.c_int(0)
        self._check(self._L.svnicp_get_ambiguous_steps(self._h, C.byref(v)), "svnicp_get_ambiguous_steps")
        return int(v.value)

    def get_ambiguous_pairs(self) -> int:
        """(point, particle) pairs the bf16 matrix-pipe search handed to its exact float64 pass (-1: another search kernel)."""
        v = C.c_int64(0)
        self._check(self._L.svnicp_get_ambiguous_pairs(self._h, C.byref(v)), "svnicp_get_ambiguous_pairs")
        return int(v.value)

    def get_candidate_dist2(self) -> np.ndarray:
        return self._getd("candidate_dist2", self._B * self._K).reshape(self._B, self._K)

    def get_trace(self, with_corr: bool = True) -> dict:
        I, P, B = int(self.config.iterations), self._P, self._B
        d = dict(H=np.zeros((I, P, 36)), b=np.zeros((I, P, 6)), newton=np.zeros((I, P, 6)), phi=np.zeros((I, P, 6)),
                 h=np.zeros(I))
        corr = np.zeros((I, P, B), np.int32) if with_corr else None
        dp = C.POINTER(C.c_double)
        rc = self._L.svnicp_get_trace(self._h, corr.ctypes.data_as(C.POINTER(C.c_int32)) if with_corr else None,
                                      d["H"].ctypes.data_as(dp), d["b"].ctypes.data_as(dp),
                                      d["newton"].ctypes.data_as(dp), d["phi"].ctypes.data_as(dp),
                                      d["h"].ctypes.data_as(dp))
        self._check(rc, "svnicp_get_trace")
        if with_corr:
            d["corr"] = corr
        return d


class SVNICP(_SolverBase):
    """svnicp::SVNICP (include/core/SVNICP.h:29-78)."""
    _mode = 0


class SVGDICP(_SolverBase):
    """svnicp::SVGDICP (include/core/SVGDICP.h:64-210), first-order mode."""
    _mode = 1

"""Caller glue around the solver: the scan-to-map sequence of the reference's odometry node, ROS-free.

SURVEY.md §8(f)-1.  Mirrors ``OdometryPipeline::ICP_processing`` for the ``estimator == ICP`` configuration
(/root/reference/svn-icp/src/core/OdometryPipeline.cpp:556-647): crop -> uniform down-sample (map cloud at
0.5·voxel, solver cloud at 1.5·voxel of that) -> constant-velocity pose prediction -> particle prior ->
local-map query -> solver -> pose = prediction · correction -> map insert.  The pre-/post-processing is host
code (numpy), exactly as it is host code (PCL / GTSAM / tsl::robin_map) in the reference; only the solver call
touches the GPU, through the same ``SVNICP`` class the parity tests use.

Parity status: *unpinned*.  PCL, GTSAM and rclcpp are absent from this image, so the reference's pipeline cannot
be built or run here and it holds no fixtures for these helpers; every function below restates the cited
reference lines and is covered by property tests (tests/test_pipeline_cpu.py).  Where PCL leaves an order
unspecified (hash-map iteration) this code picks a deterministic one and says so.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np

from .solver import SVNICP, ParticleWeightOpt, SteinICPParam, SteinICPState, initialize_particles

# particle prior bounds, OdometryPipeline.cpp:661-667
PRIOR_UB = np.array([0.3, 0.2, 0.1, 0.004, 0.004, 0.012])
PRIOR_LB = -PRIOR_UB


# ----------------------------------------------------------------------------- SE(3) helpers (gtsam::Pose3 semantics)
def _hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(w) -> np.ndarray:
    w = np.asarray(w, float)
    th = float(np.linalg.norm(w))
    K = _hat(w)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + (math.sin(th) / th) * K + ((1.0 - math.cos(th)) / th ** 2) * K @ K


def so3_log(R) -> np.ndarray:
    R = np.asarray(R, float)
    c = max(-1.0, min(1.0, 0.5 * (np.trace(R) - 1.0)))
    th = math.acos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    return (th / (2.0 * math.sin(th))) * v


def se3_exp(xi) -> np.ndarray:
    """gtsam::Pose3::Expmap, xi = [omega, v]."""
    xi = np.asarray(xi, float)
    w, v = xi[:3], xi[3:]
    th = float(np.linalg.norm(w))
    K = _hat(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + ((1.0 - math.cos(th)) / th ** 2) * K + ((th - math.sin(th)) / th ** 3) * K @ K
    T = np.eye(4)
    T[:3, :3] = so3_exp(w)
    T[:3, 3] = V @ v
    return T


def se3_log(T) -> np.ndarray:
    """gtsam::Pose3::Logmap -> [omega, v]."""
    T = np.asarray(T, float)
    w = so3_log(T[:3, :3])
    th = float(np.linalg.norm(w))
    K = _hat(w)
    if th < 1e-10:
        Vinv = np.eye(3) - 0.5 * K
    else:
        Vinv = np.eye(3) - 0.5 * K + (1.0 / th ** 2 - (1.0 + math.cos(th)) / (2.0 * th * math.sin(th))) * K @ K
    return np.concatenate([w, Vinv @ T[:3, 3]])


def correction_to_pose(x6) -> np.ndarray:
    """svnicp::tensor2gtsamPose3 (src/core/ICPUtils.cpp:84-98): [x,y,z,rx,ry,rz] -> Pose3(Rot3::Expmap(r), t)."""
    x6 = np.asarray(x6, float).reshape(6)
    T = np.eye(4)
    T[:3, :3] = so3_exp(x6[3:])
    T[:3, 3] = x6[:3]
    return T


# ----------------------------------------------------------------------------- pre-processing
def crop_pointcloud(points: np.ndarray, min_range: float, max_range: float, scan_max_range: float = 0.0):
    """OdometryPipeline::crop_pointcloud (OdometryPipeline.cpp:692-704): keep min_range² < |p|² < max_range².
    Also returns the updated ``scan_max_range_`` — which the reference sets to the largest SQUARED norm seen
    (:699) and later uses as a length (:578); mirrored as is."""
    p = np.asarray(points, float)[:, :3]
    f = p.astype(np.float32)
    # float32 arithmetic, left to right, as pt.x*pt.x + pt.y*pt.y + pt.z*pt.z on pcl::PointXYZ (:698); the comparisons
    # promote the float to double (max_range_ is a double)
    n2 = ((f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1]) + f[:, 2] * f[:, 2]).astype(np.float64)
    keep = (n2 < max_range * max_range) & (n2 > min_range * min_range)
    if n2.size and not np.all(np.isnan(n2)):
        scan_max_range = max(scan_max_range, float(np.nanmax(n2)))
    return p[keep], scan_max_range


def downsample_uniform(points: np.ndarray, radius: float) -> np.ndarray:
    """pcl::UniformSampling with ``setRadiusSearch(radius)`` (OdometryPipeline.cpp:684-690): a grid of leaf size
    ``radius`` anchored at floor(min/leaf); per occupied leaf the point closest to the leaf centre survives (first
    one wins ties).  PCL emits leaves in hash-map order; here: ascending leaf index."""
    p = np.asarray(points, float)
    if p.shape[0] == 0 or radius <= 0:
        return p.copy()
    inv = 1.0 / radius
    mn = np.floor(p.min(axis=0) * inv).astype(np.int64)
    mx = np.floor(p.max(axis=0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = np.floor(p * inv).astype(np.int64) - mn
    leaf = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    centre = (ijk + mn + 0.5) * radius
    e = p - centre
    d2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
    order = np.lexsort((np.arange(p.shape[0]), d2, leaf))   # by leaf, then distance to the centre, then input order
    first = np.ones(order.size, bool)
    first[1:] = leaf[order][1:] != leaf[order][:-1]
    return p[order[first]]


# ----------------------------------------------------------------------------- local map
def transform_f32(cloud, T) -> np.ndarray:
    """pcl::transformPointCloud of float32 points with gtsam's DOUBLE Matrix4 (VoxelHashMap.cpp:23-25): every coordinate
    q = ((R0·x + R1·y) + R2·z) + t is formed in float64 from the widened float32 point and rounded once to float32 — the
    same expression as registration_pipeline.hpp and voxel_map.hip, so all three maps hold identical points.  (Parity with
    PCL itself is unpinned: PCL is not in this image.)"""
    c = np.asarray(cloud, np.float32).astype(np.float64)
    R = np.asarray(T, float)[:3, :3]
    t = np.asarray(T, float)[:3, 3]
    out = np.empty((c.shape[0], 3), np.float32)
    for d in range(3):
        out[:, d] = (((R[d, 0] * c[:, 0] + R[d, 1] * c[:, 1]) + R[d, 2] * c[:, 2]) + t[d]).astype(np.float32)
    return out


class DeviceVoxelHashMap:
    """The same map resident in HBM (svnicp_map_* of the C ABI, csrc/voxel_map.hip): ``add_pointcloud`` uploads only the new
    points, ``get_map`` leaves the selected points in device memory as float64 rows and returns (device pointer, count) for
    ``SVNICP.add_cloud_device_target``.  Voxels come out in ascending (x, y, z) index, points of a voxel in insertion order."""

    def __init__(self, voxel_size: float, max_range: float, max_points: int, device: int = 0, capacity_voxels: int = 0):
        import ctypes as C
        from . import binding
        self._C = C
        self._L = binding.load_library()
        self._h = C.c_void_p()
        rc = self._L.svnicp_map_create(int(device), float(voxel_size), float(max_range), int(max_points), int(capacity_voxels),
                                       C.byref(self._h))
        if rc != 0:
            raise binding.SvnIcpError(f"svnicp_map_create failed ({rc}): {self._L.svnicp_map_last_error(None).decode()}")
        self.bytes_uploaded = 0

    def _chk(self, rc, what):
        if rc != 0:
            from . import binding
            raise binding.SvnIcpError(f"{what} failed ({rc}): {self._L.svnicp_map_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.svnicp_map_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = self._C.c_int64(0)
        self._chk(self._L.svnicp_map_size(self._h, self._C.byref(n)), "svnicp_map_size")
        return int(n.value)

    def empty(self) -> bool:
        return len(self) == 0

    def skipped_points(self) -> int:
        """Points add_pointcloud has not stored (outside +-2^20 voxels, or NaN) since creation."""
        n = self._C.c_int64(0)
        self._chk(self._L.svnicp_map_skipped_points(self._h, self._C.byref(n)), "svnicp_map_skipped_points")
        return int(n.value)

    def add_pointcloud(self, cloud: np.ndarray, pose: np.ndarray):
        C = self._C
        pts = np.ascontiguousarray(np.asarray(cloud, np.float32)[:, :3])
        T = np.asarray(pose, float)
        R = np.ascontiguousarray(T[:3, :3]).reshape(9)
        t = np.ascontiguousarray(T[:3, 3])
        dp = C.POINTER(C.c_double)
        self._chk(self._L.svnicp_map_add_cloud(self._h, pts.ctypes.data_as(C.c_void_p), pts.shape[0], 0, R.ctypes.data_as(dp),
                                               t.ctypes.data_as(dp)), "svnicp_map_add_cloud")
        self.bytes_uploaded += pts.nbytes

    def add_pointcloud_device(self, devptr: int, n: int, pose: np.ndarray):
        """add_pointcloud for float32 rows that already live in HBM (DevicePreprocessor)."""
        C = self._C
        T = np.asarray(pose, float)
        R = np.ascontiguousarray(T[:3, :3]).reshape(9)
        t = np.ascontiguousarray(T[:3, 3])
        dp = C.POINTER(C.c_double)
        self._chk(self._L.svnicp_map_add_cloud(self._h, C.c_void_p(int(devptr)), int(n), 1, R.ctypes.data_as(dp), t.ctypes.data_as(dp)),
                  "svnicp_map_add_cloud")

    def get_map(self, pose=None, max_range: float | None = None):
        """-> (device pointer of float64 [M][3], M)"""
        C = self._C
        n = C.c_int64(0)
        if pose is None:
            self._chk(self._L.svnicp_map_query(self._h, None, -1.0, C.byref(n)), "svnicp_map_query")
        else:
            c = np.ascontiguousarray(np.asarray(pose, float)[:3, 3])
            self._chk(self._L.svnicp_map_query(self._h, c.ctypes.data_as(C.POINTER(C.c_double)), float(max_range), C.byref(n)),
                      "svnicp_map_query")
        return int(self._L.svnicp_map_points_devptr(self._h) or 0), int(n.value)

    def download(self) -> np.ndarray:
        """The rows of the last get_map as a host array (test tap)."""
        C = self._C
        n = C.c_int64(0)
        self._chk(self._L.svnicp_map_download(self._h, None, 0, C.byref(n)), "svnicp_map_download")
        out = np.zeros((int(n.value), 3))
        if out.size:
            self._chk(self._L.svnicp_map_download(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.shape[0], C.byref(n)),
                      "svnicp_map_download")
        return out


class DevicePreprocessor:
    """crop_pointcloud + the two uniform samplings of a scan on the device (svnicp_prep_* of the C ABI, csrc/scan_prep.hip):
    the raw float32 scan is uploaded once, the cropped cloud, the map cloud (float32) and the source cloud (float64 rows)
    stay in HBM.  Same points in the same order as crop_pointcloud / downsample_uniform above."""

    def __init__(self, device: int = 0):
        import ctypes as C
        from . import binding
        self._C = C
        self._L = binding.load_library()
        self._h = C.c_void_p()
        rc = self._L.svnicp_prep_create(int(device), C.byref(self._h))
        if rc:
            raise binding.SvnIcpError(f"svnicp_prep_create failed ({rc}): {self._L.svnicp_prep_last_error(None).decode()}")
        self.n_cropped = self.n_map = self.n_source = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.svnicp_prep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scan(self, points: np.ndarray, min_range: float, max_range: float, voxel_size: float, scan_max_range: float) -> float:
        """-> updated scan_max_range; counts in n_cropped / n_map / n_source, clouds behind the *_ptr properties."""
        from . import binding
        C = self._C
        pts = np.ascontiguousarray(np.asarray(points, np.float32)[:, :3])
        smr = C.c_double(float(scan_max_range))
        nc, nm, ns = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = self._L.svnicp_prep_scan(self._h, pts.ctypes.data_as(C.c_void_p), pts.shape[0], 0, float(min_range), float(max_range),
                                      float(voxel_size), C.byref(smr), C.byref(nc), C.byref(nm), C.byref(ns))
        if rc:
            raise binding.SvnIcpError(f"svnicp_prep_scan failed ({rc}): {self._L.svnicp_prep_last_error(self._h).decode()}")
        self.n_cropped, self.n_map, self.n_source = int(nc.value), int(nm.value), int(ns.value)
        self.bytes_uploaded = pts.nbytes
        return float(smr.value)

    @property
    def cropped_ptr(self) -> int:
        return int(self._L.svnicp_prep_cropped_devptr(self._h) or 0)

    @property
    def map_cloud_ptr(self) -> int:
        return int(self._L.svnicp_prep_map_cloud_devptr(self._h) or 0)

    @property
    def source_f32_ptr(self) -> int:
        return int(self._L.svnicp_prep_source_f32_devptr(self._h) or 0)

    @property
    def source_ptr(self) -> int:
        return int(self._L.svnicp_prep_source_devptr(self._h) or 0)

    def download(self, which: int) -> np.ndarray:
        """0 cropped, 1 map cloud, 2 source — float32 rows (test tap)."""
        C = self._C
        n = C.c_int64(0)
        self._L.svnicp_prep_download(self._h, int(which), None, 0, C.byref(n))
        out = np.zeros((int(n.value), 3), np.float32)
        if out.size:
            self._L.svnicp_prep_download(self._h, int(which), out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(n))
        return out


class VoxelHashMap:
    """svnicp::VoxelHashMap (src/core/VoxelHashMap.cpp:22-101): voxel -> at most ``max_points`` points, in insertion
    order; voxel index = coordinates / voxel_size truncated TOWARD ZERO (Eigen ``cast<int>``, :29); a voxel is dropped
    when its FIRST point is farther than ``max_range`` from the current position (:89-97), and is selected by
    ``get_map(pose, r)`` when its first point is closer than ``r`` (:48-58).  Points are float32 like pcl::PointXYZ."""

    def __init__(self, voxel_size: float, max_range: float, max_points: int):
        self.voxel_size = float(voxel_size)
        self.max_range = float(max_range)
        self.max_points = int(max_points)
        self._vox: dict[tuple, list] = {}

    def __len__(self):
        return len(self._vox)

    def empty(self) -> bool:
        return not self._vox

    def add_pointcloud(self, cloud: np.ndarray, pose: np.ndarray):
        T = np.asarray(pose, float)
        pts = transform_f32(cloud, T)
        idx = np.trunc(pts / np.float32(self.voxel_size)).astype(np.int64)
        # group by voxel, keeping the input order inside a voxel
        order = np.lexsort((np.arange(idx.shape[0]), idx[:, 2], idx[:, 1], idx[:, 0]))
        si = idx[order]
        brk = np.ones(order.size, bool)
        brk[1:] = np.any(si[1:] != si[:-1], axis=1)
        starts = np.flatnonzero(brk)
        ends = np.append(starts[1:], order.size)
        for s, e in zip(starts, ends):
            key = (int(si[s, 0]), int(si[s, 1]), int(si[s, 2]))
            lst = self._vox.get(key)
            if lst is None:
                lst = []
                self._vox[key] = lst
            room = self.max_points - len(lst)
            if room > 0:
                lst.extend(pts[order[s:min(e, s + room)]])
        self.remove_far(T[:3, 3])

    def remove_far(self, position):
        pos = np.asarray(position, float)
        r2 = self.max_range * self.max_range
        far = [k for k, v in self._vox.items() if float(np.sum((v[0].astype(float) - pos) ** 2)) > r2]
        for k in far:
            del self._vox[k]

    def get_map(self, pose=None, max_range: float | None = None) -> np.ndarray:
        if not self._vox:
            return np.zeros((0, 3))
        if pose is None:
            chunks = [np.asarray(v) for v in self._vox.values()]
        else:
            pos = np.asarray(pose, float)[:3, 3]
            r2 = max_range * max_range
            chunks = [np.asarray(v) for v in self._vox.values() if float(np.sum((v[0].astype(float) - pos) ** 2)) < r2]
        if not chunks:
            return np.zeros((0, 3))
        return np.concatenate(chunks, 0).astype(np.float64)


# ----------------------------------------------------------------------------- pose prediction
def pose_prediction(poses: list, times: list, new_time: float) -> np.ndarray:
    """OdometryPipeline::pose_prediction (OdometryPipeline.cpp:706-737): constant twist between the last two poses,
    scaled by the time ratio; identity / last pose while fewer than two poses exist."""
    if len(poses) == 0:
        return np.eye(4)
    if len(poses) < 2:
        return np.array(poses[-1], float)
    T0, T1 = np.asarray(poses[-2], float), np.asarray(poses[-1], float)
    dt = times[-1] - times[-2]
    delta = np.linalg.inv(T0) @ T1
    ratio = (new_time - times[-1]) / dt
    return T1 @ se3_exp(ratio * se3_log(delta))


# ----------------------------------------------------------------------------- the sequence
@dataclass
class PipelineConfig:
    """Field names follow the node's parameters (OdometryPipeline.cpp parameter block; config/*.yaml)."""
    min_range: float = 1.0
    max_range: float = 100.0
    voxel_size: float = 1.0
    map_voxel_size: float = 1.0
    map_voxel_max_points: int = 20
    map_range: float = 100.0
    particle_count: int = 128
    gpu_map: bool = False          # keep the local map in HBM (DeviceVoxelHashMap): the target never crosses PCIe
    gpu_prep: bool = False         # with gpu_map: crop and both uniform samplings on the device (DevicePreprocessor): the raw scan is uploaded, no host pass over the points
    solver: SteinICPParam = field(default_factory=lambda: SteinICPParam(iterations=20, lr=1.0, max_dist=1.0, KNN_count=100))
    seed: int = 0


@dataclass
class ScanResult:
    stamp: float
    pose: np.ndarray            # 4x4, map <- sensor
    initial_guess: np.ndarray   # 4x4
    correction: np.ndarray | None = None     # [6] solver mean (x,y,z,rx,ry,rz)
    variance: np.ndarray | None = None       # [6]
    cov: np.ndarray | None = None            # [36]
    particles: np.ndarray | None = None      # [6*P]
    weights: np.ndarray | None = None        # [P]
    preprocessing_s: float = 0.0
    align_s: float = 0.0
    state: int | None = None


class RegistrationPipeline:
    """One instance per sensor; ``process_scan`` is one pass of ICP_processing's loop body for one LiDAR frame."""

    def __init__(self, cfg: PipelineConfig | None = None, device: int = 0):
        self.cfg = cfg or PipelineConfig()
        self.device = device
        self.map = (DeviceVoxelHashMap(self.cfg.map_voxel_size, self.cfg.map_range, self.cfg.map_voxel_max_points, device)
                    if self.cfg.gpu_map else
                    VoxelHashMap(self.cfg.map_voxel_size, self.cfg.map_range, self.cfg.map_voxel_max_points))
        self.bytes_h2d = 0            # cloud bytes sent to the GPU so far (source scans + map traffic)
        self.poses: list[np.ndarray] = []
        self.times: list[float] = []
        self.scan_max_range = 0.0
        self._rng = np.random.default_rng(self.cfg.seed)
        self._solver: SVNICP | None = None
        self._prep: DevicePreprocessor | None = None

    def _particles(self) -> np.ndarray:
        return initialize_particles(self.cfg.particle_count, PRIOR_UB, PRIOR_LB, self._rng)   # set_initPose, :661-667

    def process_scan(self, points: np.ndarray, stamp: float) -> ScanResult:
        c = self.cfg
        t0 = time.perf_counter()
        dev = c.gpu_map and c.gpu_prep
        if dev:
            if self._prep is None:
                self._prep = DevicePreprocessor(self.device)
            self.scan_max_range = self._prep.scan(points, c.min_range, c.max_range, c.voxel_size, self.scan_max_range)  # :556-560
            self.bytes_h2d += self._prep.bytes_uploaded
        else:
            cropped, self.scan_max_range = crop_pointcloud(points, c.min_range, c.max_range, self.scan_max_range)   # :556
            to_map = downsample_uniform(cropped, 0.5 * c.voxel_size)                                                # :559
            source = downsample_uniform(to_map, 1.5 * c.voxel_size)                                                 # :560
        guess = pose_prediction(self.poses, self.times, stamp)                                                  # :563-564
        init = self._particles()                                                                                # :573
        if self.map.empty():                                                                                    # :585-593
            # the reference's downsample_uniform filters its input IN PLACE (:684-690): at :585 *cropped_cloud already holds
            # the 0.5-voxel sampling, and that is what seeds the map
            if dev:
                self.map.add_pointcloud_device(self._prep.map_cloud_ptr, self._prep.n_map, guess)
            else:
                self.map.add_pointcloud(to_map, guess)
            self.poses.append(guess); self.times.append(stamp)
            return ScanResult(stamp, guess, guess, preprocessing_s=time.perf_counter() - t0)
        if self._solver is None:
            self._solver = SVNICP(c.solver, init, ParticleWeightOpt(), device=self.device)
        s = self._solver
        if c.gpu_map:
            ptr, M = self.map.get_map(guess, self.scan_max_range + 10.0)                                        # :577-578
            if M == 0:
                ptr, M = self.map.get_map()                                                                     # :579-581
            if dev:
                s.add_cloud_device(self._prep.source_ptr, self._prep.n_source, ptr, M, init)                    # :583
            else:
                s.add_cloud_device_target(source, ptr, M, init)                                                 # :583
                self.bytes_h2d += source.shape[0] * 24
        else:
            target = self.map.get_map(guess, self.scan_max_range + 10.0)                                        # :577-578
            if target.shape[0] == 0:
                target = self.map.get_map()                                                                     # :579-581
            s.add_cloud(source, target, init)                                                                   # :583
            self.bytes_h2d += (source.shape[0] + target.shape[0]) * 24
        t1 = time.perf_counter()
        s.set_initial_mean(guess)                                                                               # :601
        state = s.stein_align()                                                                                 # :602
        if state != SteinICPState.ALIGN_SUCCESS:                                                                # :602-604
            return ScanResult(stamp, guess, guess, preprocessing_s=t1 - t0, align_s=time.perf_counter() - t1, state=int(state))
        corr = s.get_transformation()                                                                           # :605
        pose = guess @ correction_to_pose(corr)                                                                 # updater_, :37-46
        res = ScanResult(stamp, pose, guess, corr, s.get_distribution(), s.get_cov_matrix(), s.get_particles().reshape(-1),
                         s.get_particle_weight(), t1 - t0, 0.0, int(state))
        # … and at :630 *voxelized_cloud_toMap holds the 1.5-voxel sampling (the second in-place filter, :560): the map is
        # updated with the same points the solver registered
        if dev:
            self.map.add_pointcloud_device(self._prep.source_f32_ptr, self._prep.n_source, pose)                # :630
        else:
            self.map.add_pointcloud(source, pose)                                                               # :630
            if c.gpu_map:
                self.bytes_h2d += source.shape[0] * 12
        self.poses.append(pose); self.times.append(stamp)                                                       # :630
        res.align_s = time.perf_counter() - t1
        return res

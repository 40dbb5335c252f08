"""Synthetic 64-beam spinning-LiDAR scan pairs — deterministic, numpy only.

Scene: ground plane z = -1.7 m, the four walls of a 60 m x 40 m box, 32 random axis-aligned
boxes.  Sensor: 64 beams with elevations uniform in [-24.8 deg, +2 deg] (HDL-64E span, cf.
/root/reference/svn-icp/include/segmentation/ImageProjection.h:63-67), B/64 azimuth columns, range
noise N(0, 0.02 m), ranges clipped to [1, 100] m, coordinates rounded to float32 and widened to
float64 (the reference converts PCL float points to double, src/core/ICPUtils.cpp:27-43).
Randomness is a counter-based splitmix64 stream owned by this file (seed 20250718), so the
clouds are bit-identical on every machine and numpy version.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SEED = 20250718
_M64 = (1 << 64) - 1

CONFIGS = {  # BASELINE.json configs (P particles, B source points, M target points, iterations)
    "C1": dict(P=1, B=4096, M=8192, I=20),
    "C2": dict(P=32, B=65536, M=131072, I=20),
    "C3": dict(P=128, B=131072, M=262144, I=20),
    "C4": dict(P=512, B=131072, M=262144, I=20),
    "C5": dict(P=128, B=131072, M=2097152, I=20),
    "P256": dict(P=256, B=131072, M=262144, I=20),
    "P1024": dict(P=1024, B=131072, M=262144, I=20),  # per-GPU view of an 8-GPU weak-scaling run (update kernel sees all particles)
}
# known displacement of the source scan from the first target pose (x,y,z m; roll,pitch,yaw deg)
TRUE_OFFSET = (0.20, -0.10, 0.05, 0.5, -0.3, 1.0)
# particle prior bounds, OdometryPipeline.cpp:661-667
PARTICLE_UB = (0.3, 0.2, 0.1, 0.004, 0.004, 0.012)
PARTICLE_LB = tuple(-v for v in PARTICLE_UB)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def uniform01(stream: int, n: int, seed: int = SEED) -> np.ndarray:
    """n doubles in [0,1) from counter-based stream `stream`."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 1000003 + stream], dtype=np.uint64))[0]
        ctr = np.arange(n, dtype=np.uint64) + base
        bits = _splitmix64(ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal01(stream: int, n: int, seed: int = SEED) -> np.ndarray:
    u1 = uniform01(2 * stream + 1, n, seed)
    u2 = uniform01(2 * stream + 2, n, seed)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def rot_zyx(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    return np.array([[cp * cy, sr * sp * cy - cr * sy, sr * sy + cr * sp * cy],
                     [cp * sy, cr * cy + sr * sp * sy, cr * sp * sy - sr * cy],
                     [-sp, sr * cp, cr * cp]])


@dataclass
class Scene:
    boxes_lo: np.ndarray  # [32,3]
    boxes_hi: np.ndarray  # [32,3]
    half_x: float = 30.0
    half_y: float = 20.0
    ground_z: float = -1.7


def make_scene(seed: int = SEED) -> Scene:
    u = uniform01(7, 32 * 6, seed).reshape(32, 6)
    cx = -26.0 + 52.0 * u[:, 0]
    cy = np.where(u[:, 1] < 0.5, -17.0 + 13.0 * (2 * u[:, 1]), 4.0 + 13.0 * (2 * u[:, 1] - 1.0))  # keep |y| < 3 m free
    sx, sy, sz = 0.5 + 2.5 * u[:, 2], 0.5 + 2.5 * u[:, 3], 0.5 + 3.5 * u[:, 4]
    lo = np.stack([cx - sx / 2, cy - sy / 2, np.full(32, -1.7)], 1)
    hi = np.stack([cx + sx / 2, cy + sy / 2, -1.7 + sz], 1)
    return Scene(lo, hi)


def _raycast(scene: Scene, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """Nearest positive hit distance of rays origin + t*dirs ([N,3]) with the scene."""
    n = dirs.shape[0]
    t_best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground
        t = (scene.ground_z - origin[2]) / dirs[:, 2]
        t_best = np.where((t > 0) & (t < t_best), t, t_best)
        # walls (interior faces of the box, infinitely tall)
        for axis, half in ((0, scene.half_x), (1, scene.half_y)):
            for sgn in (-1.0, 1.0):
                t = (sgn * half - origin[axis]) / dirs[:, axis]
                t_best = np.where((t > 0) & (t < t_best), t, t_best)
        # axis-aligned boxes: slab test, rays x boxes in chunks
        inv = 1.0 / dirs
        for c0 in range(0, n, 32768):
            sl = slice(c0, c0 + 32768)
            t0 = (scene.boxes_lo[None, :, :] - origin[None, None, :]) * inv[sl, None, :]
            t1 = (scene.boxes_hi[None, :, :] - origin[None, None, :]) * inv[sl, None, :]
            tn = np.minimum(t0, t1).max(axis=2)
            tf = np.maximum(t0, t1).min(axis=2)
            hit = (tf >= tn) & (tf > 0)
            tb = np.where(hit, np.where(tn > 0, tn, tf), np.inf).min(axis=1)
            t_best[sl] = np.minimum(t_best[sl], tb)
    return t_best


def lidar_scan(scene: Scene, R: np.ndarray, t: np.ndarray, n_points: int, stream: int, seed: int = SEED,
               noise: float = 0.02) -> np.ndarray:
    """One scan of n_points (= 64 beams x n_points/64 columns) from sensor pose (R, t) in the
    scene frame, returned in the SENSOR frame, float32-rounded float64 [n_points,3]."""
    cols = max(1, n_points // 64)
    el = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    az = (np.arange(cols) + 0.5) * (2.0 * math.pi / cols)
    azg, elg = np.meshgrid(az, el, indexing="ij")  # column-major firing order
    d_s = np.stack([np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], -1).reshape(-1, 3)
    d_w = d_s @ R.T
    rng = _raycast(scene, t, d_w)
    rng = rng + noise * normal01(stream, rng.shape[0], seed)
    rng = np.clip(rng, 1.0, 100.0)
    pts = d_s * rng[:, None]
    if pts.shape[0] < n_points:  # n_points not a multiple of 64: repeat the head
        pts = np.concatenate([pts, pts[: n_points - pts.shape[0]]], 0)
    return pts[:n_points].astype(np.float32).astype(np.float64)


@dataclass
class ScanPair:
    source: np.ndarray      # [B,3] sensor frame of the displaced pose
    target: np.ndarray      # [M,3] frame of the first target pose
    true_pose: np.ndarray   # [6] (x,y,z, so3-log) of the displacement the solver should recover
    R_true: np.ndarray
    t_true: np.ndarray


def make_pair(B: int, M: int, seed: int = SEED, offset=TRUE_OFFSET) -> ScanPair:
    scene = make_scene(seed)
    n_scans = (M + B - 1) // B
    tgt = []
    for i in range(n_scans):  # target = union of scans taken 0.5 m apart, expressed in the frame of pose 0
        ti = np.array([0.5 * i, 0.0, 0.0])
        pts = lidar_scan(scene, np.eye(3), ti, B, stream=100 + i, seed=seed)
        tgt.append((pts + ti).astype(np.float32).astype(np.float64))
    target = np.concatenate(tgt, 0)[:M]
    x, y, z, r, p, yw = offset
    R_true = rot_zyx(math.radians(r), math.radians(p), math.radians(yw))
    t_true = np.array([x, y, z])
    source = lidar_scan(scene, R_true, t_true, B, stream=900, seed=seed)
    # so(3) log of R_true
    ang = math.acos(max(-1.0, min(1.0, 0.5 * (np.trace(R_true) - 1.0))))
    w = np.zeros(3) if abs(math.sin(ang)) < 1e-12 else ang / (2 * math.sin(ang)) * np.array(
        [R_true[2, 1] - R_true[1, 2], R_true[0, 2] - R_true[2, 0], R_true[1, 0] - R_true[0, 1]])
    return ScanPair(source, target, np.concatenate([t_true, w]), R_true, t_true)


def make_particles(P: int, seed: int = SEED) -> np.ndarray:
    """Uniform particle prior of OdometryPipeline.cpp:661-667 / ICPUtils.cpp:45-58; [6,P]."""
    if P == 1:
        return np.zeros((6, 1))
    u = uniform01(5000, 6 * P, seed).reshape(6, P)
    ub = np.array(PARTICLE_UB).reshape(6, 1)
    lb = np.array(PARTICLE_LB).reshape(6, 1)
    return (ub - lb) * u + lb


def random_clouds(B: int, M: int, seed: int = 1, extent: float = 10.0, offset=(0.05, -0.03, 0.02, 0.004, -0.003, 0.006)):
    """Small generic test clouds: a random surface-ish target and a displaced, noisy subset as source."""
    u = uniform01(11, 3 * M, seed).reshape(M, 3)
    tgt = np.empty((M, 3))
    tgt[:, 0] = extent * (u[:, 0] - 0.5)
    tgt[:, 1] = extent * (u[:, 1] - 0.5)
    tgt[:, 2] = 0.5 * np.sin(tgt[:, 0]) * np.cos(0.7 * tgt[:, 1]) + 0.05 * (u[:, 2] - 0.5)
    tgt = tgt.astype(np.float32).astype(np.float64)
    pick = (uniform01(12, B, seed) * M).astype(np.int64) % M
    w = np.array(offset[3:])
    ang = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (math.sin(ang) / ang) * Kx + ((1 - math.cos(ang)) / ang ** 2) * Kx @ Kx if ang > 0 else np.eye(3)
    t = np.array(offset[:3])
    src_w = tgt[pick] + 0.01 * normal01(13, 3 * B, seed).reshape(B, 3)
    src = (src_w - t) @ R  # R^T (p - t) : source frame such that R*src + t ~ target
    return src.astype(np.float32).astype(np.float64), tgt

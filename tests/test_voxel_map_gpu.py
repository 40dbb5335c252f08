"""The local map in HBM (csrc/voxel_map.hip, svnicp_map_* of the C ABI) against the host map it replaces:
svnicp::VoxelHashMap (VoxelHashMap.cpp:22-101) as restated in svn-icp_amd/pipeline.py.  Same voxels, same points, same
order inside a voxel; range cull and range query pick the same voxels."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_rows(hm, pose=None, r=None):
    """The host map's content in the device map's output order: ascending (x, y, z) voxel index, insertion order inside."""
    items = hm._vox.items()
    if pose is not None:
        pos = np.asarray(pose, float)[:3, 3]
        items = [(k, v) for k, v in items if float(np.sum((v[0].astype(float) - pos) ** 2)) < r * r]
    rows = [np.asarray(v, np.float32) for _, v in sorted(items, key=lambda kv: kv[0])]
    return np.concatenate(rows, 0).astype(np.float64) if rows else np.zeros((0, 3))


def _pose(rng, scale):
    import math
    a = rng.normal(size=3) * 0.2
    T = np.eye(4)
    c, s = math.cos(a[2]), math.sin(a[2])
    T[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, math.cos(a[0]), -math.sin(a[0])], [0, math.sin(a[0]), math.cos(a[0])]])
    T[:3, 3] = rng.normal(size=3) * scale
    return T


@pytest.mark.parametrize("voxel,max_pts,n,extent,steps", [(1.0, 20, 30000, 40.0, 5), (0.5, 3, 20000, 6.0, 4), (2.0, 20, 120000, 300.0, 3),
                                                          (0.25, 1, 5000, 3.0, 6)])
def test_device_map_equals_host_map(hip, voxel, max_pts, n, extent, steps):
    pl = importlib.import_module(hip.__name__ + ".pipeline")
    rng = np.random.default_rng(int(voxel * 100) + max_pts)
    max_range = 0.6 * extent
    hm = pl.VoxelHashMap(voxel, max_range, max_pts)
    dm = pl.DeviceVoxelHashMap(voxel, max_range, max_pts, device=0)
    assert dm.empty() and len(dm) == 0
    for k in range(steps):
        cloud = (rng.uniform(-1, 1, size=(n, 3)) * extent * 0.5).astype(np.float32)
        cloud[: n // 10] = cloud[0]                      # duplicates: one voxel sees many points of one scan (max_points cut)
        T = _pose(rng, extent * 0.15 * k)                # the sensor moves: far voxels get culled (VoxelHashMap.cpp:89-97)
        hm.add_pointcloud(cloud, T); dm.add_pointcloud(cloud, T)
        assert len(dm) == len(hm), k
        ptr, M = dm.get_map()                            # GetMap() — everything
        want = _host_rows(hm)
        assert M == want.shape[0] and ptr != 0
        assert np.array_equal(dm.download(), want), k
        Q = _pose(rng, extent * 0.1)
        ptr, M = dm.get_map(Q, 0.3 * extent)             # GetMap(pose, r)
        want = _host_rows(hm, Q, 0.3 * extent)
        assert M == want.shape[0]
        assert np.array_equal(dm.download(), want), k
    ptr, M = dm.get_map(np.eye(4), 1e-6)                 # empty selection
    assert M == 0 and dm.download().shape == (0, 3)


def test_device_map_grows_and_rejects_out_of_range(hip):
    pl = importlib.import_module(hip.__name__ + ".pipeline")
    dm = pl.DeviceVoxelHashMap(0.1, 1e9, 2, device=0, capacity_voxels=1)      # smallest table: 65536 slots
    hm = pl.VoxelHashMap(0.1, 1e9, 2)
    rng = np.random.default_rng(0)
    cloud = (rng.uniform(-1, 1, size=(200000, 3)) * 30.0).astype(np.float32)  # ~200 k distinct voxels: the table must grow
    dm.add_pointcloud(cloud, np.eye(4)); hm.add_pointcloud(cloud, np.eye(4))
    assert len(dm) == len(hm) > 100000
    dm.get_map()
    assert np.array_equal(dm.download(), _host_rows(hm))
    # voxel index beyond +-2^20 or NaN: the point is counted and not stored, the call succeeds (one stray point of a scan
    # must not abort a drive whose map has already been updated) and the rest of the cloud goes in
    mixed = np.array([[1e9, 0, 0], [40.5, 40.5, 40.5], [np.nan, 0, 0]], np.float32)
    dm.add_pointcloud(mixed, np.eye(4)); hm.add_pointcloud(mixed[1:2], np.eye(4))
    assert dm.skipped_points() == 2
    assert len(dm) == len(hm)
    dm.get_map()
    assert np.array_equal(dm.download(), _host_rows(hm))


def test_pipeline_with_device_map_matches_host_map(hip):
    """The scan-to-map loop with the map in HBM: the same poses as with the host map (same target points; their order
    differs — voxel index instead of insertion — which only matters for exact ties), and the per-scan host-to-device
    traffic drops from source + target to source + new map points."""
    pl = importlib.import_module(hip.__name__ + ".pipeline")
    sc = hip.scans
    scene = sc.make_scene()
    out = {}
    for gpu_map in (False, True):
        cfg = pl.PipelineConfig(min_range=1.0, max_range=80.0, voxel_size=0.5, map_voxel_size=0.5, map_voxel_max_points=20, map_range=100.0,
                                particle_count=32, gpu_map=gpu_map, seed=5,
                                solver=hip.SteinICPParam(iterations=20, lr=1.0, max_dist=1.0, KNN_count=50, SVN_full_grad=False))
        pipe = pl.RegistrationPipeline(cfg, device=0)
        poses = []
        for k in range(6):
            t = np.array([0.0, 0.0, 0.05 * k]); R = sc.rot_zyx(0.0, 0.0, np.radians(0.3 * k))
            poses.append(pipe.process_scan(sc.lidar_scan(scene, R, t, 32768, stream=900 + k), stamp=0.1 * k).pose)
        out[gpu_map] = (np.array(poses), pipe.bytes_h2d, len(pipe.map))
    assert out[True][2] == out[False][2]
    assert np.allclose(out[True][0], out[False][0], rtol=0, atol=1e-9)
    print(f"host->device cloud bytes over 6 scans: host map {out[False][1]}, device map {out[True][1]} "
          f"({out[False][1] / max(1, out[True][1]):.1f}x less)")
    assert out[True][1] < 0.5 * out[False][1]

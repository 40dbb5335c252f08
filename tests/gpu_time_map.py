"""bring-up timing helper (not a pytest file): the HBM-resident voxel-hash local map at scan-to-map sizes.
20 synthetic 64-beam scans of 131 072 points, 0.5 m apart, inserted into a map (voxel 0.5 m, 20 points per voxel by default)
and the local map queried before each insertion, as RegistrationPipeline does.  Wall times with a device sync per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
from svnicp_amd.pipeline import DeviceVoxelHashMap
voxel = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
maxpts = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = pkg.scans
scene = sc.make_scene(sc.SEED)
m = DeviceVoxelHashMap(voxel, 100.0, maxpts)
ta, tq, nq = [], [], []
for i in range(20):
    t = np.array([0.5 * i, 0.0, 0.0])
    pts = sc.lidar_scan(scene, np.eye(3), t, 131072, stream=100 + i).astype(np.float32)
    T = np.eye(4); T[:3, 3] = t
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ptr, M = m.get_map(T, 110.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m.add_pointcloud(pts, T)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tq.append(t1 - t0); ta.append(t2 - t1); nq.append(M)
print("voxel %.2f m, max %d points per voxel: %d voxels after 20 scans" % (voxel, maxpts, len(m)))
print("query   : median %.3f ms (last: %d points, %.3f ms)" % (1e3 * np.median(tq[2:]), nq[-1], 1e3 * tq[-1]))
print("add     : median %.3f ms for 131072 points (includes the 1.5 MB upload)" % (1e3 * np.median(ta[2:])))

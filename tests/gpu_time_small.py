"""bring-up timing helper (not a pytest file): a registration at the size the scan-to-map loop actually hands to the solver
(a few thousand source points after the 1.5 * voxel sampling against a local map of ~50 k points)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get("SVNICP_TEST_LIB"):   # A/B builds of the library (bring-up only)
    pkg.binding._LIB_PATH = os.path.abspath(os.environ["SVNICP_TEST_LIB"])
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 128
from svnicp_amd.pipeline import downsample_uniform, crop_pointcloud
pair = pkg.scans.make_pair(65536, M); init = pkg.scans.make_particles(P)
srcc, _ = crop_pointcloud(pair.source, 1.0, 100.0)
vox = 1.0
src_ds = downsample_uniform(downsample_uniform(srcc, 0.5 * vox), 1.5 * vox)      # what the scan-to-map loop hands to the solver
print("source after the two samplings:", src_ds.shape[0], "points")
pair.source = src_ds; B = src_ds.shape[0]
prm = pkg.SteinICPParam(iterations=20, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False)
if os.environ.get("SHIPPED"):   # the reference's shipped solver settings (config/geodeAlpha.yaml: 100 iterations with early stop at 5e-4, max_dist 3)
    prm = pkg.SteinICPParam(iterations=100, lr=1.0, max_dist=3.0, KNN_count=100, SVN_full_grad=False, check_early_stop=True, convergence_threshold=5e-4)
s = pkg.SVNICP(prm, init)
src = torch.from_numpy(pair.source).cuda(); tgt = torch.from_numpy(pair.target).cuda()
def step():
    s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4)); s.stein_align(); return s.get_transformation()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): step()
torch.cuda.synchronize(); t1 = time.perf_counter()
s.set_profile(True); step()
print("B %d M %d P %d: %.3f ms per registration wall; kernel classes (ms):" % (B, M, P, 1e3 * (t1 - t0) / n), {k: round(v[0], 3) for k, v in s.get_kernel_ms().items()}, "GPU span", s.get_gpu_ms().round(3), "ambiguous wave-steps", s.get_ambiguous_steps(), "iterations run", s.get_iterations_run())

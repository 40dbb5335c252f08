"""bring-up timing helper (not a pytest file): host-side phases of one bench step at C3 and the GPU span inside stein_align."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = pkg.scans.CONFIGS[wl]
P, B, M, I = cfg["P"], cfg["B"], cfg["M"], cfg["I"]
pair = pkg.scans.make_pair(B, M); init = pkg.scans.make_particles(P)
prm = pkg.SteinICPParam(iterations=I, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False)
s = pkg.SVNICP(prm, init)
src = torch.from_numpy(pair.source).cuda(); tgt = torch.from_numpy(pair.target).cuda()
T0 = np.eye(4)
acc = np.zeros(5); n = 20
for rep in range(n + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); s.add_cloud(src, tgt, init)
    t1 = time.perf_counter(); s.set_initial_mean(T0)
    t2 = time.perf_counter(); s.stein_align_async()
    t3 = time.perf_counter(); s.synchronize()
    t4 = time.perf_counter(); s.get_transformation(); s.get_cov_matrix()
    t5 = time.perf_counter()
    if rep >= 3:
        acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4])
acc *= 1e3 / n
print("%s per registration (ms): add_cloud %.3f  set_initial_mean %.3f  enqueue of stein_align %.3f  wait for the GPU %.3f  getters %.3f  total %.3f" % ((wl,) + tuple(acc) + (acc.sum(),)))
print("GPU span of the last align [stage A + table, iterations, total] ms:", s.get_gpu_ms().round(3))

"""Where do the heavy k_knn_tiles queries live?  (survivors of the f32 pre-filter vs geometry; run on the GPU box)"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
sc = pkg.scans
cfg = sc.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
pair = sc.make_pair(cfg["B"], cfg["M"])
init = sc.make_particles(2)
prm = pkg.SteinICPParam(iterations=1, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False, check_early_stop=False,
                        record_trace=True)
s = pkg.SVNICP(prm, init, pkg.ParticleWeightOpt(), device=0)
s.set_profile(True)
for rep in range(3):
    s.add_cloud(pair.source, pair.target, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    print('stage A ms', round(s.get_kernel_ms()['stage_a_knn'][0], 3), 'fallback rows', s.get_knn_fallbacks(), flush=True)
n = s.get_knn_survivors()
d2 = s.get_candidate_dist2()
r = np.linalg.norm(pair.source, axis=1)
kth = np.sqrt(d2[:, -1])
print("total survivors %d (%.1f MB of slots); beyond 512 per query: %d entries in %d queries; beyond 256: %d entries in %d queries" % (
    n.sum(), n.sum() * 4 / 1e6, np.maximum(n - 512, 0).sum(), (n > 512).sum(), np.maximum(n - 256, 0).sum(), (n > 256).sum()))
print("survivors: median %d mean %.0f p90 %d p99 %d max %d" % (np.median(n), n.mean(), np.percentile(n, 90), np.percentile(n, 99), n.max()))
for lo, hi in ((0, 128), (128, 256), (256, 512), (512, 1024), (1024, 2049), (2049, 10**9)):
    m = (n >= lo) & (n < hi)
    if m.sum() == 0:
        continue
    print("n in [%d,%d): %6d queries  range med %.1f m  z med %.2f  K-th NN dist med %.3f m  ratio n/K med %.1f" % (
        lo, hi, m.sum(), np.median(r[m]), np.median(pair.source[m, 2]), np.median(kth[m]), np.median(n[m]) / 100))

"""Where do the heavy k_knn_tiles queries live?  (survivors of the f32 pre-filter vs geometry; run on the GPU box)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
sc = pkg.scans
cfg = sc.CONFIGS["C3"]
pair = sc.make_pair(cfg["B"], cfg["M"])
init = sc.make_particles(2)
prm = pkg.SteinICPParam(iterations=1, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False, check_early_stop=False,
                        record_trace=True)
s = pkg.SVNICP(prm, init, pkg.ParticleWeightOpt(), device=0)
s.add_cloud(pair.source, pair.target, init); s.set_initial_mean(np.eye(4)); s.stein_align()
n = s.get_knn_survivors()
d2 = s.get_candidate_dist2()
r = np.linalg.norm(pair.source, axis=1)
kth = np.sqrt(d2[:, -1])
print("survivors: median %d mean %.0f p90 %d p99 %d max %d" % (np.median(n), n.mean(), np.percentile(n, 90), np.percentile(n, 99), n.max()))
for lo, hi in ((0, 128), (128, 256), (256, 512), (512, 1024), (1024, 2049), (2049, 10**9)):
    m = (n >= lo) & (n < hi)
    if m.sum() == 0:
        continue
    print("n in [%d,%d): %6d queries  range med %.1f m  z med %.2f  K-th NN dist med %.3f m  ratio n/K med %.1f" % (
        lo, hi, m.sum(), np.median(r[m]), np.median(pair.source[m, 2]), np.median(kth[m]), np.median(n[m]) / 100))

"""bring-up helper: distribution of f32-filter survivors per query in the pruned stage-A kernel at C3."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
cfg = pkg.scans.CONFIGS["C3"]
pair = pkg.scans.make_pair(cfg["B"], cfg["M"]); init = pkg.scans.make_particles(2)
prm = pkg.SteinICPParam(iterations=1, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False, record_trace=True)
s = pkg.SVNICP(prm, init); s.add_cloud(pair.source, pair.target, init); s.stein_align()
n = s.get_knn_survivors()
print("survivors: mean", n.mean(), "median", np.median(n), "p90", np.percentile(n, 90), "p99", np.percentile(n, 99), "max", n.max(), "count>512", (n > 512).sum())
big = np.where(n > 512)[0]
print("overflow rows (first 40):", big[:40])
r = np.linalg.norm(pair.source, axis=1)
print("range of overflow queries: mean", r[big].mean() if len(big) else None, "overall mean", r.mean())
d2 = s.get_candidate_dist2()
print("K-th distance of overflow queries (m):", np.sqrt(d2[big, -1])[:10] if len(big) else None, " overall median", np.sqrt(np.median(d2[:, -1])))
print("z of overflow queries", pair.source[big, 2][:10] if len(big) else None)

"""bring-up timing helper (not a pytest file): C3 stage timings through the C ABI."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get("SVNICP_TEST_LIB"):   # A/B builds of the library (bring-up only): point the binding at another .so before it loads
    pkg.binding._LIB_PATH = os.path.abspath(os.environ["SVNICP_TEST_LIB"])
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = pkg.scans.CONFIGS[wl]
P, B, M, I = cfg["P"], cfg["B"], cfg["M"], cfg["I"]
pair = pkg.scans.make_pair(B, M); init = pkg.scans.make_particles(P)
prm = pkg.SteinICPParam(iterations=I, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=bool(int(os.environ.get("FULL", "0"))))
s = pkg.SVNICP(prm, init); s.set_profile(True)
for rep in range(3):
    s.add_cloud(pair.source, pair.target, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    print({k: round(v[0], 3) for k, v in s.get_kernel_ms().items()}, s.get_gpu_ms().round(2), "fallbacks", s.get_knn_fallbacks(), "ambiguous wave-steps", s.get_ambiguous_steps(),
          "ambiguous pairs", s.get_ambiguous_pairs(), "= %.4f %% of %d pairs" % (100.0 * s.get_ambiguous_pairs() / (float(P) * B * I), P * B * I), flush=True)
print("mean", s.get_transformation())
h = s.get_particle_history().reshape(I, 6, P)
print("particle std per iteration (x, yaw):", [ (round(float(h[i,0].std()),4), round(float(h[i,5].std()),5)) for i in range(0, I, 2)])
print("max |x - mean| per iteration:", [round(float(np.abs(h[i,0]-h[i,0].mean()).max()),4) for i in range(0,I,2)])

"""Soak run of the brute-force stage A alone (not a pytest file; run on the GPU box): random structures (integer lattices,
2000-fold duplicates, UTM-sized offsets, millimetre scale, thin planes — the fuzz generator of tests/test_gpu_parity.py),
sizes from one query / one target up to 1.3e8 pairs, K, and every queries-per-workgroup instantiation: candidate rows and
their d² against the oracle, bit for bit.  The float32 pre-filter of knn_brute.hip must never lose a neighbour.
   python tests/gpu_soak_brute.py [n_cases] [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from test_gpu_parity import _fuzz_cloud, _hip_solver
hip = g.load_package(); orc = g.load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
bad = 0
t0 = time.time()
init = np.zeros((6, 1))
for c in range(n_cases):
    seed = seed0 + c
    rng = np.random.default_rng(seed)
    kind = ["grid", "dups", "far", "tiny", "aniso"][seed % 5]
    r = rng.random()
    if r < 0.15: B, M = int(rng.integers(1, 40)), int(rng.integers(1, 300))
    elif r < 0.6: B, M = int(rng.integers(100, 3000)), int(rng.integers(500, 20000))
    else: B, M = int(rng.integers(300, 2200)), int(rng.integers(20000, 60000))
    K = int(rng.choice([1, 5, 16, 17, 50, 96, 97, 100, 128]))
    qb = int(rng.integers(0, 7))
    src, tgt = _fuzz_cloud(kind, rng, B, M)
    r2 = rng.random()
    if r2 < 0.05: src, tgt = src * 1e-18, tgt * 1e-18      # float32 products underflow: everything passes the filter, the float64 path decides
    elif r2 < 0.08: src, tgt = src * 1e25, tgt * 1e25      # … overflow
    if rng.random() < 0.2:      # a far outlier among the targets (E of the float32 frame grows: the bound loosens, nothing may be lost)
        tgt[int(rng.integers(0, M))] += 1e6
    s = _hip_solver(hip, init, trace=False, iterations=1, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=False)
    s.set_option("knn", "brute"); s.set_option("brute_qb", qb)
    s.add_cloud(src, tgt, init); s.stein_align()
    oi, od = orc.knn_topk(src, tgt, K)
    ok = np.array_equal(s.get_candidates().astype(np.int64), oi) and np.array_equal(s.get_candidate_dist2(), od)
    if not ok:
        bad += 1
        print("FAIL seed %d kind %s B %d M %d K %d qb %d" % (seed, kind, B, M, K, qb), flush=True)
    if c % 100 == 99:
        print("... %d cases, %d failures, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("soak (brute-force stage A): %d cases, %d failures" % (n_cases, bad))
sys.exit(1 if bad else 0)

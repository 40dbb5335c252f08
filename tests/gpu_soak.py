"""Soak run (not a pytest file; run on the GPU box): many random structure / size / K / particle-count draws of the fuzz test
in tests/test_gpu_parity.py (same generator, more seeds, larger sizes incl. Morton-tile stage A and the chunk arena): stage-A
lists bit-exact, first-iteration correspondences bit-exact, sums to 1e-9, bandwidth to 1e-12.  Prints one line per failure
and a summary; exit code 1 on any failure.   python tests/gpu_soak.py [n_cases] [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from test_gpu_parity import _fuzz_cloud, _hip_solver
hip = g.load_package()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as orc
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
t0 = time.time()
for c in range(n_cases):
    seed = seed0 + c
    rng = np.random.default_rng(seed)
    kind = ["grid", "dups", "far", "tiny", "aniso"][seed % 5]
    big = rng.random() < 0.4                      # Morton tiles need >= 8192 padded targets
    B = int(rng.integers(200, 4000)); M = int(rng.integers(9000, 40000) if big else rng.integers(600, 9000))
    if os.environ.get("SOAK_BIG"):               # larger draws: Morton-tile stage A, several particle groups per wave, 256-particle workgroups
        B = int(rng.integers(8000, 40000)); M = int(rng.integers(40000, 300000))
    K = int(rng.choice([1, 5, 16, 17, 50, 96, 97, 100, 128]))
    P = int(rng.choice([1, 2, 9, 16, 30, 33, 64, 96, 128, 130, 200]))
    if os.environ.get("SOAK_BIG"):
        P = int(rng.choice([16, 32, 64, 100, 128, 200, 256, 512]))
    src, tgt = _fuzz_cloud(kind, rng, B, M)
    scale = 1e-3 if kind == "tiny" else 1.0
    init = hip.scans.make_particles(P, seed=seed + 1) * (0.2 * scale)
    cfg = dict(iterations=1, lr=1.0, max_dist=(1.0 if kind != "tiny" else 1e-6), knn_count=K, svn_full_grad=bool(seed & 1))
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg)
    if seed % 2:
        s.set_option("knn", "tiles")        # every other case: the Morton-tile stage A where the size would choose the brute-force kernel
    if seed % 4 >= 2:
        s.set_option("chain", "general")    # and half of them the general launch chain
    s.add_cloud(src, tgt, init); s.stein_align()
    tr = s.get_trace()
    ok = (np.array_equal(s.get_candidates().astype(np.int64), o.candidates()) and np.array_equal(s.get_candidate_dist2(), o.candidate_dist2())
          and np.array_equal(tr["corr"][:1], tro["corr"][:1])
          and np.allclose(tr["H"][:1], tro["H"][:1], rtol=1e-9, atol=1e-9 * scale * scale)
          and (P < 2 or np.allclose(tr["h"][:1], tro["h"][:1], rtol=1e-10, atol=0, equal_nan=True)))
    if not ok:
        bad += 1
        print("FAIL seed %d kind %s B %d M %d K %d P %d" % (seed, kind, B, M, K, P), flush=True)
    if c % 20 == 19:
        print("... %d cases, %d failures, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("soak: %d cases, %d failures" % (n_cases, bad))
sys.exit(1 if bad else 0)

"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden
vectors.  Bars (BASELINE.json north_star): correspondence indices bit-exact under brute-force NN,
final SE(3) pose within 1e-4 m / 1e-4 rad.  The small cases are held to ~1e-9."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import golden_cases, load_golden, oracle_from_golden, POSE_TOL, TIGHT

pytestmark = pytest.mark.gpu


def _hip_solver(pkg, init, trace=True, **cfg):
    prm = pkg.SteinICPParam(iterations=cfg["iterations"], lr=cfg["lr"], max_dist=cfg["max_dist"],
                            check_early_stop=cfg.get("check_early_stop", False),
                            convergence_threshold=cfg.get("convergence_threshold", 1e-5), KNN_count=cfg["knn_count"],
                            SVN_full_grad=cfg.get("svn_full_grad", False), record_trace=trace)
    return pkg.SVNICP(prm, init, pkg.ParticleWeightOpt())


def _compare(s, o, tro, P, strict_pose=TIGHT):
    n = o.iterations_run()
    assert np.array_equal(s.get_candidates().astype(np.int64), o.candidates()), "stage-A indices"
    assert np.array_equal(s.get_candidate_dist2(), o.candidate_dist2()), "stage-A dist2 bits"
    tr = s.get_trace()
    assert s.get_iterations_run() == n
    assert int(s.get_runtime()[2]) == o.finish_iter()          # finish_iter_: SVN mode never updates it (SVNICP.cpp:95-101)
    assert np.array_equal(tr["corr"][:n], tro["corr"][:n]), "per-iteration correspondence positions"
    assert np.allclose(tr["H"][:n], tro["H"][:n], rtol=1e-11, atol=1e-9)
    assert np.allclose(tr["b"][:n], tro["b"][:n], rtol=1e-9, atol=1e-9)
    assert np.allclose(tr["newton"][:n], tro["newton"][:n], rtol=1e-7, atol=1e-10)
    assert np.allclose(tr["phi"][:n], tro["phi"][:n], rtol=1e-7, atol=1e-10)
    if P > 1:
        assert np.allclose(tr["h"][:n], tro["h"][:n], rtol=1e-10)
    assert np.abs(s.get_transformation() - o.get_transformation()).max() < strict_pose
    assert np.allclose(s.get_distribution(), o.get_distribution(), atol=strict_pose)
    assert np.allclose(s.get_cov_matrix(), o.get_cov_matrix(), atol=strict_pose)
    assert np.allclose(s.get_particles(), o.get_particles(), atol=strict_pose)
    assert np.array_equal(s.get_particle_weight(), o.get_particle_weight())
    assert np.allclose(s.get_particle_history(), o.get_particle_history(), atol=1e-6)


# ------------------------------------------------------------------ the matrix pipe behind the search certificate
def test_mfma_bf16x3_error_budget():
    """The exactness certificate of k_stein_search_bf16 budgets the accumulation error of v_mfma_f32_16x16x32_bf16 at
    48·u·Σ|products| — the worst case of any faithfully rounding or truncating float32 adder tree over the 21 live
    products (svn-icp_amd/csrc/stein_split.hip).  That bound is a statement about arithmetic, not about
    this chip; this gate runs the kernel's own operand construction and adversarial operand sets (half-ulp ties, terms
    just below one ulp, graded magnitudes with alternating signs, cancelling pairs, 40 binades of exponents, every rotation
    over the live K slots) on the device and fails when any result is off by more than HALF of the budget."""
    import os
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "microbench", "mfma_bf16x3_err")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(here, "microbench")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout)
    m = re.search(r"RESULT split_max=([0-9.eE+-]+) adversarial_max=([0-9.eE+-]+) split_inexact=(\d+)", r.stdout)
    assert m, r.stdout
    split_max, adv_max, inexact = float(m.group(1)), float(m.group(2)), int(m.group(3))
    assert inexact == 0, "a float32 operand was not split exactly into three bf16 pieces"
    assert split_max <= 24.0, f"accumulation error {split_max} u·Σ|products| on the kernel's operands exceeds half of the 48u budget"
    assert adv_max <= 24.0, f"accumulation error {adv_max} u·Σ|products| on adversarial operands exceeds half of the 48u budget"


# ------------------------------------------------------------------ golden fixtures (SVN mode)
@pytest.mark.parametrize("name", golden_cases("svn"))
def test_hip_reproduces_golden(hip, orc, name):
    g = load_golden(name)
    s = _hip_solver(hip, g["init"], **g["cfg"])
    s.add_cloud(g["src"], g["tgt"], g["init"])
    s.set_initial_mean((g["R0"], g["t0"]))
    assert s.stein_align() == int(g["state"])
    n = int(g["iters_run"])
    assert np.array_equal(s.get_candidates(), g["cand_idx"])
    tr = s.get_trace()
    assert (tr["corr"][:n] != g["corr"]).mean() <= 1e-4
    assert np.allclose(tr["phi"][:n], g["phi"], atol=TIGHT)
    assert np.allclose(tr["H"][:n], g["H"], rtol=1e-11, atol=1e-9)
    assert np.abs(s.get_transformation()[:3] - g["mean"][:3]).max() < POSE_TOL  # the stated bar …
    assert np.abs(s.get_transformation()[3:] - g["mean"][3:]).max() < POSE_TOL
    assert np.allclose(s.get_transformation(), g["mean"], atol=TIGHT)            # … and the tight one
    assert np.allclose(s.get_cov_matrix(), g["cov"], atol=TIGHT)
    assert np.allclose(s.get_distribution(), g["var"], atol=TIGHT)
    assert np.allclose(s.get_particles(), g["particles"], atol=TIGHT)
    assert np.array_equal(s.get_particle_weight(), g["weights"])
    assert np.allclose(s.get_particle_history(), g["history"], atol=1e-6)
    # and against the oracle run on the same inputs
    o = oracle_from_golden(orc, g)
    tro = o.enable_trace(); o.stein_align()
    _compare(s, o, tro, g["init"].shape[1])


# ------------------------------------------------------------------ SVGD-ICP mode (first-order sibling)
def _hip_svgd(pkg, init, cfg, trace=True):
    prm = pkg.SteinICPParam(iterations=cfg["iterations"], lr=cfg["lr"], max_dist=cfg["max_dist"],
                            check_early_stop=cfg["check_early_stop"], convergence_threshold=cfg["convergence_threshold"],
                            KNN_count=cfg["knn_count"], optimizer=cfg["optimizer"], record_trace=trace)
    return pkg.SVGDICP(prm, init)


@pytest.mark.parametrize("name", golden_cases("svgd"))
def test_hip_svgd_reproduces_golden_and_oracle(hip, orc, name):
    """SVGD-ICP (SVGDICP.cpp:66-140) with each torch optimizer: golden vectors + oracle."""
    g = load_golden(name)
    s = _hip_svgd(hip, g["init"], g["cfg"])
    s.add_cloud(g["src"], g["tgt"], g["init"]); s.set_initial_mean((g["R0"], g["t0"]))
    assert s.stein_align() == int(g["state"])
    n = int(g["iters_run"])
    assert np.array_equal(s.get_candidates(), g["cand_idx"])
    tr = s.get_trace()
    assert (tr["corr"][:n] != g["corr"]).mean() <= 1e-4
    assert np.allclose(tr["newton"][:n], g["newton"], rtol=1e-9, atol=1e-9)      # sgd gradient
    assert np.allclose(tr["phi"][:n], g["phi"], rtol=1e-9, atol=1e-9)
    if g["init"].shape[1] > 1:
        assert np.allclose(tr["h"][:n], g["h"], rtol=1e-10)
    for got, want in ((s.get_transformation(), g["mean"]), (s.get_distribution(), g["var"]), (s.get_cov_matrix(), g["cov"]),
                      (s.get_particles(), g["particles"])):
        assert np.allclose(got, want, rtol=0, atol=TIGHT)
    assert np.array_equal(s.get_particle_weight(), g["weights"])
    assert np.allclose(s.get_particle_history(), g["history"], atol=1e-6)
    o = oracle_from_golden(orc, g); o.stein_align()
    assert int(s.get_runtime()[2]) == o.finish_iter() and s.get_iterations_run() == o.iterations_run()
    assert np.allclose(s.get_particles(), o.get_particles(), rtol=0, atol=TIGHT)


def test_finish_iter_on_async_and_split_phase_paths(hip, orc):
    """finish_iter_ (svnicp_get_runtime()[2], Runtime.msg) after an SVGD-mode early stop must not depend on which entry
    point ran the registration: the blocking svnicp_align, svnicp_align_async + svnicp_synchronize, and the split-phase
    sequence of the sharded driver all report epoch + 1 (SVGDICP.cpp:128)."""
    from svnicp_amd.sharded import ShardedSVGDICP
    P, B, M, I = 12, 600, 3000, 8
    src, tgt = hip.scans.random_clouds(B, M, seed=41)
    init = hip.scans.make_particles(P, seed=41) * 0.3
    cfg = dict(iterations=I, lr=0.01, max_dist=1.0, check_early_stop=True, convergence_threshold=0.05, knn_count=16, optimizer="Adam")
    o = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **cfg); o.add_cloud(src, tgt, init); o.stein_align()
    assert o.iterations_run() < I, "the early stop must fire for this test to mean anything"
    fi = o.finish_iter()
    prm = hip.SteinICPParam(iterations=I, lr=0.01, max_dist=1.0, check_early_stop=True, convergence_threshold=0.05, KNN_count=16,
                            optimizer="Adam")
    a = hip.SVGDICP(prm, init); a.add_cloud(src, tgt, init); a.stein_align()
    assert int(a.get_runtime()[2]) == fi
    b = hip.SVGDICP(prm, init); b.add_cloud(src, tgt, init); b.stein_align_async(); b.synchronize()
    assert int(b.get_runtime()[2]) == fi and b.get_iterations_run() == o.iterations_run()
    c = ShardedSVGDICP(prm, init, device_index=0); c.add_cloud(src, tgt, init); c.set_initial_mean(np.eye(4)); c.stein_align()
    assert int(c.get_runtime()[2]) == fi and c.get_iterations_run() == o.iterations_run()
    assert np.allclose(c.get_particles(), o.get_particles(), rtol=0, atol=TIGHT)
    # no stop: the constructor's value stays (SVGDICP.cpp:42)
    prm2 = hip.SteinICPParam(iterations=3, lr=0.01, max_dist=1.0, check_early_stop=True, convergence_threshold=1e-9, KNN_count=16,
                             optimizer="Adam")
    o2 = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **dict(cfg, iterations=3, convergence_threshold=1e-9))
    o2.add_cloud(src, tgt, init); o2.stein_align()
    assert o2.iterations_run() == 3
    e = hip.SVGDICP(prm2, init); e.add_cloud(src, tgt, init); e.stein_align()
    assert int(e.get_runtime()[2]) == o2.finish_iter()   # constructor value: no stop


def test_hip_svgd_stale_pose_quirk_and_no_optimizer(hip, orc):
    """Two registrations on one solver object: the RBF kernel of the second run's first epoch sees the
    FIRST run's final particles (pose_particles_ is not reset by add_cloud, SVGDICP.cpp:46-62,110)."""
    P, B, M = 9, 400, 1500
    cfg = dict(iterations=5, lr=0.01, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=12,
               optimizer="Adam")
    init1 = hip.scans.make_particles(P, seed=3) * 0.3
    init2 = hip.scans.make_particles(P, seed=4) * 0.3
    s = _hip_svgd(hip, init1, cfg, trace=False)
    o = orc.Solver(init1, mode=orc.MODE_SVGD, svn_full_grad=False, **cfg)
    for seed, init in ((31, init1), (32, init2)):
        src, tgt = hip.scans.random_clouds(B, M, seed=seed)
        s.add_cloud(src, tgt, init); s.stein_align()
        o.add_cloud(src, tgt, init); o.stein_align()
        assert np.allclose(s.get_particles(), o.get_particles(), rtol=0, atol=TIGHT)
        assert np.allclose(s.get_cov_matrix(), o.get_cov_matrix(), rtol=0, atol=TIGHT)
    bad = hip.SVGDICP(hip.SteinICPParam(iterations=3, KNN_count=5, optimizer="LBFGS"), init1)
    src, tgt = hip.scans.random_clouds(100, 300, seed=1)
    bad.add_cloud(src, tgt, init1)
    assert bad.stein_align() == hip.SteinICPState.NO_OPTIMIZER     # SVGDICP.cpp:73-75,166-169


# ------------------------------------------------------------------ seeded cases vs the oracle
CASES = [
    # P, B, M, K, I, full, early_stop, thr, max_dist, lr
    (1, 256, 1000, 7, 8, False, False, 1e-5, 1.0, 1.0),      # plain ICP, one particle
    (1, 1, 50, 1, 3, False, False, 1e-5, 1.0, 1.0),          # single source point, K = 1
    (3, 65, 130, 1, 4, True, False, 1e-5, 1.0, 1.0),         # ragged sizes, K = 1
    (4, 300, 1000, 10, 8, True, False, 1e-5, 1.0, 0.5),
    (8, 512, 2048, 32, 10, False, False, 1e-5, 0.05, 1.0),   # many rows masked by point_filter
    (8, 512, 2048, 32, 1, False, False, 1e-5, 1e-9, 1.0),    # every row masked: H = B·I₃ ⊕ 1e-6 (degenerate: 1 iteration)
    (8, 512, 2048, 32, 30, True, True, 2e-2, 1.0, 1.0),      # early stop fires
    (33, 200, 150, 100, 5, False, False, 1e-5, 1.0, 1.0),    # K close to M
    (5, 100, 40, 64, 3, False, False, 1e-5, 1.0, 1.0),       # M < K: padded candidates (idx 0)
    (64, 256, 500, 16, 6, True, False, 1e-5, 1.0, 0.5),
    (128, 2048, 8192, 100, 5, False, False, 1e-5, 1.0, 1.0), # two particle waves per workgroup
    (130, 1000, 5000, 100, 4, True, False, 1e-5, 1.0, 1.0),  # particle count not a multiple of 64; smallest sizes on the workgroup-parallel Stein step (P > 128)
    (300, 700, 3000, 24, 3, False, False, 1e-5, 1.0, 1.0),   # two particle groups; workgroup-parallel Stein step (P > 256)
    (257, 300, 1000, 8, 3, True, False, 1e-5, 1.0, 0.5),     # workgroup-parallel step, full SVN branch
    (600, 400, 2000, 16, 4, False, True, 5e-3, 1.0, 1.0),    # P = 600, early-stop compare on the ordered norm sum
    (1030, 260, 900, 8, 2, False, False, 1e-5, 1.0, 1.0),    # P > 1024 (8-GPU weak-scaling particle count and beyond)
    (32, 4096, 8192, 100, 6, False, False, 1e-5, 1.0, 1.0),
    (16, 3000, 9000, 200, 4, False, False, 1e-5, 1.0, 1.0),  # K > 128: larger candidate pool
]


@pytest.mark.parametrize("P,B,M,K,I,full,es,thr,md,lr", CASES)
def test_hip_vs_oracle(hip, orc, P, B, M, K, I, full, es, thr, md, lr):
    src, tgt = hip.scans.random_clouds(B, M, seed=P + B)
    init = hip.scans.make_particles(P, seed=P) * 0.3
    R0, t0 = hip.scans.rot_zyx(0.001, 0.002, -0.001), np.array([0.01, -0.02, 0.005])
    cfg = dict(iterations=I, lr=lr, max_dist=md, check_early_stop=es, convergence_threshold=thr, knn_count=K,
               svn_full_grad=full)
    o = orc.Solver(init, **cfg)
    o.add_cloud(src, tgt, init); o.set_initial_mean(R0, t0)
    tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg)
    s.add_cloud(src, tgt, init); s.set_initial_mean((R0, t0))
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    _compare(s, o, tro, P)



# ------------------------------------------------------------------ correspondence = full
@pytest.mark.parametrize("P,B,M,K,I,full,md,knn", [
    (4, 300, 1000, 10, 5, False, 1.0, None),       # few particles
    (20, 777, 5000, 16, 4, True, 1.0, None),       # ragged source, full SVN branch
    (70, 512, 2048, 32, 3, False, 0.05, None),     # point_filter masks most rows
    (12, 400, 3000, 8, 3, False, 1.0, "v1"),       # streaming stage A carries the K = 1 searches
    (130, 260, 900, 8, 2, False, 1.0, None),       # workgroup-parallel Stein step
    (3, 65, 40, 64, 3, False, 1.0, None),          # M < K
])
def test_full_correspondence_mode(hip, orc, P, B, M, K, I, full, md, knn):
    """The reference's get_correspondence (SVGDICP.cpp:274-298): every particle searches the whole target, K = 1.
    Trace corr holds target indices (bit-exact against the oracle's brute force); pose to 1e-9."""
    src, tgt = hip.scans.random_clouds(B, M, seed=3 * P + B)
    init = hip.scans.make_particles(P, seed=P) * 0.3
    R0, t0 = hip.scans.rot_zyx(0.001, 0.002, -0.001), np.array([0.01, -0.02, 0.005])
    cfg = dict(iterations=I, lr=1.0, max_dist=md, check_early_stop=False, convergence_threshold=1e-5, knn_count=K,
               svn_full_grad=full)
    o = orc.Solver(init, **cfg)
    o.set_correspondence_full(True)
    o.add_cloud(src, tgt, init); o.set_initial_mean(R0, t0)
    tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg)
    s.set_option("correspondence", "full")
    if knn:
        s.set_option("knn", knn)
    s.add_cloud(src, tgt, init); s.set_initial_mean((R0, t0))
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    assert tro["corr"].max() >= K or M <= K          # really target indices, not positions in the K list
    _compare(s, o, tro, P)
    # and the fast mode on the same inputs differs somewhere (K nearest of the INITIAL pose are not always enough)
    s.set_option("correspondence", "fast")
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS


def test_full_correspondence_mode_refusals(hip):
    src, tgt = hip.scans.random_clouds(300, 2000, seed=5)
    init = hip.scans.make_particles(4, seed=1) * 0.1
    s = _hip_solver(hip, init, iterations=2, lr=1.0, max_dist=1.0, knn_count=200)      # K > 128: seeded-scan stage A
    s.set_option("correspondence", "full")
    s.add_cloud(src, tgt, init)
    with pytest.raises(Exception):
        s.stein_align()
    with pytest.raises(Exception):
        s.set_option("correspondence", "sometimes")

# ------------------------------------------------------------------ stage A variants
@pytest.mark.parametrize("B,M,K", [(700, 20000, 7), (300, 9000, 1), (1000, 16384, 128), (513, 8192, 100), (64, 40000, 33),
                                   (900, 12000, 150)])
def test_stage_a_variants_bit_exact(hip, orc, B, M, K):
    """The four stage-A kernels — streaming (knn_topk.hip, option knn=v1), seeded f32 pre-filter
    (knn_scan.hip, v2), Morton-tile pruning (knn_tiles.hip, K <= 128) and the one-launch brute force for small
    registrations (knn_brute.hip, the default up to 2^28 point pairs and K <= 128) — must all give indices and dist²
    bit-identical to the oracle's f64 brute force."""
    src, tgt = hip.scans.random_clouds(B, M, seed=B + K, extent=40.0)
    src = src + np.array([100.0, -50.0, 3.0])      # large coordinates: the filter slack must cover them
    tgt = tgt + np.array([100.0, -50.0, 3.0])
    init = np.zeros((6, 2)); init[0, 1] = 0.01
    cfg = dict(iterations=1, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=False)
    oi, od = orc.knn_topk(src, tgt, K)
    fbs = {}
    for variant in ("v1", "v2", "tiles", "brute", "default"):
        if variant == "brute" and K > 128:
            continue
        s = _hip_solver(hip, init, trace=False, **cfg); s.add_cloud(src, tgt, init)
        if variant != "default":
            s.set_option("knn", variant)
        s.set_initial_mean((hip.scans.rot_zyx(0.01, -0.02, 0.03), np.array([0.3, -0.2, 0.1])))
        s.stein_align()
        fbs[variant] = s.get_knn_fallbacks()
        q = orc.transform(src, hip.scans.rot_zyx(0.01, -0.02, 0.03), np.array([0.3, -0.2, 0.1]))
        oi, od = orc.knn_topk(q, tgt, K)
        assert np.array_equal(s.get_candidates().astype(np.int64), oi), variant
        assert np.array_equal(s.get_candidate_dist2(), od), variant
    assert fbs["v1"] == -1
    assert 0 <= fbs["v2"] <= max(2, B // 100)       # the probabilistic seed is good: (almost) nothing falls back
    if K > 128:
        assert fbs["default"] == fbs["v2"] and fbs["tiles"] == fbs["v2"]   # neither knn_tiles nor knn_brute applies: the v2 kernel
    else:
        assert 0 <= fbs["tiles"] <= max(2, B // 100)    # guaranteed seed + in-kernel overflow recovery
        assert fbs["brute"] == 0 and fbs["default"] == 0    # these sizes default to the brute-force kernel, which never falls back


@pytest.mark.parametrize("sliced_max,K", [(None, 50), ("0", 50), ("100", 50), (None, 128), (None, 7)])
def test_stage_a_fallback_on_pool_overflow(hip, orc, sliced_max, K):
    """Clustered duplicates: thousands of targets at exactly the same distance overflow the
    candidate pool of every query; all of them must be redone by the streaming fallback and still
    come out bit-exact (ties broken by lowest index).  Both fallback regimes: target-sliced scan +
    merge (few failures, default here) and one wave per two queries (option fallback_sliced_max
    below the failure count)."""
    rng = np.random.default_rng(3)
    centers = rng.normal(size=(6, 3)) * 5
    tgt = np.repeat(centers, 2000, axis=0).astype(np.float32).astype(np.float64)   # M = 12000, 2000 duplicates each
    tgt = tgt[rng.permutation(tgt.shape[0])]
    src = (centers[rng.integers(0, 6, 300)] + rng.normal(size=(300, 3)) * 0.1)
    init = np.zeros((6, 1))
    cfg = dict(iterations=1, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=False)
    s = _hip_solver(hip, init, trace=False, **cfg); s.add_cloud(src, tgt, init)
    s.set_option("knn", "tiles")          # the default at this size is the brute-force kernel (next test)
    if sliced_max is not None:
        s.set_option("fallback_sliced_max", sliced_max)
    s.stein_align()
    assert s.get_knn_fallbacks() == 300
    oi, od = orc.knn_topk(src, tgt, K)
    assert np.array_equal(s.get_candidates().astype(np.int64), oi)
    assert np.array_equal(s.get_candidate_dist2(), od)


@pytest.mark.parametrize("B,M,K", [(1, 1, 1), (5, 3, 7), (64, 130, 128), (9, 1025, 16), (300, 2047, 100), (2, 1023, 1), (1000, 1024, 33)])
def test_stage_a_brute_force_small_shapes(hip, orc, B, M, K):
    """knn_brute.hip at the edges of its launch geometry: fewer targets than threads (most threads hold no minimum), one
    query, one target, K above the target count (zero padding), block counts that are not a multiple of four queries."""
    src, tgt = hip.scans.random_clouds(B, M, seed=B * 7 + M + K, extent=10.0)
    init = np.zeros((6, 1))
    s = _hip_solver(hip, init, trace=False, iterations=1, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=False)
    s.add_cloud(src, tgt, init); s.stein_align()
    assert s.get_knn_fallbacks() == 0          # the brute-force kernel (it reports no fallbacks; the other kernels would report -1 or a count)
    oi, od = orc.knn_topk(src, tgt, K)
    assert np.array_equal(s.get_candidates().astype(np.int64), oi)
    assert np.array_equal(s.get_candidate_dist2(), od)


@pytest.mark.parametrize("K", [1, 7, 50, 128])
def test_stage_a_brute_force_exact_ties(hip, orc, K):
    """knn_brute.hip's general path: 2000 targets at exactly the same distance put far more than 256 entries below the
    K-th bin, so the selection goes through the full 64 bits of d² and then through the index bits of the exact ties
    (lowest index wins, as everywhere); a query ON a duplicated target has d² = 0 (below the key window); fewer targets
    than K pad with zeros."""
    rng = np.random.default_rng(3)
    centers = rng.normal(size=(6, 3)) * 5
    tgt = np.repeat(centers, 2000, axis=0).astype(np.float32).astype(np.float64)   # M = 12000, 2000 duplicates each
    tgt = tgt[rng.permutation(tgt.shape[0])]
    src = (centers[rng.integers(0, 6, 301)] + rng.normal(size=(301, 3)) * 0.1)
    src[:6] = tgt[:6]                                   # d² = 0 against 2000 targets each
    init = np.zeros((6, 1))
    cfg = dict(iterations=1, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=False)
    s = _hip_solver(hip, init, trace=False, **cfg); s.add_cloud(src, tgt, init)
    s.stein_align()
    assert s.get_knn_fallbacks() == 0
    oi, od = orc.knn_topk(src, tgt, K)
    assert np.array_equal(s.get_candidates().astype(np.int64), oi)
    assert np.array_equal(s.get_candidate_dist2(), od)
    # fewer targets than K
    small = tgt[: max(2, K // 2)].copy()
    s = _hip_solver(hip, init, trace=False, **cfg); s.add_cloud(src[:37], small, init)
    s.stein_align()
    oi, od = orc.knn_topk(src[:37], small, K)
    assert np.array_equal(s.get_candidates().astype(np.int64), oi)
    assert np.array_equal(s.get_candidate_dist2(), od)


@pytest.mark.parametrize("qb", [1, 2, 3, 4, 5, 6])
def test_stage_a_brute_force_queries_per_workgroup(hip, orc, qb):
    """knn_brute.hip is instantiated for 1..6 queries per workgroup (launch_knn_brute picks the one with the fewest
    rounds of workgroups over the CUs); every instantiation gives the oracle's rows — ordinary clouds (pass A / pass B),
    2000-fold duplicates (the histogram path incl. index bits), a query count that is no multiple of qb, M < K."""
    rng = np.random.default_rng(11 + qb)
    init = np.zeros((6, 1))
    K = 100
    cfg = dict(iterations=1, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=False)
    src, tgt = hip.scans.random_clouds(8 * 37 + 3, 5000, seed=40 + qb, extent=10.0)
    centers = rng.normal(size=(4, 3)) * 5
    dup = np.repeat(centers, 2000, axis=0).astype(np.float32).astype(np.float64)
    dup = dup[rng.permutation(dup.shape[0])]
    qdup = centers[rng.integers(0, 4, 61)] + rng.normal(size=(61, 3)) * 0.1
    qdup[:4] = dup[:4]
    for s_, t_ in ((src, tgt), (qdup, dup), (src[:13], tgt[:40])):
        s = _hip_solver(hip, init, trace=False, **cfg)
        s.set_option("brute_qb", qb)
        s.add_cloud(s_, t_, init); s.stein_align()
        assert s.get_knn_fallbacks() == 0
        oi, od = orc.knn_topk(s_, t_, K)
        assert np.array_equal(s.get_candidates().astype(np.int64), oi)
        assert np.array_equal(s.get_candidate_dist2(), od)


def test_stage_a_survivor_arena_exhaustion_is_survivable(hip, orc):
    """Every query sits next to 2000 duplicated targets: each needs three 512-slot chunks beyond its own pool row, 40 000
    queries ask for 120 000 chunks and the arena holds 32 768.  Queries that get no chunk are marked failed (no hang, no
    write outside the arena) and redone exactly by the fallback together with the rest (more than 512 exact ties each)."""
    rng = np.random.default_rng(5)
    centers = rng.normal(size=(6, 3)) * 5
    tgt = np.repeat(centers, 2000, axis=0).astype(np.float32).astype(np.float64)
    tgt = tgt[rng.permutation(tgt.shape[0])]
    B = 40000
    src = centers[rng.integers(0, 6, B)] + rng.normal(size=(B, 3)) * 0.1
    init = np.zeros((6, 1))
    s = _hip_solver(hip, init, trace=False, iterations=1, lr=1.0, max_dist=1.0, knn_count=8, svn_full_grad=False)
    s.add_cloud(src, tgt, init)
    s.stein_align()
    assert s.get_knn_fallbacks() == B
    oi, od = orc.knn_topk(src, tgt, 8)
    assert np.array_equal(s.get_candidates().astype(np.int64), oi)
    assert np.array_equal(s.get_candidate_dist2(), od)


@pytest.mark.parametrize("P,full,K", [(70, False, 60), (20, True, 60), (128, False, 100), (9, False, 128), (40, False, 17),
                                      (33, False, 1), (65, False, 80), (17, False, 97), (30, False, 112)])
def test_stage_b_f32_search_is_bit_identical_to_f64_kernel(hip, P, full, K):
    """stein_iter.hip / stein_split.hip: the float32 searches (VALU, fused with the accumulation: option accum=valu; bf16x3
    matrix pipe + separate accumulation kernel: accum=split, the default) must reproduce the float64 baseline kernel
    (accum=f64) — same correspondences, same sums."""
    src, tgt = hip.scans.random_clouds(6000, 20000, seed=77, extent=30.0)
    src = src + np.array([80.0, -40.0, 2.0]); tgt = tgt + np.array([80.0, -40.0, 2.0])
    init = hip.scans.make_particles(P, seed=P) * 0.5
    cfg = dict(iterations=6, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=full)
    out = {}
    for mode in ("f64", "valu", "split"):
        s = _hip_solver(hip, init, trace=True, **cfg); s.add_cloud(src, tgt, init)
        s.set_option("accum", mode)
        s.stein_align()
        out[mode] = (s.get_particles(), s.get_trace()["corr"], s.get_trace()["H"], s.get_ambiguous_steps(), s.get_ambiguous_pairs())
    assert out["f64"][3] == -1 and out["valu"][3] >= 0 and out["split"][3] >= 0
    n_pairs = 6 * P * 6000
    assert 0 <= out["split"][4] < 0.05 * n_pairs or K == 1, "the matrix-pipe search should certify most pairs itself"
    for mode in ("valu", "split"):
        assert np.array_equal(out["f64"][1], out[mode][1]), mode   # correspondences: always identical
    assert np.array_equal(out["f64"][2], out["valu"][2])           # same tiling as the f64 kernel: same summation order
    assert np.array_equal(out["f64"][0], out["valu"][0])
    # the split variant partitions the source points differently (no LDS tiles) and its accumulate kernel forms Ts, d² and the
    # weight with fused operations and hand-refined rsq / rcp (stein_split.hip): same terms to a few 2^-52, other order.
    # The RAW sums must agree to 1e-12 — they are the H entries that finalize_Hb copies (Σw, ±Σw·s, −Σw·s_i·s_j, i != j);
    # the diagonal of the rotation block is a difference of sums (tr − xx: cancellation by up to |s|²/|s_perp|²) and gets the
    # oracle comparisons' 1e-11
    Hs, Hf = out["split"][2].reshape(-1, 6, 6), out["f64"][2].reshape(-1, 6, 6)
    raw = [(0, 0), (0, 4), (0, 5), (1, 5), (3, 4), (3, 5), (4, 5)]
    for (i, j) in raw:
        np.testing.assert_allclose(Hs[:, i, j], Hf[:, i, j], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(Hs, Hf, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(out["split"][0], out["f64"][0], rtol=0, atol=1e-9)


@pytest.mark.parametrize("P,B,M,K", [(8, 700, 20000, 16), (40, 2048, 16384, 100)])
def test_map_frame_coordinates_far_from_the_origin(hip, orc, P, B, M, K):
    """The local map lives in the map frame: kilometres from the origin after a long drive.  The float32 pre-filters of
    both stages work on differences (stage A) and in the local frame of a point's first candidate (stage B), so their
    error bounds grow with u*|coordinate|, not with its square; indices stay bit-exact, poses to 1e-9."""
    src, tgt = hip.scans.random_clouds(B, M, seed=B + K)
    off = np.array([4321.0, -8765.5, 120.25])
    tgt = tgt + off
    init = hip.scans.make_particles(P, seed=P) * 0.3
    R0, t0 = hip.scans.rot_zyx(0.001, 0.002, -0.001), np.array([0.01, -0.02, 0.005]) + off   # initial guess carries the offset
    cfg = dict(iterations=5, lr=1.0, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=K,
               svn_full_grad=False)
    o = orc.Solver(init, **cfg)
    o.add_cloud(src, tgt, init); o.set_initial_mean(R0, t0)
    tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg)
    s.add_cloud(src, tgt, init); s.set_initial_mean((R0, t0))
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    _compare(s, o, tro, P)


@pytest.mark.parametrize("P,same", [(6, 5), (6, 3), (9, 9), (33, 20), (128, 100), (200, 150), (301, 10), (257, 257)])
def test_bandwidth_median_with_coincident_particles(hip, orc, P, same):
    """`same` of the P particles start at exactly the same pose: their pair distances are exact zeros, as is the diagonal.
    The lower median of the P x P matrix (SVNICP.cpp:257-262) is then zero or the smallest positive distance depending on
    the count — the symmetric pair pass of k_upd_front (pairs i < j with weight 2, the diagonal as P zeros) and the
    histogram chain above 128 particles must land on
    the same entry as the oracle's full sort."""
    src, tgt = hip.scans.random_clouds(400, 1500, seed=P + same)
    init = hip.scans.make_particles(P, seed=P) * 0.3
    init[:, :same] = init[:, :1]
    cfg = dict(iterations=1, lr=1.0, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=10,
               svn_full_grad=False)
    o = orc.Solver(init, **cfg)
    o.add_cloud(src, tgt, init)
    tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg)
    s.add_cloud(src, tgt, init)
    s.stein_align()
    h_gpu, h_orc = s.get_trace()["h"][0], tro["h"][0]
    assert (h_gpu == 0.0) == (h_orc == 0.0)
    assert h_gpu == h_orc or abs(h_gpu - h_orc) <= 1e-12 * abs(h_orc)
    zeros = same * same + (P - same)
    assert (h_orc == 0.0) == (zeros > (P * P - 1) // 2)


def test_exact_ties_lowest_index_wins(hip, orc):
    """Integer-grid clouds: distances are exact and massively tied in stage A and stage B; the HIP
    path must break every tie like the reference CPU KNN (lowest index / first position)."""
    rng = np.random.default_rng(0)
    tgt = rng.integers(-5, 6, size=(3000, 3)).astype(np.float64)
    src = rng.integers(-5, 6, size=(500, 3)).astype(np.float64)
    init = np.zeros((6, 4)); init[0] = [0.0, 1.0, -2.0, 0.5]   # exact translations: stage-B ties stay exact
    cfg = dict(iterations=1, lr=1.0, max_dist=100.0, knn_count=40, svn_full_grad=False)
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg); s.add_cloud(src, tgt, init); s.stein_align()
    ci = s.get_candidates().astype(np.int64)
    assert np.array_equal(ci, o.candidates())
    d2 = s.get_candidate_dist2()
    tied = np.diff(d2, axis=1) == 0
    assert tied.mean() > 0.3 and np.all(np.diff(ci, axis=1)[tied] > 0)
    assert np.array_equal(s.get_trace()["corr"][:1], tro["corr"][:1])


def test_device_pointer_inputs_and_determinism(hip):
    import torch
    src, tgt = hip.scans.random_clouds(3000, 9000, seed=11)
    init = hip.scans.make_particles(40, seed=11) * 0.3
    cfg = dict(iterations=6, lr=1.0, max_dist=1.0, knn_count=50, svn_full_grad=True)
    a = _hip_solver(hip, init, trace=False, **cfg); a.add_cloud(src, tgt, init); a.stein_align()
    b = _hip_solver(hip, init, trace=False, **cfg)
    b.add_cloud(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), init); b.stein_align()
    assert np.array_equal(a.get_particles(), b.get_particles())      # host vs device inputs: same bits
    a.add_cloud(src, tgt, init); a.stein_align()                      # context re-use, run-to-run determinism
    assert np.array_equal(a.get_particles(), b.get_particles())
    assert np.array_equal(a.get_cov_matrix(), b.get_cov_matrix())


def test_context_reuse_with_changing_sizes(hip, orc):
    init = hip.scans.make_particles(12, seed=2) * 0.3
    cfg = dict(iterations=4, lr=1.0, max_dist=1.0, knn_count=20, svn_full_grad=False)
    s = _hip_solver(hip, init, trace=False, **cfg)
    for (B, M, K, md) in [(500, 2000, 20, 1.0), (1500, 700, 9, 0.5), (64, 64, 64, 1.0)]:
        src, tgt = hip.scans.random_clouds(B, M, seed=B)
        s.set_k(K); s.set_threshold(md)
        s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4)); s.stein_align()
        o = orc.Solver(init, **dict(cfg, knn_count=K, max_dist=md)); o.add_cloud(src, tgt, init); o.stein_align()
        assert np.array_equal(s.get_candidates().astype(np.int64), o.candidates())
        assert np.abs(s.get_transformation() - o.get_transformation()).max() < TIGHT


def test_knn_count_beyond_the_lds_budget_is_refused(hip, orc):
    """K <= 128 runs on the matrix pipe; larger K on the LDS-tile VALU search, whose smallest tile has to fit the 160 KB of
    LDS: K = 256 still runs (and matches the oracle), K = 1000 is refused in stein_align with a clear message instead of
    failing at a kernel launch after stage A has already run."""
    src, tgt = hip.scans.random_clouds(600, 3000, seed=2)
    init = hip.scans.make_particles(6, seed=2) * 0.3
    cfg = dict(iterations=2, lr=1.0, max_dist=1.0, knn_count=256, svn_full_grad=False)
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); o.stein_align()
    s = _hip_solver(hip, init, trace=False, **cfg); s.add_cloud(src, tgt, init); s.stein_align()
    assert np.array_equal(s.get_candidates().astype(np.int64), o.candidates())
    assert np.abs(s.get_transformation() - o.get_transformation()).max() < TIGHT
    s.set_k(1000)
    s.add_cloud(src, tgt, init)
    with pytest.raises(hip.SvnIcpError, match="LDS"):
        s.stein_align()


def test_error_paths(hip):
    s = hip.SVNICP(hip.SteinICPParam(iterations=2, KNN_count=4), np.zeros((6, 2)))
    with pytest.raises(hip.SvnIcpError):
        s.stein_align()                       # no clouds yet
    with pytest.raises(hip.SvnIcpError):
        s.get_transformation()                # no result yet
    with pytest.raises(hip.SvnIcpError):
        hip.SVNICP(hip.SteinICPParam(iterations=2, KNN_count=0), np.zeros((6, 2)))


@pytest.mark.parametrize("es", [False, True])
def test_single_particle_fused_iteration(hip, orc, es):
    """One particle (plain ICP through the solver, BASELINE C1's shape): by default the accumulate kernel's last workgroup
    reduces the partial sums and runs the Stein step, one launch per iteration; option single=split keeps the three
    launches (accumulate, k_reduce_partials, k_particle_update).  Same correspondences, same iteration count, poses equal to
    rounding (the two reductions group the workgroups' sums differently) and both equal to the oracle."""
    B, M, K, I = 3000, 9000, 40, 25
    src, tgt = hip.scans.random_clouds(B, M, seed=71, extent=20.0)
    init = np.zeros((6, 1))
    cfg = dict(iterations=I, lr=1.0, max_dist=1.0, check_early_stop=es, convergence_threshold=2e-3, knn_count=K, svn_full_grad=False)
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    out = {}
    for mode in ("fused", "split"):
        s = _hip_solver(hip, init, **cfg); s.set_option("single", mode); s.add_cloud(src, tgt, init)
        assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
        _compare(s, o, tro, 1)
        out[mode] = (s.get_particles(), s.get_trace()["corr"], s.get_iterations_run())
    assert np.array_equal(out["fused"][1], out["split"][1]) and out["fused"][2] == out["split"][2] == o.iterations_run()
    assert np.allclose(out["fused"][0], out["split"][0], rtol=0, atol=1e-12)
    if es and mode == "svn":
        assert o.iterations_run() < I


@pytest.mark.parametrize("P,full,mode", [(30, False, "svn"), (128, False, "svn"), (17, True, "svn"), (9, False, "svn"), (64, False, "svgd")])
def test_small_chain_equals_general_chain(hip, orc, P, full, mode):
    """Few (point, particle) pairs — the scan-to-map loop's sizes — run the small chain: at most 32 accumulate workgroups,
    no k_reduce_partials (the prepare lanes add the workgroups' records in block order) and the pair statistics in the
    prepare kernel's launch on the main stream.  Option chain=general keeps the general chain.  Same correspondences; poses
    equal to rounding (the sums are grouped differently) and both equal to the oracle."""
    B, M, K, I = 1100, 9000, 50, 8
    src, tgt = hip.scans.random_clouds(B, M, seed=P + 7, extent=20.0)
    init = hip.scans.make_particles(P, seed=P) * 0.2
    cfg = dict(iterations=I, lr=1.0, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=K, svn_full_grad=full)
    if mode == "svgd":
        cfg = dict(cfg, lr=0.01, optimizer="Adam"); cfg.pop("svn_full_grad")
        o = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **cfg)
        mk = lambda: _hip_svgd(hip, init, cfg)
    else:
        o = orc.Solver(init, **cfg)
        mk = lambda: _hip_solver(hip, init, **cfg)
    o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    out = {}
    for chain in ("auto", "general"):
        s = mk(); s.set_option("chain", chain); s.add_cloud(src, tgt, init)
        assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
        if mode == "svgd":
            assert np.allclose(s.get_particles(), o.get_particles(), rtol=0, atol=TIGHT)
        else:
            _compare(s, o, tro, P)
        out[chain] = (s.get_particles(), s.get_trace()["corr"])
    assert np.array_equal(out["auto"][1], out["general"][1])
    assert np.allclose(out["auto"][0], out["general"][0], rtol=0, atol=1e-10)


@pytest.mark.parametrize("P,full,mode,es", [(30, False, "svn", False), (128, False, "svn", False), (17, True, "svn", False), (9, False, "svn", True),
                                             (64, False, "svgd", False), (33, False, "svgd", True)])
def test_small_registration_persistent_kernel(hip, orc, P, full, mode, es):
    """Option chain=persistent: svnicp_align of a small registration (K = 97…100, no traces) runs ALL iterations in one
    cooperative launch (k_small_registration: the same device bodies on virtual blocks, grid barriers between the phases).
    Same arithmetic, same block partition, same order of additions as the default four launches per iteration: the
    results must be bit-identical — particles, history, iteration count — and equal to the oracle.  (Off by default: on
    this eight-XCD part the barriers cost more than the launches they replace)"""
    B, M, K, I = 1100, 9000, 100, 12
    src, tgt = hip.scans.random_clouds(B, M, seed=P + 11, extent=20.0)
    init = hip.scans.make_particles(P, seed=P) * 0.2
    cfg = dict(iterations=I, lr=1.0, max_dist=1.0, check_early_stop=es, convergence_threshold=(3e-2 if es else 1e-5), knn_count=K, svn_full_grad=full)
    if mode == "svgd":
        cfg = dict(cfg, lr=0.01, optimizer="Adam", convergence_threshold=(8e-3 if es else 1e-5)); cfg.pop("svn_full_grad")
        o = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **cfg)
        mk = lambda: _hip_svgd(hip, init, cfg, trace=False)
    else:
        o = orc.Solver(init, **cfg)
        mk = lambda: _hip_solver(hip, init, trace=False, **cfg)
    o.add_cloud(src, tgt, init); o.stein_align()
    out = {}
    for chain in ("persistent", "auto"):
        s = mk(); s.set_option("chain", chain); s.add_cloud(src, tgt, init)
        assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
        assert s.get_iterations_run() == o.iterations_run() and int(s.get_runtime()[2]) == o.finish_iter()
        assert np.allclose(s.get_particles(), o.get_particles(), rtol=0, atol=TIGHT)
        out[chain] = (s.get_particles(), s.get_particle_history(), s.get_transformation(), s.get_cov_matrix())
    for x, y in zip(out["persistent"], out["auto"]):
        assert np.array_equal(x, y)
    if es and mode == "svn":
        assert o.iterations_run() < I                      # (SVGD-ICP: the displacement threshold may or may not trip in 12 steps; equality is the point)


def test_blocking_align_follows_the_stop_flag(hip, orc):
    """The reference's shipped settings (config/geodeAlpha.yaml): 100 iterations, early stop at 5e-4, 10 particles, max_dist 3.
    The blocking svnicp_align enqueues the iterations in chunks and stops enqueuing soon after the device has stopped;
    svnicp_align_async enqueues all 100 (the rest return at once).  Same result, same iteration count, equal to the oracle."""
    P, B, M, K, I = 10, 1100, 9000, 100, 100
    src, tgt = hip.scans.random_clouds(B, M, seed=41, extent=20.0)
    init = hip.scans.make_particles(P, seed=P) * 0.2
    cfg = dict(iterations=I, lr=1.0, max_dist=3.0, check_early_stop=True, convergence_threshold=5e-3, knn_count=K, svn_full_grad=False)   # (5e-3 on these random clouds: stops at iteration 22)
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); o.stein_align()
    assert 8 < o.iterations_run() < I
    a = _hip_solver(hip, init, trace=False, **cfg); a.add_cloud(src, tgt, init)
    assert a.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    b = _hip_solver(hip, init, trace=False, **cfg); b.add_cloud(src, tgt, init); b.stein_align_async(); b.synchronize()
    for s in (a, b):
        assert s.get_iterations_run() == o.iterations_run() and int(s.get_runtime()[2]) == o.finish_iter()
        assert np.allclose(s.get_particles(), o.get_particles(), rtol=0, atol=TIGHT)
    assert np.array_equal(a.get_particles(), b.get_particles()) and np.array_equal(a.get_particle_history(), b.get_particle_history())


def test_two_particles_zero_bandwidth_goes_nan_like_the_reference(hip, orc):
    """Two particles: the lower median of the four pair distances {0, 0, d, d} is 0, the RBF bandwidth is 0 and the first
    Stein step is NaN (SVNICP.cpp:262, 218-252).  From then on the reference's masking BY MULTIPLICATION (SVGDICP.cpp:331-333)
    keeps every row of a NaN particle NaN, so H and b are NaN too — also in the kernels that serve up to eight particles,
    which mask with a branch and must poison the sums explicitly."""
    P, B, M, K, I = 2, 1100, 9000, 50, 4
    src, tgt = hip.scans.random_clouds(B, M, seed=P + 7, extent=20.0)
    init = hip.scans.make_particles(P, seed=P) * 0.2
    cfg = dict(iterations=I, lr=1.0, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=K, svn_full_grad=False)
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    for accum in ("valu", "f64"):
        s = _hip_solver(hip, init, **cfg); s.set_option("accum", accum); s.add_cloud(src, tgt, init)
        assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
        tr = s.get_trace()
        assert np.isnan(tro["H"][1:]).all() and np.isfinite(tro["H"][0]).all()
        for key in ("b", "phi"):
            assert np.array_equal(np.isnan(tr[key]), np.isnan(tro[key])), (accum, key)
            assert np.allclose(tr[key], tro[key], rtol=1e-9, atol=1e-9, equal_nan=True), (accum, key)
        # H is assembled from the 22 raw sums: its structural zeros (and the 1e-6 on the rotation block's diagonal where the
        # sums cancel) stay numbers where the reference's literal JᵀwJ gives NaN·0 = NaN; everything fed by a sum is NaN
        hn, on = np.isnan(tr["H"]), np.isnan(tro["H"])
        assert not (hn & ~on).any() and hn[1:].reshape(I - 1, P, 36)[:, :, 0].all()
        assert np.allclose(tr["H"][0], tro["H"][0], rtol=1e-11, atol=1e-9)
        # (the rotation entries are 0, not NaN: rotm_to_ypr_tensor fills them where |sin| > 1e-12 fails, SVNICP.cpp:205-213)
        assert np.array_equal(np.isnan(s.get_particles()), np.isnan(o.get_particles())) and np.isnan(s.get_particles()).any()
        assert np.allclose(s.get_particles(), o.get_particles(), equal_nan=True)


# ------------------------------------------------------------------ split-phase ABI (multi-GPU path) on one GPU
def test_split_phase_two_shards_equal_single_context(hip):
    """Two contexts on one GPU play two ranks: particle shards [0,P/2) and [P/2,P), candidate rows
    split in two; the 'all-gathers' are host copies.  Result must equal the one-shot svnicp_align."""
    import torch
    L = hip.load_library()
    P, B, M, K, I = 24, 1200, 4000, 30, 5
    src, tgt = hip.scans.random_clouds(B, M, seed=5)
    init = hip.scans.make_particles(P, seed=5) * 0.3
    cfg = dict(iterations=I, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=True)
    ref = _hip_solver(hip, init, trace=False, **cfg); ref.add_cloud(src, tgt, init); ref.stein_align()
    ranks = []
    for r in range(2):
        s = _hip_solver(hip, init, trace=False, **cfg); s.add_cloud(src, tgt, init)
        assert L.svnicp_set_shard(s.handle, r * P // 2, (r + 1) * P // 2) == 0
        assert L.svnicp_align_begin(s.handle) == 0
        assert L.svnicp_stage_candidates(s.handle, r * B // 2, (r + 1) * B // 2) == 0
        ranks.append(s)

    def view(s, fn, shape, dt):
        from svnicp_amd.sharded import _DevView
        return torch.as_tensor(_DevView(getattr(L, fn)(s.handle), shape, dt), device="cuda")
    for s in ranks:
        L.svnicp_synchronize(s.handle)
    c0, c1 = (view(s, "svnicp_candidates_devptr", (B, K), "<i4") for s in ranks)
    c0[B // 2:] = c1[B // 2:]; c1[:B // 2] = c0[:B // 2]
    torch.cuda.synchronize()
    for s in ranks:
        assert L.svnicp_build_candidate_table(s.handle) == 0
    for it in range(I):
        for s in ranks:
            assert L.svnicp_iter_accumulate(s.handle, it) == 0
            L.svnicp_synchronize(s.handle)
        r0, r1 = (view(s, "svnicp_sums_devptr", (P, 22), "<f8") for s in ranks)
        r0[P // 2:] = r1[P // 2:]; r1[:P // 2] = r0[:P // 2]
        torch.cuda.synchronize()
        for s in ranks:
            assert L.svnicp_iter_update(s.handle, it) == 0
    for s in ranks:
        assert L.svnicp_finish(s.handle) == 0
        L.svnicp_synchronize(s.handle)
        # a shard of P/2 particles uses a different lane tiling (hence summation order) than the
        # one-shot run: equal to rounding, not bitwise …
        assert np.allclose(s.get_particles(), ref.get_particles(), rtol=0, atol=1e-12)
        assert np.allclose(s.get_cov_matrix(), ref.get_cov_matrix(), rtol=0, atol=1e-12)
        assert np.array_equal(s.get_candidates(), ref.get_candidates())
    # … but the two replicas must agree bit for bit (identical inputs, identical code)
    assert np.array_equal(ranks[0].get_particles(), ranks[1].get_particles())
    assert np.array_equal(ranks[0].get_cov_matrix(), ranks[1].get_cov_matrix())


def test_sharded_driver_world1_equals_plain(hip):
    from svnicp_amd.sharded import ShardedSVNICP
    P, B, M = 10, 800, 2500
    src, tgt = hip.scans.random_clouds(B, M, seed=9)
    init = hip.scans.make_particles(P, seed=9) * 0.3
    prm = hip.SteinICPParam(iterations=4, lr=1.0, max_dist=1.0, KNN_count=16, SVN_full_grad=False)
    a = hip.SVNICP(prm, init); a.add_cloud(src, tgt, init); a.stein_align()
    b = ShardedSVNICP(prm, init, device_index=0); b.add_cloud(src, tgt, init); b.set_initial_mean(np.eye(4))
    b.stein_align()
    assert np.array_equal(a.get_particles(), b.get_particles())


def _two_rank_worker(rank, world, port, out_dir):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    import torch.distributed as dist
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svnicp_amd.sharded import ShardedSVNICP
    P, B, M = 37, 3001, 20000          # ragged particle shards and ragged source rows
    src, tgt = pkg.scans.random_clouds(B, M, seed=23, extent=25.0)
    init = pkg.scans.make_particles(P, seed=23) * 0.3
    prm = pkg.SteinICPParam(iterations=5, lr=1.0, max_dist=1.0, KNN_count=40, SVN_full_grad=True)
    s = ShardedSVNICP(prm, init, device_index=0, split="particles")
    s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), particles=s.get_particles(), cov=s.get_cov_matrix(),
             cand=s.get_candidates())
    dist.barrier(); dist.destroy_process_group()


def test_two_processes_one_gpu_sharded_driver(hip, tmp_path):
    """Rehearsal of the N > 1 path with the real HIP backend: two ranks (processes) share cuda:0,
    particles and source rows are sharded, the two all-gathers go through torch.distributed (gloo here;
    RCCL on a multi-GPU node).  Replicas must agree bit for bit and match the single-process run."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (np.load(str(tmp_path / f"r{r}.npz")) for r in range(2))
    assert np.array_equal(r0["particles"], r1["particles"]) and np.array_equal(r0["cov"], r1["cov"])
    P, B, M = 37, 3001, 20000
    src, tgt = hip.scans.random_clouds(B, M, seed=23, extent=25.0)
    init = hip.scans.make_particles(P, seed=23) * 0.3
    ref = hip.SVNICP(hip.SteinICPParam(iterations=5, lr=1.0, max_dist=1.0, KNN_count=40, SVN_full_grad=True), init)
    ref.add_cloud(src, tgt, init); ref.stein_align()
    assert np.array_equal(r0["cand"], ref.get_candidates())
    assert np.allclose(r0["particles"], ref.get_particles(), rtol=0, atol=1e-12)


@pytest.mark.parametrize("W,mode", [(2, "svn"), (3, "svn"), (2, "svgd")])
def test_row_shard_contexts_equal_single_context(hip, W, mode):
    """Source-row sharding through the C ABI (svnicp_set_row_shard): W contexts on one GPU play W ranks, each is handed only
    its rows of the source scan; the per-iteration 'all-gather' of the W x P x 22 partial records is a device copy.  The
    replicas must agree bit for bit; against the one-shot run the sums are grouped differently (1e-12)."""
    import torch
    from svnicp_amd.sharded import _DevView, shard_range
    L = hip.load_library()
    P, B, M, K, I = 70, 2501, 6000, 40, 5
    src, tgt = hip.scans.random_clouds(B, M, seed=9)
    init = hip.scans.make_particles(P, seed=9) * 0.3
    svgd = mode == "svgd"
    prm = hip.SteinICPParam(iterations=I, lr=0.01 if svgd else 1.0, max_dist=1.0, KNN_count=K, SVN_full_grad=False, optimizer="Adam")
    mk = (lambda: hip.SVGDICP(prm, init)) if svgd else (lambda: hip.SVNICP(prm, init))
    ref = mk(); ref.add_cloud(src, tgt, init); ref.stein_align()
    ranks = []
    for r in range(W):
        lo, hi = shard_range(B, W, r)
        s = mk(); s.add_cloud(np.ascontiguousarray(src[lo:hi]), tgt, init)
        assert L.svnicp_set_row_shard(s.handle, r, W, B) == 0
        assert L.svnicp_align(s.handle) < 0, "the one-shot entry point must refuse a row shard"
        assert L.svnicp_align_begin(s.handle) == 0
        assert L.svnicp_stage_candidates(s.handle, 0, hi - lo) == 0
        assert L.svnicp_build_candidate_table(s.handle) == 0
        assert np.array_equal(s.get_candidates(), ref.get_candidates()[lo:hi])
        ranks.append(s)
    views = [torch.as_tensor(_DevView(L.svnicp_rank_sums_devptr(s.handle), (W, P, 22), "<f8"), device="cuda") for s in ranks]
    for it in range(I):
        for s in ranks:
            assert L.svnicp_iter_accumulate(s.handle, it) == 0
            L.svnicp_synchronize(s.handle)
        for r in range(W):            # all-gather: slot r of every context <- rank r's own slot
            for q in range(W):
                if q != r:
                    views[q][r].copy_(views[r][r])
        torch.cuda.synchronize()
        for s in ranks:
            assert L.svnicp_iter_update(s.handle, it) == 0
    for s in ranks:
        assert L.svnicp_finish(s.handle) == 0
        L.svnicp_synchronize(s.handle)
        assert np.array_equal(s.get_particles(), ranks[0].get_particles())
        assert np.array_equal(s.get_cov_matrix(), ranks[0].get_cov_matrix())
        assert np.allclose(s.get_particles(), ref.get_particles(), rtol=0, atol=1e-12)
        assert np.allclose(s.get_cov_matrix(), ref.get_cov_matrix(), rtol=0, atol=1e-12)
    # back to one rank: the same context runs the one-shot registration again
    s = ranks[0]
    s.add_cloud(src, tgt, init)
    assert L.svnicp_set_row_shard(s.handle, 0, 1, 0) == 0
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    if not svgd:   # (SVGD keeps pose_particles_ across registrations: the second run starts elsewhere)
        assert np.array_equal(s.get_particles(), ref.get_particles())


def _two_rank_rows_worker(rank, world, port, out_dir, mode):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    import torch.distributed as dist
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svnicp_amd.sharded import ShardedSVGDICP, ShardedSVNICP
    P, B, M = 150, 2001, 9000          # P > 128: the workgroup-parallel Stein step on every rank; ragged row shards
    src, tgt = pkg.scans.random_clouds(B, M, seed=29, extent=25.0)
    init = pkg.scans.make_particles(P, seed=29) * 0.3
    svgd = mode == "svgd"
    prm = pkg.SteinICPParam(iterations=6, lr=0.01 if svgd else 1.0, max_dist=1.0, KNN_count=24, optimizer="Adam", SVN_full_grad=not svgd)
    s = (ShardedSVGDICP if svgd else ShardedSVNICP)(prm, init, device_index=0, split="rows")
    assert (s.Wp, s.Wb) == (1, world)
    s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), particles=s.get_particles(), cov=s.get_cov_matrix(), cand=s.get_candidates())
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["svn", "svgd"])
def test_two_processes_one_gpu_row_sharded(hip, orc, tmp_path, mode):
    """The row-sharded driver (ShardedSVNICP / ShardedSVGDICP, split="rows") with two ranks on cuda:0 over gloo: each rank
    uploads only its half of the source scan; replicas bit-identical, equal to the single-process run to 1e-12 and to the
    oracle to 1e-9."""
    import socket
    import torch.multiprocessing as mp
    from svnicp_amd.sharded import shard_range
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_two_rank_rows_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (np.load(str(tmp_path / f"r{r}.npz")) for r in range(2))
    assert np.array_equal(r0["particles"], r1["particles"]) and np.array_equal(r0["cov"], r1["cov"])
    P, B, M = 150, 2001, 9000
    src, tgt = hip.scans.random_clouds(B, M, seed=29, extent=25.0)
    init = hip.scans.make_particles(P, seed=29) * 0.3
    svgd = mode == "svgd"
    prm = hip.SteinICPParam(iterations=6, lr=0.01 if svgd else 1.0, max_dist=1.0, KNN_count=24, optimizer="Adam", SVN_full_grad=not svgd)
    ref = (hip.SVGDICP if svgd else hip.SVNICP)(prm, init); ref.add_cloud(src, tgt, init); ref.stein_align()
    lo, hi = shard_range(B, 2, 1)
    assert np.array_equal(r1["cand"], ref.get_candidates()[lo:hi])
    assert np.allclose(r0["particles"], ref.get_particles(), rtol=0, atol=1e-12)
    o = orc.Solver(init, mode=orc.MODE_SVGD if svgd else orc.MODE_SVN, iterations=6, lr=prm.lr, max_dist=1.0, knn_count=24,
                   svn_full_grad=not svgd, optimizer="Adam")
    o.add_cloud(src, tgt, init); o.stein_align()
    assert np.allclose(r0["particles"], o.get_particles(), rtol=0, atol=TIGHT)


def _two_rank_svgd_worker(rank, world, port, out_dir):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    import torch.distributed as dist
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svnicp_amd.sharded import ShardedSVGDICP
    P, B, M = 150, 2001, 9000          # P > 128: the workgroup-parallel SVGD step on every rank; ragged shards
    src, tgt = pkg.scans.random_clouds(B, M, seed=29, extent=25.0)
    init = pkg.scans.make_particles(P, seed=29) * 0.3
    prm = pkg.SteinICPParam(iterations=6, lr=0.01, max_dist=1.0, KNN_count=24, optimizer="Adam")
    s = ShardedSVGDICP(prm, init, device_index=0, split="particles")
    s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), particles=s.get_particles(), cov=s.get_cov_matrix())
    dist.barrier(); dist.destroy_process_group()


def test_two_processes_one_gpu_sharded_svgd(hip, orc, tmp_path):
    """SVGD-ICP through the sharded driver (two ranks on cuda:0, gloo): replicas bit-identical, equal to the
    single-process SVGDICP run and to the oracle."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_two_rank_svgd_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (np.load(str(tmp_path / f"r{r}.npz")) for r in range(2))
    assert np.array_equal(r0["particles"], r1["particles"]) and np.array_equal(r0["cov"], r1["cov"])
    P, B, M = 150, 2001, 9000
    src, tgt = hip.scans.random_clouds(B, M, seed=29, extent=25.0)
    init = hip.scans.make_particles(P, seed=29) * 0.3
    cfg = dict(iterations=6, lr=0.01, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=24, optimizer="Adam")
    ref = _hip_svgd(hip, init, cfg, trace=False); ref.add_cloud(src, tgt, init); ref.stein_align()
    assert np.allclose(r0["particles"], ref.get_particles(), rtol=0, atol=1e-12)
    o = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **cfg); o.add_cloud(src, tgt, init); o.stein_align()
    assert np.allclose(r0["particles"], o.get_particles(), rtol=0, atol=TIGHT)


def _two_rank_nccl_worker(rank, world, port, out_dir, split):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    import torch
    import torch.distributed as dist
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from svnicp_amd.sharded import ShardedSVNICP
    P, B, M = 37, 3001, 20000          # ragged particle shards and ragged source rows: padded all-gathers
    src, tgt = pkg.scans.random_clouds(B, M, seed=23, extent=25.0)
    init = pkg.scans.make_particles(P, seed=23) * 0.3
    prm = pkg.SteinICPParam(iterations=5, lr=1.0, max_dist=1.0, KNN_count=40, SVN_full_grad=True)
    s = ShardedSVNICP(prm, init, device_index=rank, split=split)
    s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), particles=s.get_particles(), cov=s.get_cov_matrix(),
             cand=s.get_candidates())
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("split", ["rows", "particles"])
def test_two_gpus_rccl_sharded_driver(hip, tmp_path, split):
    """The RCCL branch of the sharded driver (in-place all-gathers into library-owned device memory through
    __cuda_array_interface__) on two real GPUs, both splits: replicas bit-identical, equal to the single-process run.
    Skipped on a one-GPU box — the driver's multi-GPU node runs it."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_two_rank_nccl_worker, args=(2, port, str(tmp_path), split), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (np.load(str(tmp_path / f"r{r}.npz")) for r in range(2))
    assert np.array_equal(r0["particles"], r1["particles"]) and np.array_equal(r0["cov"], r1["cov"])
    P, B, M = 37, 3001, 20000
    src, tgt = hip.scans.random_clouds(B, M, seed=23, extent=25.0)
    init = hip.scans.make_particles(P, seed=23) * 0.3
    ref = hip.SVNICP(hip.SteinICPParam(iterations=5, lr=1.0, max_dist=1.0, KNN_count=40, SVN_full_grad=True), init)
    ref.add_cloud(src, tgt, init); ref.stein_align()
    rows = slice(0, (B + 1) // 2) if split == "rows" else slice(0, B)     # rank 0 holds only its rows when they are sharded
    assert np.array_equal(r0["cand"], ref.get_candidates()[rows])
    assert np.allclose(r0["particles"], ref.get_particles(), rtol=0, atol=1e-12)


# ------------------------------------------------------------------ the C++ RCCL host
def _write_case(path, src, tgt, init, iterations, knn, full, es, lr, max_dist, thr):
    import struct
    B, M, P = src.shape[0], tgt.shape[0], init.shape[1]
    with open(path, "wb") as f:
        f.write(struct.pack("<qq", B, M)); f.write(struct.pack("<iiiii", P, iterations, knn, int(full), int(es)))
        f.write(struct.pack("<ddd", lr, max_dist, thr))
        f.write(np.ascontiguousarray(src, np.float64).tobytes()); f.write(np.ascontiguousarray(tgt, np.float64).tobytes())
        f.write(np.ascontiguousarray(init, np.float64).tobytes())


def _read_result(path, P, knn):
    raw = open(path, "rb").read()
    state, iters = np.frombuffer(raw, np.int32, 2, 0)
    off = 8
    mean = np.frombuffer(raw, np.float64, 6, off); off += 48
    cov = np.frombuffer(raw, np.float64, 36, off); off += 288
    part = np.frombuffer(raw, np.float64, 6 * P, off); off += 48 * P
    rows = int(np.frombuffer(raw, np.int64, 1, off)[0]); off += 8
    cand = np.frombuffer(raw, np.int32, rows * knn, off).reshape(rows, knn)
    return int(state), int(iters), mean, cov, part, cand


@pytest.mark.parametrize("split,mode", [("rows", "svn"), ("particles", "svn"), ("rows", "svgd")])
def test_cpp_rccl_host_world1(hip, tmp_path, split, mode):
    """svn-icp_amd/host/sharded_drive (C++: svnicp::Sharded<> on the split-phase C ABI, ncclCommInitRank, ncclAllGather on the
    library's stream) as a single rank: the split-phase sequence must give the bits of the one-shot svnicp_align."""
    import subprocess
    import __graft_entry__ as graft
    exe = graft.build_cpp_host()
    P, B, M, K, I = 40, 3000, 9000, 24, 6
    src, tgt = hip.scans.random_clouds(B, M, seed=61, extent=25.0)
    init = hip.scans.make_particles(P, seed=61) * 0.3
    svgd = mode == "svgd"
    lr = 0.01 if svgd else 1.0
    _write_case(str(tmp_path / "case.bin"), src, tgt, init, I, K, False, True, lr, 1.0, 1e-7)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([exe, str(tmp_path / "case.bin"), str(tmp_path / "out.bin"), split, mode], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    state, iters, mean, cov, part, cand = _read_result(str(tmp_path / "out.bin"), P, K)
    prm = hip.SteinICPParam(iterations=I, lr=lr, max_dist=1.0, KNN_count=K, SVN_full_grad=False, check_early_stop=True,
                            convergence_threshold=1e-7, optimizer="Adam")
    ref = (hip.SVGDICP if svgd else hip.SVNICP)(prm, init); ref.add_cloud(src, tgt, init)
    assert ref.stein_align() == state == hip.SteinICPState.ALIGN_SUCCESS
    assert iters == ref.get_iterations_run()
    assert np.array_equal(cand, ref.get_candidates())
    assert np.array_equal(part, ref.get_particles().ravel())
    assert np.array_equal(mean, ref.get_transformation()) and np.array_equal(cov, ref.get_cov_matrix().ravel())


@pytest.mark.parametrize("split", ["rows", "particles"])
def test_cpp_rccl_host_two_gpus(hip, tmp_path, split):
    """Two ranks of the C++ host on two GPUs (ncclUniqueId through a file): replicas bit-identical, the single-process result
    to 1e-12.  Skipped on a one-GPU box (RCCL refuses two ranks on one device)."""
    import subprocess
    import torch
    import __graft_entry__ as graft
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    exe = graft.build_cpp_host()
    P, B, M, K, I = 40, 3000, 9000, 24, 6
    src, tgt = hip.scans.random_clouds(B, M, seed=61, extent=25.0)
    init = hip.scans.make_particles(P, seed=61) * 0.3
    _write_case(str(tmp_path / "case.bin"), src, tgt, init, I, K, True, False, 1.0, 1.0, 1e-5)
    procs = [subprocess.Popen([exe, str(tmp_path / "case.bin"), str(tmp_path / f"out{r}.bin"), split, "svn", str(tmp_path / "nccl.id")],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=300) for p in procs] == [0, 0]
    res = [_read_result(str(tmp_path / f"out{r}.bin"), P, K) for r in range(2)]
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][3], res[1][3])
    ref = hip.SVNICP(hip.SteinICPParam(iterations=I, lr=1.0, max_dist=1.0, KNN_count=K, SVN_full_grad=True), init)
    ref.add_cloud(src, tgt, init); ref.stein_align()
    assert np.allclose(res[0][4], ref.get_particles().ravel(), rtol=0, atol=1e-12)


# ------------------------------------------------------------------ BASELINE sizes
def test_c1_full_parity(hip, orc):
    """BASELINE config C1 (the reference-CPU-path config): 1 particle, 4096 x 8192, 20 iterations."""
    cfg = hip.scans.CONFIGS["C1"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(1)
    c = dict(iterations=20, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    o = orc.Solver(init, **c); o.add_cloud(pair.source, pair.target, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **c); s.add_cloud(pair.source, pair.target, init); s.stein_align()
    _compare(s, o, tro, 1)


def test_c2_full_parity(hip, orc):
    """BASELINE config C2 at full size, all 20 iterations, against the oracle (all host cores): candidate
    lists and dist² bit-exact, every iteration's correspondences bit-exact for all 32 x 65536 pairs, H/b/step to
    the trace tolerances, pose/covariance/particles to 1e-9 (stated bar 1e-4)."""
    cfg = hip.scans.CONFIGS["C2"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=20, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    o = orc.Solver(init, **c); o.add_cloud(pair.source, pair.target, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **c); s.add_cloud(pair.source, pair.target, init); s.stein_align()
    _compare(s, o, tro, cfg["P"])
    err = np.abs(s.get_transformation() - o.get_transformation())
    assert err[:3].max() < POSE_TOL and err[3:].max() < POSE_TOL


@pytest.mark.parametrize("wl,Bs,iters", [("C3", 4096, 20), ("C4", 16384, 20), ("C5", 4096, 20)])
def test_headline_configs_reduced_source_full_parity(hip, orc, wl, Bs, iters):
    """BASELINE configs C3 / C4 / C5 with their real particle count, K, target cloud and (for C3) iteration
    count, on an evenly spaced subset of the source scan so that the CPU oracle finishes in seconds: candidate
    lists and dist² bit-exact, per-iteration correspondences bit-exact, H/b/step to the trace tolerances, pose,
    covariance and particles to 1e-9.  All three run their full 20 iterations.  C4 (16384 rows) exercises the pair statistics
    of 512 particles (k_upd_hist chain) and four particle groups per search wave, C5 (4096 rows) the 2 M-point target
    (4096 Morton tiles, sliced fallback)."""
    cfg = hip.scans.CONFIGS[wl]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"])
    rows = np.linspace(0, cfg["B"] - 1, Bs).astype(np.int64)
    src = np.ascontiguousarray(pair.source[rows])
    init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=iters, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    o = orc.Solver(init, **c); o.add_cloud(src, pair.target, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **c); s.add_cloud(src, pair.target, init)
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    _compare(s, o, tro, cfg["P"])


def _fuzz_cloud(kind, rng, B, M):
    """Point sets built to stress the float32 searches: exact ties, duplicates, huge offsets, tiny scales."""
    if kind == "grid":        # integer lattice: exact ties everywhere
        tgt = rng.integers(-6, 7, size=(M, 3)).astype(np.float64) * 0.25
        src = rng.integers(-6, 7, size=(B, 3)).astype(np.float64) * 0.25 + rng.choice([0.0, 0.125], size=(B, 1))
    elif kind == "dups":      # few distinct targets, many copies
        base = rng.normal(size=(max(8, M // 50), 3))
        tgt = base[rng.integers(0, base.shape[0], M)]
        src = base[rng.integers(0, base.shape[0], B)] + rng.normal(size=(B, 3)) * 0.05
    elif kind == "far":       # UTM-like coordinates: the error bounds scale with |coordinate|
        off = np.array([4.5e5, -3.2e5, 1.2e3])
        tgt = rng.normal(size=(M, 3)) * 3 + off
        src = tgt[rng.integers(0, M, B)] + rng.normal(size=(B, 3)) * 0.05
    elif kind == "tiny":      # millimetre-scale cloud
        tgt = rng.normal(size=(M, 3)) * 1e-3
        src = tgt[rng.integers(0, M, B)] + rng.normal(size=(B, 3)) * 2e-5
    else:                     # thin plane + line: strongly anisotropic neighbourhoods
        tgt = np.concatenate([np.c_[rng.uniform(-5, 5, (M // 2, 2)), rng.normal(size=M // 2) * 1e-4],
                              np.c_[rng.uniform(-5, 5, M - M // 2), np.zeros(M - M // 2), np.full(M - M // 2, 0.3)]])
        src = tgt[rng.integers(0, M, B)] + rng.normal(size=(B, 3)) * 0.02
    return (src.astype(np.float32).astype(np.float64) if kind != "far" else src), tgt


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_searches_stay_exact(hip, orc, seed):
    """Randomised structure / size / K / particle count: candidate lists bit-exact and the first iteration's
    correspondences bit-exact for every (particle, point) — the float32 MFMA search with its ambiguity test must
    never mis-pick.  Later iterations are not compared here: several of these clouds are deliberately
    ill-conditioned (UTM-sized coordinates, millimetre scale, a handful of distinct points), so a last-bit
    difference in the summation order legitimately moves the poses, and with them the later correspondences."""
    rng = np.random.default_rng(1000 + seed)
    kind = ["grid", "dups", "far", "tiny", "aniso"][seed % 5]
    B, M = int(rng.integers(200, 1500)), int(rng.integers(600, 9000))
    K = int(rng.choice([1, 5, 16, 17, 50, 96, 97, 100, 128]))
    P = int(rng.choice([2, 9, 16, 33, 64, 96, 130]))
    src, tgt = _fuzz_cloud(kind, rng, B, M)
    scale = 1e-3 if kind == "tiny" else 1.0
    init = hip.scans.make_particles(P, seed=seed + 1) * (0.2 * scale)
    init[:3] *= 1.0
    cfg = dict(iterations=1, lr=1.0, max_dist=(1.0 if kind != "tiny" else 1e-6), knn_count=min(K, 128), svn_full_grad=bool(seed & 1))
    o = orc.Solver(init, **cfg); o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **cfg); s.add_cloud(src, tgt, init); s.stein_align()
    assert np.array_equal(s.get_candidates().astype(np.int64), o.candidates()), kind
    assert np.array_equal(s.get_candidate_dist2(), o.candidate_dist2()), kind
    assert np.array_equal(s.get_trace()["corr"][:1], tro["corr"][:1]), kind
    assert np.allclose(s.get_trace()["H"][:1], tro["H"][:1], rtol=1e-9, atol=1e-9 * scale * scale), kind


def test_c3_headline_properties(hip, orc):
    """Headline config C3 at full size through size-independent properties: (a) a random sample of
    candidate rows equals the oracle's brute force on those rows bit-for-bit, (b) every row is
    ascending by (dist², idx) with in-range indices, (c) the run is deterministic, (d) weights sum
    to one and the covariance is symmetric PSD, (e) the registration moves toward the planted pose."""
    cfg = hip.scans.CONFIGS["C3"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=20, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    s = _hip_solver(hip, init, trace=False, **c); s.add_cloud(pair.source, pair.target, init); s.stein_align()
    ci, d2 = s.get_candidates(), s.get_candidate_dist2()
    rows = np.random.default_rng(1).choice(cfg["B"], 256, replace=False)
    oi, od = orc.knn_topk(pair.source[rows], pair.target, 100)
    assert np.array_equal(ci[rows].astype(np.int64), oi) and np.array_equal(d2[rows], od)
    assert ci.min() >= 0 and ci.max() < cfg["M"]
    dd = np.diff(d2, axis=1)
    assert np.all(dd >= 0) and np.all(np.diff(ci, axis=1)[dd == 0] > 0)
    p1, cov = s.get_particles(), s.get_cov_matrix().reshape(6, 6)
    s.add_cloud(pair.source, pair.target, init); s.stein_align()
    assert np.array_equal(p1, s.get_particles())
    assert abs(s.get_particle_weight().sum() - 1.0) < 1e-6
    assert np.allclose(cov, cov.T, atol=1e-15) and np.linalg.eigvalsh(cov).min() > -1e-12
    m = s.get_transformation()
    assert np.linalg.norm(m[:3] - pair.true_pose[:3]) < np.linalg.norm(pair.true_pose[:3])
    assert np.isfinite(s.get_particle_history()).all()

"""GPU parity at BASELINE.json's full sizes, where the size-dependent code paths live (Morton tiles, deep
survivor pools, the sliced stage-A fallback, two particle waves per workgroup, the workgroup-parallel Stein
step), plus the corner cases the small suite only covered on the CPU oracle: the NaN left Jacobian of an
exactly-zero rotation step (SVNICP.cpp:188-192) and SVGD-ICP at production particle counts.
Everything goes through the C ABI; the oracle is the checker."""
import numpy as np
import pytest

from helpers import POSE_TOL, TIGHT
from test_gpu_parity import _compare, _hip_solver, _hip_svgd

pytestmark = pytest.mark.gpu


def test_c3_full_size_every_row_against_oracle(hip, orc):
    """Headline config C3 (128 particles, 131072 x 262144, K = 100, 20 iterations) at FULL size against the
    oracle: all 131072 candidate rows idx + dist² bit-exact, the correspondences of all 20 iterations bit-exact
    for every one of the 128 x 131072 pairs, H/b/step per iteration, final pose / covariance / particles to 1e-9
    (stated bar: 1e-4 m / 1e-4 rad).  The oracle takes ~25 s on the box's host cores."""
    cfg = hip.scans.CONFIGS["C3"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=20, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    o = orc.Solver(init, **c); o.add_cloud(pair.source, pair.target, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **c); s.add_cloud(pair.source, pair.target, init)
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    _compare(s, o, tro, cfg["P"])
    err = np.abs(s.get_transformation() - o.get_transformation())
    assert err[:3].max() < POSE_TOL and err[3:].max() < POSE_TOL
    assert 0 <= s.get_ambiguous_steps() < 0.5 * 20 * 2 * cfg["B"]   # the f32 search decides most wave steps itself


def test_c3_full_size_full_svn_branch(hip, orc):
    """C3 with SVNFullGrad = true (BASELINE.md §2 reports both branches): 4 iterations at full size vs the oracle."""
    cfg = hip.scans.CONFIGS["C3"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=4, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=True)
    o = orc.Solver(init, **c); o.add_cloud(pair.source, pair.target, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **c); s.add_cloud(pair.source, pair.target, init); s.stein_align()
    _compare(s, o, tro, cfg["P"])


def test_c5_full_size_fallback_rows_and_sample(hip, orc):
    """C5 (2 M-point target, 4096 Morton tiles) at FULL size: the tiles kernel hands a few queries to the sliced
    fallback (scan from the proven threshold + 64-way merge); every one of those rows, plus 2048 random rows, must
    equal the oracle's brute force bit for bit (idx and dist²), and their first-iteration correspondences for
    all 128 particles must be the oracle's.  All rows: ascending by (dist², idx), indices in range."""
    cfg = hip.scans.CONFIGS["C5"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=1, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    s = _hip_solver(hip, init, **c); s.add_cloud(pair.source, pair.target, init)
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    fb = s.get_knn_fallback_rows()
    assert fb.size == s.get_knn_fallbacks()
    rows = np.unique(np.concatenate([fb.astype(np.int64), np.random.default_rng(5).choice(cfg["B"], 2048, replace=False)]))
    ci, d2 = s.get_candidates(), s.get_candidate_dist2()
    src = np.ascontiguousarray(pair.source[rows])
    o = orc.Solver(init, **c); o.add_cloud(src, pair.target, init); tro = o.enable_trace(); o.stein_align()
    assert np.array_equal(ci[rows].astype(np.int64), o.candidates())
    assert np.array_equal(d2[rows], o.candidate_dist2())
    assert np.array_equal(s.get_trace()["corr"][0][:, rows], tro["corr"][0])
    assert ci.min() >= 0 and ci.max() < cfg["M"]
    dd = np.diff(d2, axis=1)
    assert np.all(dd >= 0) and np.all(np.diff(ci, axis=1)[dd == 0] > 0)
    print(f"C5 full size: {fb.size} fallback rows checked, {rows.size} rows against the oracle")


def test_c5_full_size_twenty_iterations_properties(hip):
    """C5 at full size and full iteration count: deterministic, finite, symmetric PSD covariance."""
    cfg = hip.scans.CONFIGS["C5"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"]); init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=20, lr=1.0, max_dist=1.0, knn_count=100, svn_full_grad=False)
    s = _hip_solver(hip, init, trace=False, **c)
    s.add_cloud(pair.source, pair.target, init); s.stein_align(); p1 = s.get_particles()
    s.add_cloud(pair.source, pair.target, init); s.stein_align()
    assert np.array_equal(p1, s.get_particles()) and np.isfinite(p1).all()
    cov = s.get_cov_matrix().reshape(6, 6)
    assert np.allclose(cov, cov.T, atol=1e-15) and np.linalg.eigvalsh(cov).min() > -1e-12


@pytest.mark.parametrize("P", [1, 8, 130])
def test_zero_rotation_step_gives_nan_left_jacobian(hip, orc, P):
    """SVNICP::to_rotation_tensor divides by the rotation angle (SVNICP.cpp:188-192): an exactly-zero rotation
    step makes J_l NaN and the translation update NaN with it.  All source points at the origin make the
    rotational Jacobian block vanish (J = [R | -R s^], s = 0), so b, the Newton step and the repulsion term have
    exactly zero rotation components while the translations move.  The HIP update kernels (fused P = 1,
    front + direction P <= 128, workgroup-parallel chain above) must reproduce the oracle's NaNs and its
    untouched rotations."""
    rng = np.random.default_rng(P)
    tgt = rng.normal(size=(500, 3)) * 0.5
    src = np.zeros((64, 3))
    init = np.zeros((6, P)); init[:3] = rng.normal(size=(3, P)) * 0.1
    c = dict(iterations=2, lr=1.0, max_dist=10.0, knn_count=8, svn_full_grad=False)
    o = orc.Solver(init, **c); o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_solver(hip, init, **c); s.add_cloud(src, tgt, init); s.stein_align()
    po, ps = o.get_particles().reshape(6, P), s.get_particles().reshape(6, P)
    assert np.isnan(po[:3]).all(), "the oracle must hit the 0/0 of SVNICP.cpp:188"
    assert np.array_equal(np.isnan(ps), np.isnan(po))
    assert np.array_equal(ps[3:], po[3:])                     # rotations: exactly the initial zeros
    tr = s.get_trace()
    assert np.allclose(tr["phi"][0], tro["phi"][0], rtol=1e-7, atol=1e-10)   # first step: finite, zero rotation part
    assert np.all(tr["phi"][0][:, 3:] == 0.0)
    assert np.array_equal(np.isnan(s.get_transformation()), np.isnan(o.get_transformation()))
    assert np.array_equal(np.isnan(s.get_cov_matrix()), np.isnan(o.get_cov_matrix()))


@pytest.mark.parametrize("P,B,M,opt,es", [(128, 4096, 16384, "Adam", False), (512, 1536, 8192, "RMSprop", False), (200, 1000, 4000, "SGD", False),
                                          (130, 800, 3000, "Adagrad", False), (300, 900, 3000, "Adam", True), (1030, 400, 2000, "Adam", False)])
def test_svgd_mode_at_production_particle_counts(hip, orc, P, B, M, opt, es):
    """SVGD-ICP (SVGDICP.cpp:66-140, 398-494) with 128…1030 particles (one-workgroup step up to 128, the workgroup-parallel
    chain above; one case with the early stop firing) against the oracle: candidate lists bit-exact,
    gradients / Stein direction per iteration, final particles, mean and covariance to 1e-9."""
    src, tgt = hip.scans.random_clouds(B, M, seed=P + 1, extent=20.0)
    init = hip.scans.make_particles(P, seed=P) * 0.3
    cfg = dict(iterations=(30 if es else 6), lr=0.01, max_dist=1.0, check_early_stop=es, convergence_threshold=(0.05 if es else 1e-5),
               knn_count=32, optimizer=opt)
    o = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **cfg)
    o.add_cloud(src, tgt, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_svgd(hip, init, cfg); s.add_cloud(src, tgt, init); s.set_initial_mean(np.eye(4))
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    assert np.array_equal(s.get_candidates().astype(np.int64), o.candidates())
    tr = s.get_trace()
    n = o.iterations_run()
    assert s.get_iterations_run() == n and int(s.get_runtime()[2]) == o.finish_iter() and (not es or n < cfg["iterations"])
    assert np.array_equal(tr["corr"][:n], tro["corr"][:n])
    assert np.allclose(tr["newton"][:n], tro["newton"][:n], rtol=1e-9, atol=1e-9)
    assert np.allclose(tr["phi"][:n], tro["phi"][:n], rtol=1e-9, atol=1e-9)
    assert np.allclose(tr["h"][:n], tro["h"][:n], rtol=1e-10)
    assert np.allclose(s.get_particle_history(), o.get_particle_history(), atol=1e-6)
    for got, want in ((s.get_transformation(), o.get_transformation()), (s.get_distribution(), o.get_distribution()),
                      (s.get_cov_matrix(), o.get_cov_matrix()), (s.get_particles(), o.get_particles())):
        assert np.allclose(got, want, rtol=0, atol=TIGHT)


def test_svgd_adam_at_c3_size(hip, orc):
    """SVGD-ICP (Adam, lr 0.01 — the configuration `bench.py --mode svgd` times) at C3's particle count, target cloud, K and
    iteration count on 8192 evenly spaced source rows against the oracle: candidate lists bit-exact, every iteration's
    correspondences bit-exact, gradients and Stein directions per iteration, final particles / mean / covariance to 1e-9."""
    cfg = hip.scans.CONFIGS["C3"]
    pair = hip.scans.make_pair(cfg["B"], cfg["M"])
    rows = np.linspace(0, cfg["B"] - 1, 8192).astype(np.int64)
    src = np.ascontiguousarray(pair.source[rows])
    init = hip.scans.make_particles(cfg["P"])
    c = dict(iterations=20, lr=0.01, max_dist=1.0, check_early_stop=False, convergence_threshold=1e-5, knn_count=100, optimizer="Adam")
    o = orc.Solver(init, mode=orc.MODE_SVGD, svn_full_grad=False, **c)
    o.add_cloud(src, pair.target, init); tro = o.enable_trace(); o.stein_align()
    s = _hip_svgd(hip, init, c); s.add_cloud(src, pair.target, init); s.set_initial_mean(np.eye(4))
    assert s.stein_align() == hip.SteinICPState.ALIGN_SUCCESS
    assert np.array_equal(s.get_candidates().astype(np.int64), o.candidates())
    tr = s.get_trace()
    assert np.array_equal(tr["corr"], tro["corr"])
    assert np.allclose(tr["newton"], tro["newton"], rtol=1e-9, atol=1e-9)
    assert np.allclose(tr["phi"], tro["phi"], rtol=1e-9, atol=1e-9)
    assert np.allclose(tr["h"], tro["h"], rtol=1e-10)
    for got, want in ((s.get_transformation(), o.get_transformation()), (s.get_cov_matrix(), o.get_cov_matrix()),
                      (s.get_particles(), o.get_particles())):
        assert np.allclose(got, want, rtol=0, atol=TIGHT)

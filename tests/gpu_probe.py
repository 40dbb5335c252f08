"""Ad-hoc GPU probe used during bring-up: HIP path vs oracle on a few shapes (not a pytest file)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()

def run(P, B, M, K, I, full, es=False, thr=1e-5, md=1.0, lr=1.0, seed=1, trace=True):
    src, tgt = pkg.scans.random_clouds(B, M, seed=seed)
    init = pkg.scans.make_particles(P, seed=seed) * 0.3
    R0 = np.eye(3); t0 = np.array([0.01, -0.02, 0.005])
    o = orc.Solver(init, iterations=I, lr=lr, max_dist=md, knn_count=K, svn_full_grad=full, check_early_stop=es, convergence_threshold=thr)
    o.add_cloud(src, tgt, init); o.set_initial_mean(R0, t0); tro = o.enable_trace(); t = time.time(); o.stein_align(); to = time.time() - t
    prm = pkg.SteinICPParam(iterations=I, lr=lr, max_dist=md, KNN_count=K, SVN_full_grad=full, check_early_stop=es, convergence_threshold=thr, record_trace=trace)
    s = pkg.SVNICP(prm, init, pkg.ParticleWeightOpt()); s.add_cloud(src, tgt, init); s.set_initial_mean((R0, t0))
    t = time.time(); s.stein_align(); th = time.time() - t
    ci = np.array_equal(o.candidates(), s.get_candidates().astype(np.int64))
    cd = np.array_equal(o.candidate_dist2(), s.get_candidate_dist2())
    tr = s.get_trace()
    n = o.iterations_run()
    mism = int((tr['corr'][:n] != tro['corr'][:n]).sum())
    dH = np.abs(tr['H'][:n] - tro['H'][:n]).max() / np.abs(tro['H'][:n]).max()
    db = np.abs(tr['b'][:n] - tro['b'][:n]).max() / max(np.abs(tro['b'][:n]).max(), 1e-300)
    dphi = np.abs(tr['phi'][:n] - tro['phi'][:n]).max()
    dh = np.nanmax(np.abs(tr['h'][:n] - tro['h'][:n])) if P > 1 else 0
    dm = np.abs(o.get_transformation() - s.get_transformation()).max()
    dc = np.abs(o.get_cov_matrix() - s.get_cov_matrix()).max()
    dv = np.abs(o.get_distribution() - s.get_distribution()).max()
    dhist = np.abs(o.get_particle_history() - s.get_particle_history()).max()
    print(f"P{P} B{B} M{M} K{K} I{I} full{int(full)} es{int(es)}: cand_eq={ci} d2_eq={cd} corr_mism={mism} dH={dH:.1e} db={db:.1e} dphi={dphi:.1e} dh={dh:.1e} dmean={dm:.1e} dvar={dv:.1e} dcov={dc:.1e} dhist={dhist:.1e} fin={n}/{s.get_runtime()[2]:.0f} t_orc={to:.2f}s t_hip={th:.3f}s gpu_ms={s.get_gpu_ms().round(3)}", flush=True)

if __name__ == "__main__":
    run(1, 256, 1000, 7, 8, False)
    run(4, 300, 1000, 10, 8, False)
    run(4, 300, 1000, 10, 8, True)
    run(8, 512, 2048, 32, 10, False, md=0.05)
    run(8, 512, 2048, 32, 30, True, es=True, thr=2e-2)
    run(33, 200, 150, 100, 5, False)
    run(64, 256, 500, 16, 6, True, lr=0.5)
    run(128, 2048, 8192, 100, 5, False)
    run(130, 1000, 5000, 100, 4, True)
    run(32, 4096, 8192, 100, 6, False)

"""tests/golden/make_golden.py — regenerates tests/golden/*.npz.

The vectors are produced by oracle/torch_restatement.py, i.e. by replaying the reference's tensor
program op-for-op on libtorch CPU float64 (torch version recorded in each file).  They are NOT
outputs of the reference binary: its solver TUs cannot be built in this image (PCL, Eigen, GTSAM,
rclcpp headers are absent and stand-ins are not allowed), and the reference ships no test vectors
of its own — solver parity is therefore "unpinned" by the reference; these files pin
the C oracle and the HIP path to the ATen arithmetic instead.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as graft  # noqa: E402
import torch_restatement as tr  # noqa: E402

pkg = graft.load_package()
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, mode, P, B, M, K, I, full, early_stop, thr, max_dist, lr, optimizer, seed
    ("svn_p1", "svn", 1, 64, 200, 7, 6, False, False, 1e-5, 1.0, 1.0, "Adam", 1),
    ("svn_p4_default", "svn", 4, 96, 300, 10, 6, False, False, 1e-5, 1.0, 1.0, "Adam", 2),
    ("svn_p4_full", "svn", 4, 96, 300, 10, 6, True, False, 1e-5, 1.0, 0.7, "Adam", 3),
    ("svn_p8_mask", "svn", 8, 128, 400, 32, 6, False, False, 1e-5, 0.05, 1.0, "Adam", 4),
    ("svn_p8_earlystop", "svn", 8, 128, 400, 16, 12, True, True, 3e-2, 1.0, 1.0, "Adam", 5),
    ("svn_p64_k100", "svn", 64, 64, 150, 100, 3, False, False, 1e-5, 1.0, 1.0, "Adam", 6),
    ("svgd_adam", "svgd", 6, 96, 300, 10, 6, False, False, 1e-5, 1.0, 0.02, "Adam", 7),
    ("svgd_rmsprop", "svgd", 6, 96, 300, 10, 6, False, False, 1e-5, 1.0, 0.005, "RMSprop", 8),
    ("svgd_sgd", "svgd", 6, 96, 300, 10, 6, False, False, 1e-5, 1.0, 1e-4, "SGD", 9),
    ("svgd_adagrad", "svgd", 6, 96, 300, 10, 8, False, True, 3e-2, 1.0, 0.02, "Adagrad", 10),
]


def main():
    for (name, mode, P, B, M, K, I, full, es, thr, md, lr, optname, seed) in CASES:
        src, tgt = pkg.scans.random_clouds(B, M, seed=seed)
        init = pkg.scans.make_particles(P, seed=seed) * 0.3
        R0 = pkg.scans.rot_zyx(0.002, -0.001, 0.003)
        t0 = np.array([0.01, -0.02, 0.005])
        prm = tr.SteinICPParam(iterations=I, lr=lr, max_dist=md, check_early_stop=es, convergence_threshold=thr,
                               KNN_count=K, SVN_full_grad=full, optimizer=optname)
        cls = tr.SVNICP if mode == "svn" else tr.SVGDICP
        s = cls(prm, torch.tensor(init))
        s.add_cloud(torch.tensor(src), torch.tensor(tgt), torch.tensor(init))
        s.set_initial_mean(R0, t0)
        state = s.stein_align()
        n = len(s.trace["phi"])
        d = dict(
            src=src, tgt=tgt, init=init, R0=R0, t0=t0,
            params=np.array([I, lr, md, int(es), thr, K, int(full)], np.float64), optimizer=optname, mode=mode,
            torch_version=torch.__version__, state=state, iters_run=n,
            cand_idx=s.sourceKNN_idx.numpy().astype(np.int32), cand_d2=s.sourceKNN_d2.numpy(),
            corr=np.stack([c.numpy() for c in s.trace["corr"]]).astype(np.int16),
            mask=np.stack([c.numpy() for c in s.trace["mask"]]).astype(np.uint8),
            phi=np.stack([c.numpy() for c in s.trace["phi"]]), h=np.array(s.trace["h"]),
            newton=np.stack([c.numpy() for c in s.trace["newton"]]),
            mean=s.get_transformation().numpy(), var=s.get_distribution().numpy(), cov=s.get_cov_matrix().numpy(),
            particles=s.get_particles().numpy(), weights=s.get_particle_weight().numpy(),
            history=s.get_particle_history().reshape(I, -1).numpy())
        if mode == "svn":
            d["H"] = np.stack([c.numpy() for c in s.trace["H"]]).reshape(n, P, 36)
            d["b"] = np.stack([c.numpy() for c in s.trace["b"]]).reshape(n, P, 6)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "iters", n, "mean", d["mean"].round(5))


if __name__ == "__main__":
    main()

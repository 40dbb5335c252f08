"""Shared helpers of the test-suite (test infrastructure; may use the oracle)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases(mode=None):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(f)[:-4]
        if mode is None or name.startswith(mode):
            out.append(name)
    return out


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: d[k] for k in d.files}
    I, lr, md, es, thr, K, full = g["params"]
    g["cfg"] = dict(iterations=int(I), lr=float(lr), max_dist=float(md), check_early_stop=bool(es),
                    convergence_threshold=float(thr), knn_count=int(K), svn_full_grad=bool(full),
                    optimizer=str(g["optimizer"]))
    g["mode"] = str(g["mode"])
    return g


def oracle_from_golden(orc, g):
    mode = orc.MODE_SVN if g["mode"] == "svn" else orc.MODE_SVGD
    s = orc.Solver(g["init"], mode=mode, **g["cfg"])
    s.add_cloud(g["src"], g["tgt"], g["init"])
    s.set_initial_mean(g["R0"], g["t0"])
    return s


# tolerances of the parity bar (BASELINE.json north_star): pose within 1e-4 m / 1e-4 rad.
# The suite holds the implementations to a far tighter band on the small cases.
POSE_TOL = 1e-4
TIGHT = 1e-9

"""bring-up timing helper (not a pytest file): k_knn_brute's launch time against the queries per workgroup (option brute_qb)
at the scan-to-map loop's sizes.   python tests/gpu_time_brute.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
M, K = int(os.environ.get("M", "50000")), 100
for B in [int(x) for x in sys.argv[1:]] or [300, 700, 1113, 1400, 2000, 2600]:
    src_h, tgt_h = pkg.scans.random_clouds(B, M, seed=B, extent=30.0)
    src = torch.from_numpy(src_h).cuda(); tgt = torch.from_numpy(tgt_h).cuda()
    init = np.zeros((6, 1))
    row = []
    for qb in (-1, 0, 1, 2, 3, 4, 5, 6):
        s = pkg.SVNICP(pkg.SteinICPParam(iterations=1, lr=1.0, max_dist=1.0, KNN_count=K), init)
        if qb < 0: s.set_option("knn", "tiles")      # the Morton-tile chain (incl. its sorts and target copies) for comparison
        else: s.set_option("knn", "brute"); s.set_option("brute_qb", qb)
        s.set_profile(True)
        ts = []
        for _ in range(6):
            s.add_cloud(src, tgt, init); s.stein_align(); ts.append(s.get_kernel_ms()["stage_a_knn"][0])
        row.append("%s %.1f" % ("tiles" if qb < 0 else "auto" if qb == 0 else "qb%d" % qb, 1e3 * float(np.median(ts[2:]))))
    print("B %5d M %d K %d: stage A us  " % (B, M, K) + "  ".join(row), flush=True)

"""N > 1 path on CPU: ranks over torch.distributed/gloo drive svnicp_amd.sharded.ShardedSVNICP with the
test-side oracle backend.  split="particles": every rank must end bit-identical to the unsharded oracle (each
particle's sums are formed on one rank).  split="rows" / 2-D: the ranks must be bit-identical to EACH OTHER (same
records, same order) and within 1e-12 of the unsharded oracle (the sums over source rows are grouped by rank)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, out_dir, split):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as graft
    from oracle_backend import oracle_sharded
    pkg = graft.load_package(); orc = graft.load_oracle()
    orc.set_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svnicp_amd.sharded import ShardedSVNICP, shard_range
    P, B, M, K, I, full, es, thr = case
    src, tgt = pkg.scans.random_clouds(B, M, seed=17)
    init = pkg.scans.make_particles(P, seed=17) * 0.3
    prm = pkg.SteinICPParam(iterations=I, lr=1.0, max_dist=1.0, KNN_count=K, SVN_full_grad=full, check_early_stop=es,
                            convergence_threshold=thr)
    s = oracle_sharded(ShardedSVNICP, orc)(prm, init, split=split)
    assert s.world == world and s.rank == rank
    s.add_cloud(src, tgt, init)
    T = np.eye(4); T[:3, 3] = [0.01, -0.02, 0.005]
    s.set_initial_mean(T)
    s.stein_align()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), particles=s.get_particles(), cov=s.get_cov_matrix(),
             mean=s.get_transformation(), hist=s.get_particle_history(), cand=s.be.solver.candidates(),
             fin=s.be.solver.iterations_run(), shard=np.array(shard_range(P, s.Wp, s.rp)), rows=np.array(shard_range(B, s.Wb, s.rb)))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    (10, 301, 900, 12, 5, False, False, 1e-5),   # even particle split, ragged source rows
    (7, 256, 700, 9, 4, True, False, 1e-5),      # odd particle count: ragged shard + padded all-gather
    (8, 300, 800, 10, 30, True, True, 8e-2),     # early stop decided identically on every rank
]


def _unsharded(pkg, orc, case):
    P, B, M, K, I, full, es, thr = case
    src, tgt = pkg.scans.random_clouds(B, M, seed=17)
    init = pkg.scans.make_particles(P, seed=17) * 0.3
    o = orc.Solver(init, iterations=I, lr=1.0, max_dist=1.0, knn_count=K, svn_full_grad=full, check_early_stop=es,
                   convergence_threshold=thr)
    o.add_cloud(src, tgt, init); o.set_initial_mean(np.eye(3), [0.01, -0.02, 0.005]); o.stein_align()
    return o


@pytest.mark.parametrize("case", CASES)
def test_two_ranks_gloo_equal_unsharded(case, tmp_path, pkg, orc):
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, case, str(tmp_path), "particles"), nprocs=world, join=True, start_method="spawn")
    P, B, M, K, I, full, es, thr = case
    o = _unsharded(pkg, orc, case)
    res = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    shards = [tuple(r["shard"]) for r in res]
    assert shards[0][0] == 0 and shards[-1][1] == P and shards[0][1] == shards[1][0]
    for r in res:
        assert np.array_equal(r["cand"], o.candidates())
        assert np.array_equal(r["particles"], o.get_particles())       # bit-identical to the unsharded run
        assert np.array_equal(r["cov"], o.get_cov_matrix())
        assert np.array_equal(r["hist"], o.get_particle_history())
        assert int(r["fin"]) == o.iterations_run()
    if es:
        assert o.iterations_run() < I


@pytest.mark.parametrize("case,world,split", [(CASES[0], 2, "rows"), (CASES[1], 2, "rows"), (CASES[2], 2, "rows"),
                                              (CASES[0], 3, "rows"), (CASES[0], 4, (2, 2))])
def test_row_sharded_ranks_gloo(case, world, split, tmp_path, pkg, orc):
    """Source rows sharded (and the 2-D split): each rank holds only its rows — candidate lists are the unsharded run's rows,
    replicas are bit-identical to each other, the result is the unsharded one up to the grouping of the sums."""
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, case, str(tmp_path), split), nprocs=world, join=True, start_method="spawn")
    P, B, M, K, I, full, es, thr = case
    o = _unsharded(pkg, orc, case)
    res = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    covered = np.zeros(B, int)
    for r in res:
        lo, hi = r["rows"]
        assert np.array_equal(r["cand"], o.candidates()[lo:hi])            # stage A of the slice = the slice of stage A
        covered[lo:hi] += 1
        assert np.array_equal(r["particles"], res[0]["particles"])        # replicas: same records, same order
        assert np.array_equal(r["cov"], res[0]["cov"]) and np.array_equal(r["hist"], res[0]["hist"])
        np.testing.assert_allclose(r["particles"], o.get_particles(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["cov"], o.get_cov_matrix(), rtol=0, atol=1e-12)
        assert int(r["fin"]) == o.iterations_run()
    Wp = 1 if split == "rows" else split[0]
    assert np.all(covered == Wp)                                           # every row on exactly one row group
    if es:
        assert o.iterations_run() < I


def test_shard_range_covers_everything(pkg):
    from svnicp_amd.sharded import shard_range
    for n in (1, 7, 64, 129):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert all(0 <= lo <= hi <= n for lo, hi in r)

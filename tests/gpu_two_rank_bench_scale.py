"""Rehearsal of `bench.py --gpus 2` on ONE GPU: two processes share cuda:0 (gloo for the collectives), C3 clouds,
256 particles sharded 128 per rank, device-resident inputs as in bench.py.  Checks that the replicas agree bit for
bit and with a single-process run of the same 256 particles; prints the wall time per registration (not a
performance number: both ranks share one GPU and the all-gathers are staged through the host)."""
import os, sys, socket, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, out_dir):
    import __graft_entry__ as graft
    import torch
    import torch.distributed as dist
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svnicp_amd.sharded import ShardedSVNICP
    cfg = pkg.scans.CONFIGS["C3"]
    P = 128 * world
    pair = pkg.scans.make_pair(cfg["B"], cfg["M"]); init = pkg.scans.make_particles(P)
    prm = pkg.SteinICPParam(iterations=20, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False, check_early_stop=False)
    dev = torch.device("cuda", 0)
    src_d, tgt_d = torch.from_numpy(pair.source).to(dev), torch.from_numpy(pair.target).to(dev)
    s = ShardedSVNICP(prm, init, device_index=0)
    for k in range(3):
        dist.barrier(); t0 = time.perf_counter()
        s.add_cloud(src_d, tgt_d, init); s.set_initial_mean(np.eye(4)); s.stein_align()
        mean = s.get_transformation()
        dist.barrier(); dt = time.perf_counter() - t0
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), particles=s.get_particles(), mean=mean, cov=s.get_cov_matrix(), dt=dt)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import __graft_entry__ as graft
    d = tempfile.mkdtemp()
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(worker, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (np.load(os.path.join(d, f"r{r}.npz")) for r in range(2))
    print("replicas bit-identical:", np.array_equal(r0["particles"], r1["particles"]) and np.array_equal(r0["cov"], r1["cov"]))
    pkg = graft.load_package()
    cfg = pkg.scans.CONFIGS["C3"]
    pair = pkg.scans.make_pair(cfg["B"], cfg["M"]); init = pkg.scans.make_particles(256)
    prm = pkg.SteinICPParam(iterations=20, lr=1.0, max_dist=1.0, KNN_count=100, SVN_full_grad=False, check_early_stop=False)
    s = pkg.SVNICP(prm, init, pkg.ParticleWeightOpt(), device=0)
    s.add_cloud(pair.source, pair.target, init); s.set_initial_mean(np.eye(4)); s.stein_align()
    print("max |sharded - single| particles:", np.abs(s.get_particles() - r0["particles"]).max(), " mean:", np.abs(s.get_transformation() - r0["mean"]).max())
    print("wall per sharded registration (2 ranks on one GPU, host-staged gathers): %.1f ms" % (1e3 * float(r0["dt"])))

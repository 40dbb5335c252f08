#!/usr/bin/env python3
"""bench.py — scan-pair registrations/sec of the MI355X Stein-ICP path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload C3]

A *step* is one full registration of one synthetic 64-beam scan pair through the C ABI:
add_cloud (device→device copy of the resident clouds + particle reset) → set_initial_mean →
stein_align (stage A exact top-K + I Stein iterations) → result getters.  The clouds are
resident in HBM before the timed region starts.

N = 1 runs the headline configuration C3 (128 particles × 131 072 source / 262 144 target points,
K = 100, I = 20, float64); `value` = registrations/s.
N > 1 runs BASELINE configuration C4 — 512 particles FIXED — with the SOURCE ROWS sharded B/N per GPU
(`--split rows`, the default: every GPU searches and accumulates its rows for all particles, one all-gather
of N × P × 176 B partial sums per iteration over RCCL; `--split particles` is the 512/N-particles-per-GPU
layout) — so `value` is the raw registrations/s of the same 512-particle registration at every N
(`"scaling": "strong"`).  Next to it the line carries
`speedup_vs_1gpu` (rank 0 times the unsharded 512-particle registration on its own GPU in the same
invocation, outside the timed region) and a `weak_scaling` record (C3 clouds, 128 particles per GPU,
raw registrations/s).  The N = 1 line carries the one-GPU C4 rate as `c4_one_gpu` so that the
strong-scaling curve has its first point.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      — the dominant kernel (k_stein_search_bf16 at C3): algorithmic flops of the reference's brute
                  force per launch time against the f32 matrix/vector peak, from live hipEvent timings on the
                  library's stream; roofline_hbm is the HBM view of the same kernel,
  cpu_baseline  — the CPU oracle ("port": oracle/svnicp_oracle.c, OpenMP) timed on this box's host
                  cores on the same workload (rank 0, N = 1 only),
  svn_full_grad — the same registration with SVNFullGrad = true (BASELINE.md §2 reports both branches),
  svgd_adam     — SVGD-ICP (Adam, lr 0.01) on the same clouds; c4_one_gpu — 512 particles on one GPU;
  other_configs — wall time of whole registrations at C1, C2 and at the size the scan-to-map loop hands to the solver.
The timed region carries no profiling hooks; the roofline block comes from a second pass over the same K steps with
every kernel launch bracketed by hipEvents on the library's stream.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TF = 78.6    # MI355X f64 vector peak (FMA counted as 2 flop); 39.3 T non-fused op/s
F32_MFMA_PEAK_TF = 157.3   # MI355X dense f32-input MFMA peak = f32 vector peak, MI355X_MICROARCH.md
BF16_MFMA_PEAK_TF = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--correspondence", default="fast", choices=("fast", "full"),
                    help="full = the reference's get_correspondence (K = 1 over the whole target per particle per iteration; N = 1 only)")
    ap.add_argument("--workload", default=None, help="C1|C2|C3|C4|C5 (default: C3 at N = 1, C4 = 512 particles fixed at N > 1)")
    ap.add_argument("--full-grad", type=int, default=0, help="SVNFullGrad (shipped default false)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="source points of the CPU-baseline sample, evenly spaced over the scan (-1 = the whole source, 0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent brackets")
    ap.add_argument("--mode", default="svn", choices=("svn", "svgd"), help="svn (headline) | svgd (first-order sibling, N = 1)")
    ap.add_argument("--split", default="rows", choices=("rows", "particles"),
                    help="N > 1: what is cut across the GPUs — source rows (default: nothing of size [B] replicated) or particles")
    return ap.parse_args()


def time_steps(step, n, world, dist, torch, dev):
    """Barrier + synchronize on both sides, max over ranks; returns seconds for n steps."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # rehearsal switch (one-GPU box): SVNICP_BENCH_REHEARSAL=1 puts every rank on cuda:0 and runs the collectives over gloo
    # (host-staged) — it exercises the N > 1 control flow, its numbers mean nothing
    rehearsal = os.environ.get("SVNICP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pkg = graft.load_package()
    scans = pkg.scans
    wl = a.workload or ("C3" if world == 1 else "C4")
    cfg = dict(scans.CONFIGS[wl])
    P, B, M, I = cfg["P"], cfg["B"], cfg["M"], cfg["I"]
    K = 100
    pair = scans.make_pair(B, M)
    init = scans.make_particles(P)
    svgd = a.mode == "svgd"

    def make_param(full_grad):
        return pkg.SteinICPParam(iterations=I, lr=(0.01 if svgd else 1.0), max_dist=1.0, KNN_count=K,
                                 SVN_full_grad=bool(full_grad), check_early_stop=False, optimizer="Adam")
    prm = make_param(a.full_grad)
    src_d = torch.from_numpy(pair.source).to(dev)
    tgt_d = torch.from_numpy(pair.target).to(dev)
    T0 = np.eye(4)

    def make_step(solver, particles):
        def step():
            solver.add_cloud(src_d, tgt_d, particles)
            solver.set_initial_mean(T0)
            st = solver.stein_align()
            return st, solver.get_transformation(), solver.get_cov_matrix()
        return step

    if world == 1:
        solver = (pkg.SVGDICP(prm, init, device=local_rank) if svgd else
                  pkg.SVNICP(prm, init, pkg.ParticleWeightOpt(), device=local_rank))
        if a.correspondence == "full":
            solver.set_option("correspondence", "full")
    else:
        from svnicp_amd.sharded import ShardedSVNICP
        solver = ShardedSVNICP(prm, init, device_index=local_rank, split=a.split)
    step = make_step(solver, init)

    for i in range(a.warmup):
        step()
    last = {}

    def timed_step():
        st, mean, cov = step()
        last["mean"] = mean
    # the timed region: exactly K steps of the product as a caller runs it — no event brackets, no profiling hooks
    el = time_steps(timed_step, a.steps, world, dist, torch, dev)
    mean = last["mean"]

    # roofline region: the SAME K steps once more, every kernel launch bracketed by hipEvents on the library's stream
    # (svnicp_set_profile; ~5 us of stream time per bracket, which is why this pass is not the one `value` comes from)
    kernel_ms = {}
    if world == 1 and not a.no_profile:
        solver.set_profile(True)
        for _ in range(a.steps):
            step()
            for k, (ms, n) in solver.get_kernel_ms().items():
                acc = kernel_ms.setdefault(k, [0.0, 0])
                acc[0] += ms
                acc[1] += n
        solver.set_profile(False)

    pose_err = np.abs(mean - pair.true_pose)
    mode_name = "SVGD-ICP (Adam, lr 0.01)" if svgd else "SVN-ICP"
    out = {
        "metric": "scan-pair registrations/sec at N particles x M source pts, 1/2/4/8 GPU",
        "value": a.steps / el,                      # raw registrations per second of the named workload, whole job
        "unit": "registrations/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * el / a.steps,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else None,   # one GPU: there is nothing to scale
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo — not a measurement)",
        "config": {"workload": f"{wl}: {P} particles {mode_name}, {B}-pt source vs {M}-pt target, K={K}, I={I}, "
                               f"max_dist=1.0, lr={prm.lr}, SVNFullGrad={bool(a.full_grad)}, correspondence={a.correspondence}, early stop off; synthetic 64-beam "
                               f"scans (seed {scans.SEED})",
                   "particles": P, "source_points": B, "target_points": M, "knn_count": K, "iterations": I,
                   "parallelism": "single GPU" if world == 1 else (
                       (f"source rows sharded {B // world}/GPU, all {P} particles on every GPU, candidate search and table per shard; "
                        f"per iteration one all-gather of {world} x {P} x 176 B partial sums, added in rank order on every GPU"
                        if a.split == "rows" else
                        f"{P} particles fixed, sharded {P // world}/GPU; stage A sharded by source rows; one all-gather of 176 "
                        f"B/particle/iteration") + f" (RCCL, world size {dist.get_world_size()} as reported by the process group)")},
        "particle_registrations_per_s": P * a.steps / el,
        "pose_error_vs_planted": {"trans_m": float(pose_err[:3].max()), "rot_rad": float(pose_err[3:].max())},
    }

    nside = max(2, min(5, a.steps))
    if world > 1:
        # strong-scaling reference: the same 512-particle registration, unsharded, on rank 0's GPU (others wait)
        one = None
        if rank == 0:
            ref = pkg.SVNICP(prm, init, pkg.ParticleWeightOpt(), device=local_rank)
            rstep = make_step(ref, init)
            rstep()
            one = time_steps(rstep, nside, 1, dist, torch, dev) / nside
            ref.close()
        dist.barrier()
        # weak-scaling record: C3 clouds, 128 particles per GPU (per-GPU work fixed)
        cfgw = scans.CONFIGS["C3"]
        Pw = 128 * world
        if (cfgw["B"], cfgw["M"]) != (B, M):
            pw = scans.make_pair(cfgw["B"], cfgw["M"])
            src_d = torch.from_numpy(pw.source).to(dev); tgt_d = torch.from_numpy(pw.target).to(dev)
        initw = scans.make_particles(Pw)
        sw = ShardedSVNICP(prm, initw, device_index=local_rank, split=a.split)
        wstep = make_step(sw, initw)
        wstep()
        elw = time_steps(wstep, nside, world, dist, torch, dev)
        if rank == 0:
            out["one_gpu_ms_per_step"] = 1e3 * one
            out["speedup_vs_1gpu"] = one / (el / a.steps)
            out["weak_scaling"] = {"workload": f"C3 clouds, {Pw} particles = 128 per GPU", "registrations_per_s": nside / elw,
                                   "ms_per_step": 1e3 * elw / nside, "particle_registrations_per_s": Pw * nside / elw}

    if world == 1 and kernel_ms:
        # Algorithmic work per launch (SURVEY.md §8d per-unit figures): 8 unfused f64 flop
        # per point-pair distance of the reference's brute force; bytes = clouds/candidates in + results out.
        split = bool(kernel_ms.get("k_stein_search", (0, 0))[1])
        work = {
            # stage A prunes exactly (Morton tiles + f32 pre-filter): its brute-force flop count is not work it does, so
            # only the algorithmic bytes and the brute-force pair count are reported for it
            "stage_a_knn": dict(bytes=24.0 * (B + M) + 12.0 * B * K, pairs_bruteforce=float(B) * M),
            # split stage B: the search kernel reads the float32 candidate rows (16 B each) and the source points and
            # writes the winner's slot byte and target index (5 B) per (point, particle); the accumulate kernel reads
            # the index (4 B), the source points and one winner (24 B) per pair.  Fused kernels: all in k_stein_accumulate.
            "k_stein_search": dict(bytes=16.0 * B * K + 24.0 * B + 5.0 * P * B, flops=8.0 * P * B * K),
            "k_stein_accumulate": dict(bytes=(24.0 * B + 28.0 * P * B) if split else (24.0 * B + 24.0 * B * K),
                                       flops=(60.0 * P * B) if split else 8.0 * P * B * K),
        }
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f).get(wl, {})
        except OSError:
            pass
        details = {}
        for k, (ms, n) in kernel_ms.items():
            if n == 0:
                continue
            d = {"launches_per_registration": n // a.steps, "avg_launch_ms": ms / max(n, 1),
                 "ms_per_registration": ms / a.steps}
            if k in work:
                avg_s = ms / max(n, 1) * 1e-3
                d["alg_GBps"] = work[k]["bytes"] / avg_s / 1e9
                if "flops" in work[k]:
                    d["alg_TFLOPs"] = work[k]["flops"] / avg_s / 1e12
                else:
                    d["pairs_bruteforce"] = work[k]["pairs_bruteforce"]
                    d["note"] = "exact pruned search: the brute-force pair count is the reference's work, not this kernel's"
            details[k] = d
        dom = max((k for k in work if "flops" in work[k]), key=lambda k: kernel_ms.get(k, (0, 0))[0])
        avg_s = kernel_ms[dom][0] / max(kernel_ms[dom][1], 1) * 1e-3
        gbs = work[dom]["bytes"] / avg_s / 1e9
        tf = work[dom]["flops"] / avg_s / 1e12
        tr = traffic.get(dom, {}).get("bytes") if P == 128 else None
        hbm = {"kernel": dom, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "traffic": tr}
        if dom == "k_stein_search":
            # The scores are float32-accurate quantities: 8 algorithmic flop per (candidate, particle) pair of the
            # reference's brute force, priced against the f32 matrix/vector peak (157.3 TF).  They are
            # produced on the bf16 matrix pipe as exact three-way operand splits (K = 32 per tile instead of 4), so the
            # flops the matrix pipe EXECUTES are 8x the algorithmic ones (reported below against the bf16 peak).
            mfma_exec_tf = 2.0 * 16 * 16 * 32 * (B * ((P + 15) // 16) * ((K // 16) if K % 16 < 5 else (K + 15) // 16)) / avg_s / 1e12
            out["roofline"] = {"kernel": "k_stein_search_bf16", "bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TF,
                               "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF, "traffic": tr,
                               "executed_bf16_TFLOPs": mfma_exec_tf, "executed_frac_of_bf16_peak": mfma_exec_tf / BF16_MFMA_PEAK_TF,
                               "note": "nearest-of-K scores on v_mfma_f32_16x16x32_bf16 (exact bf16x3 splits) + about one VALU "
                                       "instruction per score to track them (minimum of each 4-candidate tile, smallest and second smallest "
                                       "tagged tile minimum; the winning tile re-scored by the owner lane); the kernel is bound by "
                                       "vector-instruction issue (3.5 vector instructions per score in all, the vector unit ~80 % busy: PMC in "
                                       "profiles/), not by the matrix pipe or "
                                       "HBM; roofline_hbm is the HBM view; traffic = rocprofv3 FETCH_SIZE*2 + WRITE_SIZE "
                                       "per launch (profiles/traffic.json)"}
            out["roofline_hbm"] = hbm
        else:
            hbm["note"] = ("nearest-neighbour search is VALU-bound, not HBM-bound (SURVEY.md §8d): see roofline_valu; traffic = "
                           "rocprofv3 FETCH_SIZE*2 + WRITE_SIZE per launch (profiles/traffic.json)")
            out["roofline"] = hbm
            out["roofline_valu"] = {"kernel": dom, "bound": "valu_f64", "achieved": tf, "peak": F64_VALU_PEAK_TF,
                                    "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TF,
                                    "note": "algorithmic flops of the reference's f64 brute force (8 per pair) per launch "
                                            "time; the kernels reach them with exact float32 pre-filters / pruning, so "
                                            "this is an effective rate, not an instruction count"}
        out["kernels"] = details

    if world == 1 and not svgd and wl == "C3":
        # side records (few untimed-region steps each): SVNFullGrad = true, the one-GPU point of the C4 strong-scaling curve,
        # and the host cost of the multi-GPU driver (ShardedSVNICP at world size 1 against svnicp_align)
        solver.set_profile(False)
        s2 = pkg.SVNICP(make_param(not a.full_grad), init, pkg.ParticleWeightOpt(), device=local_rank)
        st2 = make_step(s2, init); st2()
        e2 = time_steps(st2, nside, 1, dist, torch, dev)
        out["svn_full_grad" if not a.full_grad else "svn_default_grad"] = {"registrations_per_s": nside / e2, "ms_per_step": 1e3 * e2 / nside}
        s2.close()
        # the first-order sibling (SVGD-ICP, Adam, lr 0.01) on the same clouds, so that the driver's N = 1 line times it too
        sg = pkg.SVGDICP(pkg.SteinICPParam(iterations=I, lr=0.01, max_dist=1.0, KNN_count=K, check_early_stop=False, optimizer="Adam"),
                         init, device=local_rank)
        stg = make_step(sg, init); stg()
        eg = time_steps(stg, nside, 1, dist, torch, dev)
        out["svgd_adam"] = {"workload": f"SVGD-ICP (Adam, lr 0.01), {P} particles, C3 clouds", "registrations_per_s": nside / eg, "ms_per_step": 1e3 * eg / nside}
        sg.close()
        c4 = scans.CONFIGS["C4"]
        init4 = scans.make_particles(c4["P"])
        s4 = pkg.SVNICP(prm, init4, pkg.ParticleWeightOpt(), device=local_rank)
        st4 = make_step(s4, init4); st4()
        e4 = time_steps(st4, nside, 1, dist, torch, dev)
        out["c4_one_gpu"] = {"workload": f"C4 on one GPU: {c4['P']} particles, C3 clouds", "registrations_per_s": nside / e4,
                             "ms_per_step": 1e3 * e4 / nside}
        s4.close()
        # the other one-GPU configurations of BASELINE.json and the size the reference's scan-to-map loop hands to the solver
        # (OdometryPipeline.cpp:559-560: ~1 100 source points after the two samplings, a ~50 000-point local map): wall time
        # of whole registrations, clouds resident in HBM, like the headline
        other = {}
        def side(name, workload, Po, src_np, tgt_np, n, param=None):
            sd, td = torch.from_numpy(src_np).to(dev), torch.from_numpy(tgt_np).to(dev)
            init_o = scans.make_particles(Po)
            so = pkg.SVNICP(param or prm, init_o, pkg.ParticleWeightOpt(), device=local_rank)
            def st():
                so.add_cloud(sd, td, init_o); so.set_initial_mean(T0)
                return so.stein_align(), so.get_transformation(), so.get_cov_matrix()
            for _ in range(3): st()
            eo = time_steps(st, n, 1, dist, torch, dev)
            other[name] = {"workload": workload, "registrations_per_s": n / eo, "ms_per_step": 1e3 * eo / n}
            if param is not None:
                other[name]["iterations_run"] = int(so.get_iterations_run())
            so.close()
        for cname in ("C1", "C2"):
            cc = scans.CONFIGS[cname]
            pr = scans.make_pair(cc["B"], cc["M"])
            side(cname.lower(), f"{cname}: {cc['P']} particle(s), {cc['B']}-pt source vs {cc['M']}-pt target, K={K}, I={I}", cc["P"],
                 pr.source, pr.target, 20)
        from svnicp_amd.pipeline import downsample_uniform, crop_pointcloud
        pr = scans.make_pair(65536, 50000)
        srcc, _ = crop_pointcloud(pr.source, 1.0, 100.0)
        small = downsample_uniform(downsample_uniform(srcc, 0.5), 1.5)
        for Po in (128, 30):
            side(f"scan_to_map_size_p{Po}", f"{Po} particles, {small.shape[0]}-pt source (a 65536-pt scan after the loop's two uniform samplings) "
                 f"vs 50000-pt local map, K={K}, I={I}", Po, small, pr.target, 40)
        # the reference's shipped solver settings (config/geodeAlpha.yaml): 100 iterations, early stop at 5e-4, 10 particles, max_dist 3
        shipped = pkg.SteinICPParam(iterations=100, lr=1.0, max_dist=3.0, KNN_count=K, SVN_full_grad=False, check_early_stop=True,
                                    convergence_threshold=5e-4)
        side("scan_to_map_size_shipped_p10", f"config/geodeAlpha.yaml solver settings: 10 particles, up to 100 iterations with early stop at 5e-4, "
             f"max_dist 3, K={K}; {small.shape[0]}-pt source vs 50000-pt local map", 10, small, pr.target, 40, shipped)
        out["other_configs"] = other
        from svnicp_amd.sharded import ShardedSVNICP
        ss = ShardedSVNICP(prm, init, device_index=local_rank)
        sst = make_step(ss, init); sst()
        es_ = time_steps(sst, nside, 1, dist, torch, dev)
        plain = make_step(solver, init); plain()
        ep = time_steps(plain, nside, 1, dist, torch, dev)
        out["sharded_driver_world1"] = {"ms_per_step": 1e3 * es_ / nside, "svnicp_align_ms_per_step": 1e3 * ep / nside,
                                        "host_overhead_ms": 1e3 * (es_ - ep) / nside}

    if rank == 0 and world == 1 and a.cpu_sample != 0:
        orc = graft.load_oracle()
        Bs = B if a.cpu_sample < 0 else min(a.cpu_sample, B)
        rows = np.linspace(0, B - 1, Bs).astype(np.int64)      # evenly spaced over the scan, not its first beams
        o = orc.Solver(init, mode=(orc.MODE_SVGD if svgd else orc.MODE_SVN), iterations=I, lr=prm.lr, max_dist=1.0,
                       knn_count=K, svn_full_grad=bool(a.full_grad), optimizer="Adam")
        o.add_cloud(np.ascontiguousarray(pair.source[rows]), pair.target, init)
        t1 = time.perf_counter()
        o.stein_align()
        tc = time.perf_counter() - t1
        full = tc * (B / Bs)  # stage A and stage B are both linear in the source size
        sample = (f"the whole {B}-pt source" if Bs == B else f"{Bs} of {B} source points, evenly spaced, scaled x{B / Bs:.1f} "
                  "(cost is linear in the source size)")
        out["cpu_baseline"] = {"value": 1.0 / full, "unit": "registrations/s", "cores": orc.get_threads(), "kind": "port",
                               "sample": f"oracle/svnicp_oracle.c (OpenMP, f64) on {sample} against the full {M}-pt target, "
                                         f"P={P}, K={K}, I={I}: {tc:.2f} s",
                               "host_cpus": os.cpu_count()}
        op = o.get_transformation()
        out["cpu_baseline"]["pose"] = [float(v) for v in op]
        if Bs == B:
            out["cpu_baseline"]["max_abs_pose_diff_vs_gpu"] = float(np.abs(op - mean).max())

    if rank == 0 and world == 1 and a.cpu_sample != 0 and not svgd and wl == "C3":
        # BASELINE.json configs[0] is "the reference CPU path": the reference's solver is a libtorch tensor program, and
        # oracle/torch_restatement.py replays it op for op through the same ATen kernels (einsum / bmm / linalg_solve / median
        # on CPU, float64).  Timed here at C1 on this box's host cores beside the GPU's C1 registration (bounded: ~1-3 s).
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_restatement as tre
        c1 = scans.CONFIGS["C1"]
        pr1 = scans.make_pair(c1["B"], c1["M"])
        init1 = scans.make_particles(c1["P"])
        tp = tre.SteinICPParam(iterations=I, lr=prm.lr, max_dist=1.0, check_early_stop=False, KNN_count=K, SVN_full_grad=False)
        ts = tre.SVNICP(tp, torch.from_numpy(init1))
        ts.add_cloud(torch.from_numpy(pr1.source), torch.from_numpy(pr1.target), torch.from_numpy(init1))
        ts.set_initial_mean(np.eye(3), np.zeros(3))
        t1 = time.perf_counter()
        ts.stein_align()
        tl = time.perf_counter() - t1
        g1 = pkg.SVNICP(prm, init1, pkg.ParticleWeightOpt(), device=local_rank)
        g1.add_cloud(pr1.source, pr1.target, init1); g1.set_initial_mean(T0); g1.stein_align()
        out["cpu_baseline_libtorch_c1"] = {"value": 1.0 / tl, "unit": "registrations/s", "cores": torch.get_num_threads(), "kind": "port",
                                           "sample": f"oracle/torch_restatement.py (the reference's tensor program op for op on libtorch CPU, f64) at C1: "
                                                     f"1 particle, {c1['B']}-pt source vs {c1['M']}-pt target, K={K}, I={I}: {tl:.2f} s",
                                           "max_abs_pose_diff_vs_gpu": float(np.abs(ts.get_transformation().numpy() - g1.get_transformation()).max())}
        g1.close()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

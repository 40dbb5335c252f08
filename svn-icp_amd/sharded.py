"""Multi-GPU registration: one process per GPU, torch.distributed (RCCL over xGMI) for the one exchange
step per iteration the path has.

New functionality — the reference is single process / single GPU.  Two ways to cut
the per-iteration pass (transform → nearest-of-K → Gauss–Newton sums) over W ranks, and their product:

``split="rows"`` (default; the one bench.py --gpus N runs).  Rank r is handed source rows
  [r·ceil(B/W), …) only: its stage A, candidate table and per-iteration search + accumulation cover those
  rows for ALL particles, so nothing of size [B] is replicated or gathered and every [B]-sized cost falls
  by W.  Each rank's 22 sums per particle are a partial record; the ranks all-gather W × P × 22 doubles
  per iteration and every rank adds the W records in rank order (svnicp_iter_update does it on load):
  same values, same order ⇒ bit-identical replicas; against the one-GPU run the sums differ by summation
  order only (≈1e-16 relative).
``split="particles"``.  Rank r owns particles [r·ceil(P/W), …) over all source rows; clouds and candidate
  table are replicated, stage A is sharded by rows with one all-gather of the int32 candidate rows per
  registration, and the per-iteration all-gather carries each particle's complete 22 sums (176 B): every
  sum is formed on one rank, so the result is bit-identical to the one-GPU run.
``split=(Wp, Wb)``.  Both at once: rank = rb·Wp + rp owns particles shard rp of row group rb.

After the all-gather every rank runs the small Stein update redundantly on all P particles
(svnicp_iter_update) — identical inputs and code, so no broadcast or reduction collective is needed.

The compute backend is the HIP library through the split-phase C ABI (HipBackend, made by
``_make_backend``); it raises when libsvnicp_hip.so or a gfx950 device is missing.  The CPU tests
(gloo ranks) exercise the orchestration with a SUBCLASS of their own that overrides ``_make_backend``
(tests/oracle_backend.py) — nothing in this module knows about it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding
from .solver import SVGDICP, SVNICP, SteinICPParam, SteinICPState


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


class _DevView:
    """Zero-copy torch view of library-owned device memory via __cuda_array_interface__."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class HipBackend:
    """Split-phase driver of libsvnicp_hip.so for one rank (include/svnicp_hip.h, 'split-phase entry points')."""

    record_width = 22

    def __init__(self, param: SteinICPParam, init_pose, device_index: int, solver_cls=SVNICP):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.solver = solver_cls(param, init_pose, device=device_index)
        self._L = binding.load_library()
        self._h = self.solver.handle
        # one queue for kernels and collectives: run the library on torch's current stream
        stream = torch.cuda.current_stream(self.device)
        self.solver._check(self._L.svnicp_set_stream(self._h, C.c_void_p(stream.cuda_stream)), "svnicp_set_stream")

    def add_cloud(self, src, tgt, init_pose):
        self.solver.add_cloud(src, tgt, init_pose)
        self.P, self.B, self.K = self.solver._P, self.solver._B, self.solver._K

    def set_initial_mean(self, pose):
        self.solver.set_initial_mean(pose)

    def _chk(self, rc, what):
        self.solver._check(rc, what)

    def set_shard(self, lo, hi):
        self._chk(self._L.svnicp_set_shard(self._h, lo, hi), "svnicp_set_shard")

    def align_begin(self):
        self._chk(self._L.svnicp_align_begin(self._h), "svnicp_align_begin")

    def stage_candidates(self, lo, hi):
        self._chk(self._L.svnicp_stage_candidates(self._h, lo, hi), "svnicp_stage_candidates")

    def candidates_tensor(self):
        ptr = self._L.svnicp_candidates_devptr(self._h)
        return self.torch.as_tensor(_DevView(ptr, (self.B, self.K), "<i4"), device=self.device)

    def build_table(self):
        self._chk(self._L.svnicp_build_candidate_table(self._h), "svnicp_build_candidate_table")

    def iter_accumulate(self, it):
        self._chk(self._L.svnicp_iter_accumulate(self._h, it), "svnicp_iter_accumulate")

    def records_tensor(self):
        ptr = self._L.svnicp_sums_devptr(self._h)
        return self.torch.as_tensor(_DevView(ptr, (self.P, self.record_width), "<f8"), device=self.device)

    def set_row_shard(self, row_rank, row_world, total_rows):
        self._row_world = row_world
        self._chk(self._L.svnicp_set_row_shard(self._h, row_rank, row_world, total_rows), "svnicp_set_row_shard")

    def rank_records_tensor(self):
        """[row_world * P, 22]: the ranks' partial records, slot of row group g = rows [g*P, (g+1)*P)."""
        ptr = self._L.svnicp_rank_sums_devptr(self._h)
        return self.torch.as_tensor(_DevView(ptr, (self._row_world * self.P, self.record_width), "<f8"), device=self.device)

    def iter_update(self, it):
        self._chk(self._L.svnicp_iter_update(self._h, it), "svnicp_iter_update")

    def finish(self):
        self._chk(self._L.svnicp_finish(self._h), "svnicp_finish")

    def stopped(self) -> bool:
        rc = self._L.svnicp_stopped(self._h)   # 1 / 0, or a negative svnicp_status
        self._chk(rc, "svnicp_stopped")
        return rc > 0

    def synchronize(self):
        self.torch.cuda.current_stream(self.device).synchronize()


def _all_gather_rows(dist, group, full, lo, hi, world, rank):
    """In-place all-gather of row blocks of `full` ([n, w]); block r = rows shard_range(n, world, r).
    RCCL (backend "nccl") gathers straight into the library's device buffer.  With a host-only backend
    (gloo: the CPU test, or a single-GPU rehearsal with several ranks on one card) device rows are staged
    through host memory."""
    import torch
    if full.device.type != "cpu" and dist.get_backend(group) != "nccl":
        host = full.cpu()
        _all_gather_rows(dist, group, host, lo, hi, world, rank)
        full.copy_(host)
        return
    n = full.shape[0]
    per = (n + world - 1) // world
    if n == per * world:
        dist.all_gather_into_tensor(full, full[lo:hi].clone(), group=group)  # the clone keeps input and output disjoint
        return
    # ragged tail: gather padded blocks, then scatter the valid rows back
    pad = torch.zeros((per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    pad[: hi - lo] = full[lo:hi]
    stage = torch.empty((world * per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    dist.all_gather_into_tensor(stage, pad, group=group)
    full.copy_(stage[:n])


class ShardedSVNICP:
    """SVNICP with the same call sequence as the single-GPU class, the per-iteration pass cut over a process group."""

    _solver_cls = SVNICP

    def __init__(self, param: SteinICPParam, init_pose, group=None, device_index: int | None = None, split="rows"):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.param = param
        self.stop_poll = 4
        if split == "rows":
            self.Wp, self.Wb = 1, self.world
        elif split == "particles":
            self.Wp, self.Wb = self.world, 1
        else:
            self.Wp, self.Wb = int(split[0]), int(split[1])
            if self.Wp * self.Wb != self.world:
                raise ValueError(f"split {split} does not match the world size {self.world}")
        self.rp, self.rb = self.rank % self.Wp, self.rank // self.Wp
        self.be = self._make_backend(param, init_pose, device_index)
        self.B_total = 0

    def _make_backend(self, param, init_pose, device_index):
        """The compute backend of this rank: the HIP library through the split-phase C ABI — there is no other in the product
        (HipBackend raises without libsvnicp_hip.so or a gfx950 device)."""
        import torch
        if device_index is None:
            device_index = torch.cuda.current_device()
        return HipBackend(param, init_pose, device_index, self._solver_cls)

    def add_cloud(self, src, tgt, init_pose):
        """add_cloud of the reference (SVGDICP.cpp:46-62).  With source rows sharded the backend is handed this rank's row
        slice only (a contiguous view: nothing is copied on the host, and a device tensor stays on the device)."""
        self.B_total = int(src.shape[0])
        if self.Wb > 1:
            if self.B_total < self.Wb:
                raise ValueError("fewer source points than row groups")
            b_lo, b_hi = shard_range(self.B_total, self.Wb, self.rb)
            src = src[b_lo:b_hi]
        self.be.add_cloud(src, tgt, init_pose)

    def set_initial_mean(self, pose):
        self.be.set_initial_mean(pose)

    def _exchange_records(self, p_lo, p_hi):
        be, W = self.be, self.world
        if W == 1:
            return
        if self.Wb == 1:      # complete records of the rank's own particles (ragged shards: padded gather)
            _all_gather_rows(self.dist, self.group, be.records_tensor(), p_lo, p_hi, W, self.rank)
            return
        if be.P % self.Wp:
            raise ValueError("a 2-D split needs the particle count to be a multiple of its particle groups")
        rec = be.rank_records_tensor()       # [Wb * P, 22]; this rank's block starts at row rb * P + p_lo = rank * (P / Wp)
        _all_gather_rows(self.dist, self.group, rec, self.rb * be.P + p_lo, self.rb * be.P + p_hi, W, self.rank)

    def stein_align(self) -> SteinICPState:
        if self._solver_cls is SVGDICP and self.param.optimizer not in ("Adam", "RMSprop", "SGD", "Adagrad"):
            return SteinICPState.NO_OPTIMIZER      # set_optimizer() found none: stein_align returns at once (SVGDICP.cpp:73-75)
        be, W, r = self.be, self.world, self.rank
        p_lo, p_hi = shard_range(be.P, self.Wp, self.rp)
        be.set_shard(p_lo, p_hi)
        be.set_row_shard(self.rb, self.Wb, self.B_total if self.Wb > 1 else be.B)
        be.align_begin()
        if self.Wb == 1:      # rows replicated: stage A is sharded by rows and its int32 result rows are gathered once
            b_lo, b_hi = shard_range(be.B, W, r)
            be.stage_candidates(b_lo, b_hi)
            if W > 1:
                _all_gather_rows(self.dist, self.group, be.candidates_tensor(), b_lo, b_hi, W, r)
        else:                 # rows sharded: this rank's rows are the whole cloud it holds
            be.stage_candidates(0, be.B)
        be.build_table()
        for it in range(int(self.param.iterations)):
            be.iter_accumulate(it)
            self._exchange_records(p_lo, p_hi)
            be.iter_update(it)
            # the early-stop flag lives on the device and later launches return at once when it is set, so the host only
            # looks (a device-to-host copy + stream sync) every few iterations; all ranks see the same flag value because
            # the update runs on identical inputs
            if self.param.check_early_stop and (it % self.stop_poll == self.stop_poll - 1) and be.stopped():
                break
        be.finish()
        be.synchronize()
        return SteinICPState.ALIGN_SUCCESS

    def __getattr__(self, name):  # getters of the reference interface
        if name.startswith("get_"):
            return getattr(self.be.solver, name)
        raise AttributeError(name)


class ShardedSVGDICP(ShardedSVNICP):
    """SVGD-ICP (first-order sibling, SVGDICP.cpp:66-140) with the same splits: the per-particle record is the same 22
    raw sums (slot 4 carries the inlier count of SVGDICP.cpp:404; with rows sharded the gradient is scaled by the WHOLE
    scan's size, SVGDICP.cpp:58), and every rank steps the optimizer of ALL particles redundantly after the all-gather,
    so parameters and optimizer state stay bit-identical across ranks."""
    _solver_cls = SVGDICP

"""Test-side compute backend for svnicp_amd.sharded.ShardedSVNICP: the CPU oracle in split-phase
form.  Lives in tests/ on purpose — the product package never constructs it (it would be a CPU
fallback); it exists so that the multi-rank orchestration (partitioning, the two all-gathers, the
redundant update, early stop) can be exercised with torch.distributed/gloo on a box without GPUs."""
import ctypes as C

import numpy as np
import torch


class OracleBackend:
    record_width = 42  # H (36) | b (6) per particle

    def __init__(self, orc, param, init_pose):
        self.orc = orc
        self.L = orc.lib()
        self.solver = orc.Solver(init_pose, iterations=param.iterations, lr=param.lr, max_dist=param.max_dist,
                                 check_early_stop=param.check_early_stop,
                                 convergence_threshold=param.convergence_threshold, knn_count=param.KNN_count,
                                 svn_full_grad=param.SVN_full_grad)
        self._stop = False

    def add_cloud(self, src, tgt, init_pose):
        self.solver.add_cloud(src, tgt, init_pose)
        self.P, self.B, self.K = self.solver.P, self.solver.B, self.solver.K

    def set_initial_mean(self, pose):
        T = np.asarray(pose, np.float64).reshape(4, 4)
        self.solver.set_initial_mean(T[:3, :3], T[:3, 3])

    def set_shard(self, lo, hi):
        self.p_lo, self.p_hi = lo, hi

    def set_row_shard(self, row_rank, row_world, total_rows):
        """Source rows sharded: this solver holds a slice of the scan; its H, b are partial records."""
        self.row_rank, self.row_world = row_rank, row_world

    def align_begin(self):
        self.L.orc_sp_begin(self.solver.h)
        self._stop = False
        self._cand32 = np.zeros((self.B, self.K), np.int32)
        self._rec = np.zeros((self.P, self.record_width), np.float64)
        self._rank_rec = np.zeros((getattr(self, "row_world", 1) * self.P, self.record_width), np.float64)

    def stage_candidates(self, lo, hi):
        self.L.orc_sp_candidate_rows(self.solver.h, lo, hi)
        c64 = np.ctypeslib.as_array(self.L.orc_sp_candidates(self.solver.h), shape=(self.B, self.K))
        self._cand32[lo:hi] = c64[lo:hi]

    def candidates_tensor(self):
        return torch.from_numpy(self._cand32)

    def build_table(self):
        c64 = np.ctypeslib.as_array(self.L.orc_sp_candidates(self.solver.h), shape=(self.B, self.K))
        c64[:] = self._cand32
        self.L.orc_sp_build_table(self.solver.h)

    def iter_accumulate(self, it):
        if self._stop:
            return
        if getattr(self, "row_world", 1) > 1:
            own = self._rank_rec[self.row_rank * self.P:(self.row_rank + 1) * self.P]
            self.L.orc_sp_accumulate_rows(self.solver.h, it, self.p_lo, self.p_hi, own.ctypes.data_as(C.POINTER(C.c_double)), 0)
            return
        self.L.orc_sp_accumulate(self.solver.h, it, self.p_lo, self.p_hi,
                                 self._rec.ctypes.data_as(C.POINTER(C.c_double)))

    def records_tensor(self):
        return torch.from_numpy(self._rec)

    def rank_records_tensor(self):
        return torch.from_numpy(self._rank_rec)

    def iter_update(self, it):
        if self._stop:
            return
        if getattr(self, "row_world", 1) > 1:   # the records in rank order, then the damping (SVNICP.cpp:153), as the HIP update does
            r = self._rank_rec.reshape(self.row_world, self.P, self.record_width)
            acc = r[0].copy()
            for g in range(1, self.row_world):
                acc += r[g]
            for i in range(6):
                acc[:, 7 * i] += 1e-6
            self._rec[:] = acc
        self._stop = bool(self.L.orc_sp_update(self.solver.h, it, self._rec.ctypes.data_as(C.POINTER(C.c_double))))

    def finish(self):
        self.L.orc_sp_finish(self.solver.h)

    def stopped(self):
        return self._stop

    def synchronize(self):
        pass


def oracle_sharded(base_cls, orc):
    """A subclass of svnicp_amd.sharded.ShardedSVNICP whose ranks compute with the CPU oracle (test infrastructure: the
    product class only ever builds the HIP backend)."""
    class OracleSharded(base_cls):
        def _make_backend(self, param, init_pose, device_index):
            return OracleBackend(orc, param, init_pose)
    return OracleSharded

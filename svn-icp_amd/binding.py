"""ctypes binding of libsvnicp_hip.so (C ABI: include/svnicp_hip.h).  No torch types cross it."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB_PATH = os.path.join(_HERE, "libsvnicp_hip.so")
_HEADER = os.path.join(_ROOT, "include", "svnicp_hip.h")


class SvnIcpError(RuntimeError):
    pass


class Params(C.Structure):
    """struct svnicp_params (include/svnicp_hip.h)."""
    _fields_ = [("struct_size", C.c_int32), ("mode", C.c_int32), ("iterations", C.c_int32),
                ("knn_count", C.c_int32), ("lr", C.c_double), ("max_dist", C.c_double),
                ("convergence_threshold", C.c_double), ("check_early_stop", C.c_int32),
                ("svn_full_grad", C.c_int32), ("optimizer", C.c_int32), ("record_trace", C.c_int32)]


def library_path() -> str:
    return _LIB_PATH


def declared_symbols() -> list[str]:
    """Every function name include/svnicp_hip.h declares (used by the symbol-export test)."""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svnicp_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load_library():
    """dlopen libsvnicp_hip.so; raises SvnIcpError (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SvnIcpError(f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch wheels bundle their own libamdhip64; if the system HIP runtime gets loaded first (through
    # this library), a later `import torch` binds to it and torch.cuda reports "No HIP GPUs".  Import
    # torch first so that the whole process shares ONE runtime (torch is only plumbing here: device
    # memory, streams, torch.distributed — the library itself has no torch dependency).
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    try:
        L = C.CDLL(_LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SvnIcpError(f"cannot load {_LIB_PATH}: {e}") from e
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.svnicp_abi_version.restype = C.c_int
    L.svnicp_last_error.restype = C.c_char_p
    L.svnicp_last_error.argtypes = [vp]
    L.svnicp_create.argtypes = [C.POINTER(Params), C.c_int, dp, C.c_int, C.POINTER(vp)]
    L.svnicp_destroy.argtypes = [vp]
    L.svnicp_destroy.restype = None
    L.svnicp_set_stream.argtypes = [vp, vp]
    L.svnicp_synchronize.argtypes = [vp]
    L.svnicp_set_clouds.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int]
    L.svnicp_set_particles.argtypes = [vp, dp, C.c_int]
    L.svnicp_set_source.argtypes = [vp, vp, C.c_int64, C.c_int]
    L.svnicp_set_target.argtypes = [vp, vp, C.c_int64, C.c_int]
    L.svnicp_map_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int64, C.POINTER(vp)]
    L.svnicp_map_destroy.argtypes = [vp]
    L.svnicp_map_destroy.restype = None
    L.svnicp_map_last_error.argtypes = [vp]
    L.svnicp_map_last_error.restype = C.c_char_p
    L.svnicp_map_clear.argtypes = [vp]
    L.svnicp_map_size.argtypes = [vp, C.POINTER(C.c_int64)]
    L.svnicp_map_add_cloud.argtypes = [vp, vp, C.c_int64, C.c_int, dp, dp]
    L.svnicp_map_query.argtypes = [vp, dp, C.c_double, C.POINTER(C.c_int64)]
    L.svnicp_map_points_devptr.argtypes = [vp]
    L.svnicp_map_points_devptr.restype = vp
    L.svnicp_map_download.argtypes = [vp, dp, C.c_int64, C.POINTER(C.c_int64)]
    L.svnicp_prep_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.svnicp_prep_destroy.argtypes = [vp]
    L.svnicp_prep_destroy.restype = None
    L.svnicp_prep_last_error.argtypes = [vp]
    L.svnicp_prep_last_error.restype = C.c_char_p
    L.svnicp_prep_scan.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, dp, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.svnicp_map_skipped_points.argtypes = [vp, C.POINTER(C.c_int64)]
    for name in ("svnicp_prep_cropped_devptr", "svnicp_prep_map_cloud_devptr", "svnicp_prep_source_devptr", "svnicp_prep_source_f32_devptr"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = vp
    L.svnicp_prep_download.argtypes = [vp, C.c_int, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.svnicp_set_initial_mean.argtypes = [vp, dp, dp]
    L.svnicp_set_k.argtypes = [vp, C.c_int]
    L.svnicp_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.svnicp_set_max_dist.argtypes = [vp, C.c_double]
    L.svnicp_align.argtypes = [vp]
    L.svnicp_align_async.argtypes = [vp]
    for n in ("transformation", "distribution", "cov_matrix", "particles", "particle_weight", "runtime",
              "candidate_dist2", "gpu_ms"):
        getattr(L, "svnicp_get_" + n).argtypes = [vp, dp]
    L.svnicp_get_particle_history.argtypes = [vp, fp]
    L.svnicp_get_candidates.argtypes = [vp, ip]
    L.svnicp_get_trace.argtypes = [vp, ip, dp, dp, dp, dp, dp]
    L.svnicp_get_knn_fallbacks.argtypes = [vp, C.POINTER(C.c_int)]
    L.svnicp_get_knn_survivors.argtypes = [vp, ip]
    L.svnicp_get_knn_fallback_rows.argtypes = [vp, ip, C.c_int, C.POINTER(C.c_int)]
    L.svnicp_get_ambiguous_steps.argtypes = [vp, C.POINTER(C.c_int)]
    L.svnicp_get_ambiguous_pairs.argtypes = [vp, C.POINTER(C.c_int64)]
    L.svnicp_get_iterations_run.argtypes = [vp, C.POINTER(C.c_int)]
    L.svnicp_set_profile.argtypes = [vp, C.c_int]
    L.svnicp_get_kernel_ms.argtypes = [vp, dp, ip]
    L.svnicp_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.svnicp_set_row_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int64]
    L.svnicp_rank_sums_devptr.argtypes = [vp]
    L.svnicp_rank_sums_devptr.restype = vp
    L.svnicp_align_begin.argtypes = [vp]
    L.svnicp_stage_candidates.argtypes = [vp, C.c_int64, C.c_int64]
    L.svnicp_build_candidate_table.argtypes = [vp]
    L.svnicp_iter_accumulate.argtypes = [vp, C.c_int]
    L.svnicp_iter_update.argtypes = [vp, C.c_int]
    L.svnicp_finish.argtypes = [vp]
    L.svnicp_stopped.argtypes = [vp]
    L.svnicp_candidates_devptr.argtypes = [vp]
    L.svnicp_candidates_devptr.restype = vp
    L.svnicp_sums_devptr.argtypes = [vp]
    L.svnicp_sums_devptr.restype = vp
    for name in declared_symbols():
        getattr(L, name)  # AttributeError here = the header declares a symbol the library does not export
    _lib = L
    return L


def abi_version() -> int:
    return int(load_library().svnicp_abi_version())

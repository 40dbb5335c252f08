"""svnicp_amd — MI355X-native Stein-variational ICP registration (host-side Python mirror).

The directory is called ``svn-icp_amd`` (not an importable identifier); load it with
``__graft_entry__.load_package()`` which registers it as the module ``svnicp_amd``.

Product code lives here and in ``csrc/`` only.  Nothing in this package imports ``oracle/``:
the compute path is libsvnicp_hip.so (hand-written HIP for gfx950) and it fails loudly when that
library or a gfx950 device is missing — there is no CPU fallback.
"""
from .binding import (SvnIcpError, abi_version, library_path, load_library, declared_symbols)  # noqa: F401
from .solver import SVNICP, SVGDICP, SteinICPParam, SteinICPState, ParticleWeightOpt, initialize_particles  # noqa: F401
from . import scans  # noqa: F401
from . import pipeline, stein_msgs  # noqa: F401  (caller glue and wire formats, SURVEY.md §8(f)-1,2)

__all__ = ["SVNICP", "SVGDICP", "SteinICPParam", "SteinICPState", "ParticleWeightOpt", "initialize_particles",
           "SvnIcpError", "abi_version", "library_path", "load_library", "declared_symbols", "scans", "pipeline", "stein_msgs"]

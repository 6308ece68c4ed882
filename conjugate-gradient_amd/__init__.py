"""MI355X-native (gfx950) fp64 conjugate-gradient hot path.

The work is done by hand-written HIP kernels in csrc/ behind the C ABI of
include/cg_hip.h (libcg_hip.so, built in-tree by build.py); this package is the
host-side mirror used by the tests and the benchmark.  The directory name has a
hyphen, so import it with importlib.import_module("conjugate-gradient_amd").
"""
from . import lib
from .lib import (CGError, REDUCE_FAST, REDUCE_STRICT, SCHED_CLASSIC, SCHED_FUSED, SCHED_FUSED_NOQ, SCHED_LEAN,
                  VEC_U, VEC_G, VEC_D, VEC_Q, SCHED_BYTES_PER_POINT, SCHED_NAMES, device_count,
                  STRICT_AUTO, STRICT_REPLAY, STRICT_PARALLEL)
from .solver import (CGSolver, DeviceVector, SolveResult, init_d, init_g, stencil, dot, axpy, xpay, stencil_noalloc,
                     verify_u, init_d_block, stencil_block, pack_column)

__all__ = ["lib", "CGError", "CGSolver", "DeviceVector", "SolveResult", "init_d", "init_g", "stencil", "dot",
           "axpy", "xpay", "stencil_noalloc", "verify_u", "init_d_block", "stencil_block", "pack_column", "device_count", "REDUCE_FAST", "REDUCE_STRICT", "SCHED_CLASSIC", "SCHED_FUSED",
           "SCHED_FUSED_NOQ", "SCHED_LEAN", "STRICT_AUTO", "STRICT_REPLAY", "STRICT_PARALLEL", "VEC_U", "VEC_G", "VEC_D", "VEC_Q", "SCHED_BYTES_PER_POINT", "SCHED_NAMES"]

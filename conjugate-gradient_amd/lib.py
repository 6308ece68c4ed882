"""ctypes binding of libcg_hip.so, the C ABI declared in include/cg_hip.h.

The shared library is built in-tree by build.py (hipcc --offload-arch=gfx950).
There is no CPU fallback: if the library is missing, loading fails loudly, and
every compute entry point returns an error when no GPU is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CG_HIP_LIB") or os.path.join(_HERE, "libcg_hip.so")   # override: kernel experiments only

# status codes / enums of include/cg_hip.h
CG_OK = 0
REDUCE_FAST, REDUCE_STRICT = 0, 1
SCHED_CLASSIC, SCHED_FUSED, SCHED_FUSED_NOQ, SCHED_LEAN = 0, 1, 2, 3
OPT_REDUCE, OPT_SCHEDULE, OPT_TILE_H, OPT_GRAPH_ITERS, OPT_XCD_REMAP, OPT_FLAT_BLOCKS, OPT_PREFETCH = 1, 2, 3, 4, 5, 6, 7
OPT_STRICT_IMPL = 8
OPT_TILE_WGS = 9
OPT_PREFETCH_READ = 11
OPT_STRICT_BLOCKS = 12
OPT_OVERLAP = 13
OPT_COMM_REHEARSAL = 14
OPT_RESUM_Q0 = 19
OPT_STRICT_CHAIN, OPT_STRICT_FAULT_EVERY, OPT_STRICT_FAULTS = 20, 21, 22
OPT_STRICT_SPEC = 15
OPT_STRICT_MISSES = 16
OPT_STRICT_EVENTS = 17
OPT_STRICT_AUDIT = 18
STRICT_AUTO, STRICT_REPLAY, STRICT_PARALLEL = 1, 2, 3
VEC_U, VEC_G, VEC_D, VEC_Q = 0, 1, 2, 3

# Algorithmic HBM bytes per grid point per iteration of each schedule
# (fp64 accesses x 8): stencil kernel + update kernel (+ d-update kernel).
# (LEAN: R d,g | R d,g W d',g' | every second iteration R u W u)
SCHED_BYTES_PER_POINT = {SCHED_CLASSIC: 16 + 48 + 24, SCHED_FUSED: 32 + 48, SCHED_FUSED_NOQ: 24 + 40,
                         SCHED_LEAN: 16 + 32 + 8}
SCHED_NAMES = {SCHED_CLASSIC: "classic", SCHED_FUSED: "fused", SCHED_FUSED_NOQ: "fused_noq", SCHED_LEAN: "lean"}


class CGError(RuntimeError):
    pass


class Result(C.Structure):
    _fields_ = [("iters", C.c_long), ("q1", C.c_double), ("q_init", C.c_double),
                ("seconds", C.c_double), ("converged", C.c_int)]


_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/cg_hip.h declares
SIGNATURES = {
    "cg_last_error": (C.c_char_p, []),
    "cg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cg_malloc": (C.c_int, [C.c_int, C.c_long, C.POINTER(C.c_void_p)]),
    "cg_free": (C.c_int, [C.c_int, C.c_void_p]),
    "cg_copy_to_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_long]),
    "cg_copy_to_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_long]),
    "cg_init_d": (C.c_int, [C.c_int, C.c_long, C.c_long, C.c_long, C.c_void_p]),
    "cg_init_g": (C.c_int, [C.c_int, C.c_long, C.c_void_p, C.c_void_p]),
    "cg_stencil": (C.c_int, [C.c_int, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, _dp]),
    "cg_stencil9": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, _dp]),
    "cg_dot": (C.c_int, [C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_double, _dp]),
    "cg_init_d_block": (C.c_int, [C.c_int, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p]),
    "cg_stencil_block": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_stencil_block9": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_pack_column": (C.c_int, [C.c_int, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p]),
    "cg_verify_u": (C.c_int, [C.c_int, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_double, _dp, _dp]),
    "cg_stencil_noalloc": (C.c_int, [C.c_int, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_axpy": (C.c_int, [C.c_int, C.c_long, C.c_double, C.c_void_p, C.c_void_p]),
    "cg_xpay": (C.c_int, [C.c_int, C.c_long, C.c_double, C.c_void_p, C.c_void_p]),
    "cg_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int]),
    "cg_destroy": (C.c_int, [C.c_void_p]),
    "cg_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_long]),
    "cg_set_stencil": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cg_get_option": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_long)]),
    "cg_init_problem": (C.c_int, [C.c_void_p]),
    "cg_solve": (C.c_int, [C.c_void_p, C.c_long, C.c_double, C.POINTER(Result)]),
    "cg_sync_vectors": (C.c_int, [C.c_void_p]),
    "cg_copy_out": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_long]),
    "cg_device_ptr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cg_get_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]),
    "cg_time_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp]),
    "cg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cg_comm_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cg_comm_init_self": (C.c_int, [C.c_void_p]),
    "cg_comm_init_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_comm_selftest": (C.c_int, [C.c_void_p]),
}

# callback prototypes of cg_comm_init_host (all pointers are host pointers)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p)

_lib = None


def load():
    """Load libcg_hip.so (once) and attach the prototypes.  Raises CGError if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CGError(f"{LIB_PATH} is missing: build it with `python -m build_cg` or "
                      f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                      f"There is no CPU fallback for the CG hot path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != CG_OK:
        raise CGError(f"libcg_hip error {rc}: {load().cg_last_error().decode(errors='replace')}")


def device_count():
    n = C.c_int(0)
    check(load().cg_device_count(C.byref(n)))
    return n.value

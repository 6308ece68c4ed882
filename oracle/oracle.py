"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The CPU restatement of the reference's arithmetic (oracle/cg_oracle.c).  Only
tests, the smoke check and the benchmark's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("cg_oracle.c", "cg_oracle.h", "cg_oracle_main.c")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.oracle_init_d.argtypes = [C.c_long, C.c_long, C.c_long, _dp]
        lib.oracle_init_g.argtypes = [C.c_long, _dp, _dp]
        lib.oracle_stencil.argtypes = [C.c_long, C.c_long, C.c_long, _dp, C.c_void_p, C.c_void_p, _dp]
        lib.oracle_dot.argtypes = [C.c_long, _dp, _dp]
        lib.oracle_dot.restype = C.c_double
        lib.oracle_dot_from.argtypes = [C.c_double, C.c_long, _dp, _dp]
        lib.oracle_dot_from.restype = C.c_double
        lib.oracle_axpy.argtypes = [C.c_long, C.c_double, _dp, _dp]
        lib.oracle_xpay.argtypes = [C.c_long, C.c_double, _dp, _dp]
        for name in ("oracle_cg", "oracle_cg_rowblocks"):
            fn = getattr(lib, name)
            fn.restype = C.c_long
        lib.oracle_cg.argtypes = [C.c_long, C.c_long, C.c_double, _dp, _dp, _dp, _dp, C.c_void_p, C.c_void_p]
        lib.oracle_cg_rowblocks.argtypes = [C.c_long, C.c_long, C.c_long, C.c_double, _dp, _dp, _dp, _dp,
                                            C.c_void_p, C.c_void_p]
        lib.oracle_dot_blocks2d.argtypes = [C.c_long, C.c_long, _dp, _dp]
        lib.oracle_dot_blocks2d.restype = C.c_double
        lib.oracle_cg_blocks2d.argtypes = lib.oracle_cg_rowblocks.argtypes
        lib.oracle_cg_blocks2d.restype = C.c_long
        _lib = lib
    return _lib


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


DEFAULT_STENCIL = (-1, -1, 4, -1, -1)


def set_stencil(c5=DEFAULT_STENCIL):
    """Coefficients of every later stencil evaluation; process-wide.

    Five values: (up, left, centre, right, down), the entries the reference reads.  Nine values: the
    whole 3x3 array row-wise; the corner terms have no counterpart in the reference (parity unpinned)."""
    c = [int(x) for x in c5]
    if len(c) == 9:
        load().oracle_set_stencil9((C.c_int * 9)(*c))
    else:
        load().oracle_set_stencil((C.c_int * 5)(*c))


def init_d(n, row0=0, rows=None):
    G = n + 1
    rows = G if rows is None else rows
    d = np.empty((rows, G))
    load().oracle_init_d(n, row0, rows, d)
    return d


def init_g(d):
    g = np.empty_like(d)
    load().oracle_init_g(d.size, d, g)
    return g


def stencil(d, row0=0, G=None, top=None, bot=None, q=None):
    rows, cols = d.shape
    G = cols if G is None else G
    q = np.zeros_like(d) if q is None else q
    top = None if top is None else np.ascontiguousarray(top)
    bot = None if bot is None else np.ascontiguousarray(bot)
    load().oracle_stencil(G, row0, rows, d, _opt(top), _opt(bot), q)
    return q


def dot(v, w, start=0.0):
    return load().oracle_dot_from(start, v.size, np.ascontiguousarray(v), np.ascontiguousarray(w))


def axpy(a, x, y):
    load().oracle_axpy(y.size, a, x, y)


def xpay(b, d, g):
    load().oracle_xpay(d.size, b, d, g)


def dot_blocks2d(v, w, s):
    """v.w in the summation layout of the reference on s*s ranks (v, w: G x G arrays)."""
    return load().oracle_dot_blocks2d(v.shape[0], s, np.ascontiguousarray(v), np.ascontiguousarray(w))


def cg(n, max_iters, rtol=0.0, nblocks=None, blocks2d=None):
    """Returns dict(iters, q1, q_init, u, g, d, q, hist).

    nblocks: row blocks with chained sums (single-rank bits); blocks2d = s: the reference's
    own s*s-rank summation layout."""
    G = n + 1
    u, g, d, q = (np.empty((G, G)) for _ in range(4))
    hist = np.zeros(max(max_iters, 1))
    qi = C.c_double(0.0)
    if blocks2d is not None:
        it = load().oracle_cg_blocks2d(n, blocks2d, max_iters, rtol, u, g, d, q, _opt(hist), C.addressof(qi))
    elif nblocks is None:
        it = load().oracle_cg(n, max_iters, rtol, u, g, d, q, _opt(hist), C.addressof(qi))
    else:
        it = load().oracle_cg_rowblocks(n, nblocks, max_iters, rtol, u, g, d, q, _opt(hist), C.addressof(qi))
    if it < 0:
        raise ValueError("oracle: grid side not divisible by the block count (or unsupported rank count)")
    return dict(iters=it, q1=hist[it - 1] if it > 0 else qi.value, q_init=qi.value, u=u, g=g, d=d, q=q,
                hist=hist[:it].copy())

"""Host-side mirror of the CG hot path: thin Python objects over the C ABI.

`CGSolver` is the body of the reference's main() between set-up and output
(cg.c:41-134) with the vectors resident in HBM; `DeviceVector` plus the
module-level functions mirror the stand-alone operators (apply_stencil_serial,
dot, the three update loops) on caller-owned device arrays.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import lib as _l


@dataclass
class SolveResult:
    iters: int
    q1: float            # last g.g, the reference's "norm_g^2" (cg.c:139)
    q_init: float        # g0.g0 (cg.c:55)
    seconds: float       # the reference's timed region (cg.c:50-51,131-132)
    converged: bool


class DeviceVector:
    """`count` doubles in the HBM of `device` (zero-filled, like calloc at cg.c:44-45)."""

    def __init__(self, count, device=0):
        self.count, self.device = int(count), int(device)
        p = C.c_void_p()
        _l.check(_l.load().cg_malloc(self.device, self.count, C.byref(p)))
        self.ptr = p

    @classmethod
    def from_host(cls, arr, device=0):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        v = cls(a.size, device)
        _l.check(_l.load().cg_copy_to_device(device, v.ptr, a.ctypes.data_as(C.c_void_p), a.size))
        return v

    def copy_from_host(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if a.size != self.count:
            raise ValueError(f"{a.size} values for a vector of {self.count}")
        _l.check(_l.load().cg_copy_to_device(self.device, self.ptr, a.ctypes.data_as(C.c_void_p), a.size))
        return self

    def to_host(self):
        out = np.empty(self.count, dtype=np.float64)
        _l.check(_l.load().cg_copy_to_host(self.device, out.ctypes.data_as(C.c_void_p), self.ptr, self.count))
        return out

    def offset(self, elems):
        """Device pointer `elems` doubles into the vector."""
        return C.c_void_p(self.ptr.value + 8 * int(elems))

    def free(self):
        if self.ptr:
            _l.load().cg_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _p(x):
    if x is None:
        return None
    return x.ptr if isinstance(x, DeviceVector) else x


def init_d(n, row0, rows, d, device=0):
    """d0 on global rows [row0, row0+rows): init_locald's fill loop (utils.c:319-337)."""
    _l.check(_l.load().cg_init_d(device, n, row0, rows, _p(d)))


def init_g(count, d, g, device=0):
    """g0 = -d0: init_localg (utils.c:342-350)."""
    _l.check(_l.load().cg_init_g(device, count, _p(d), _p(g)))


def _stencil9(coef):
    """(up, left, centre, right, down), or all nine entries row-wise -> the 3x3 integer array of a stencil_struct (cg.c:33)."""
    c = [int(x) for x in coef]
    if len(c) == 9:
        return (C.c_int * 9)(*c)
    up, left, centre, right, down = c
    return (C.c_int * 9)(0, up, 0, left, centre, right, 0, down, 0)


def stencil(G, row0, rows, d, q, top=None, bot=None, device=0, want_dq=False, coef5=None):
    """q = A d on a row block: apply_stencil_serial / the row part of apply_stencil_parallel."""
    dq = C.c_double(0.0)
    if coef5 is None:
        _l.check(_l.load().cg_stencil(device, G, row0, rows, _p(d), _p(top), _p(bot), _p(q),
                                      C.byref(dq) if want_dq else None))
    else:
        _l.check(_l.load().cg_stencil9(device, _stencil9(coef5), G, row0, rows, _p(d), _p(top), _p(bot), _p(q),
                                       C.byref(dq) if want_dq else None))
    return dq.value if want_dq else None


def init_d_block(n, row0, col0, rows, cols, d, device=0):
    """d0 on one chunk of the reference's two-dimensional decomposition (utils.c:319-337)."""
    _l.check(_l.load().cg_init_d_block(device, n, row0, col0, rows, cols, _p(d)))


def stencil_block(rows, cols, d, q, top=None, bot=None, left=None, right=None, device=0, coef5=(-1, -1, 4, -1, -1), corners=None):
    """q = A d on a rows x cols chunk with its four pads (None = global boundary): apply_stencil_parallel, utils.c:38-219.
    corners: device vector of the four values diagonally across the chunk's corners (nine-point coefficients only)."""
    if corners is None:
        _l.check(_l.load().cg_stencil_block(device, _stencil9(coef5), rows, cols, _p(d), _p(top), _p(bot), _p(left), _p(right), _p(q)))
    else:
        _l.check(_l.load().cg_stencil_block9(device, _stencil9(coef5), rows, cols, _p(d), _p(top), _p(bot), _p(left), _p(right),
                                             _p(corners), _p(q)))


def pack_column(rows, cols, src, col, dst, device=0):
    """dst[r] = src[r*cols + col]: the sending side of the reference's strided column datatype (utils.c:392)."""
    _l.check(_l.load().cg_pack_column(device, rows, cols, _p(src), col, _p(dst)))


def stencil_noalloc(G, row0, rows, d, q, top=None, bot=None, device=0, dq=None, stream=None):
    """cg_stencil without any allocation: enqueued on `stream` (None = the null stream), d.q to the device vector `dq`."""
    _l.check(_l.load().cg_stencil_noalloc(device, G, row0, rows, _p(d), _p(top), _p(bot), _p(q), _p(dq), stream))


def verify_u(n, row0, rows, u, sum_start=0.0, device=0):
    """(sequential sum of the block of u continued from sum_start, max |u - x(1-x)y(1-y)|), both formed on the device."""
    s, e = C.c_double(0.0), C.c_double(0.0)
    _l.check(_l.load().cg_verify_u(device, n, row0, rows, _p(u), float(sum_start), C.byref(s), C.byref(e)))
    return s.value, e.value


def dot(count, v, w, strict=False, start=0.0, device=0):
    """sum v[k]*w[k]: the local part of dot (utils.c:368-379)."""
    r = C.c_double(0.0)          # strict: False/0 fast order, True/1 auto, 2 literal replay, 3 parallel evaluation
    _l.check(_l.load().cg_dot(device, count, _p(v), _p(w), int(strict), start, C.byref(r)))
    return r.value


def axpy(count, a, x, y, device=0):
    """y += a*x (cg.c:63-76)."""
    _l.check(_l.load().cg_axpy(device, count, a, _p(x), _p(y)))


def xpay(count, b, d, g, device=0):
    """d = b*d - g (cg.c:81-87)."""
    _l.check(_l.load().cg_xpay(device, count, b, _p(d), _p(g)))


class CGSolver:
    """One row block of the CG solve on one GPU.

    Single GPU: CGSolver(n).  Multi-GPU: rank r of P owns rows
    [r*G/P, (r+1)*G/P) and calls comm_init with the id rank 0 generated.
    """

    def __init__(self, n, device=0, row0=0, rows=None, rank=0, nranks=1):
        self.n, self.G = int(n), int(n) + 1
        self.device, self.rank, self.nranks = int(device), int(rank), int(nranks)
        self.row0 = int(row0)
        self.rows = self.G if rows is None else int(rows)
        self.count = self.rows * self.G
        self._lib = _l.load()
        ctx = C.c_void_p()
        _l.check(self._lib.cg_create(C.byref(ctx), self.device, self.n, self.row0, self.rows, self.rank, self.nranks))
        self._ctx = ctx
        self.reduce, self.schedule = _l.REDUCE_FAST, None          # None = the library picks (LEAN; FUSED_NOQ if HBM is short)

    # -- configuration ---------------------------------------------------
    def set_option(self, key, value):
        _l.check(self._lib.cg_set_option(self._ctx, key, int(value)))

    def configure(self, reduce=None, schedule=None, tile_h=None, graph_iters=None, xcd_remap=None,
                  flat_blocks=None, prefetch=None, strict_impl=None, tile_wgs=None, prefetch_read=None, strict_blocks=None,
                  overlap=None, comm_rehearsal=None, strict_spec=None):
        if reduce is not None:
            self.set_option(_l.OPT_REDUCE, reduce)
            self.reduce = reduce
        if schedule is not None:
            self.set_option(_l.OPT_SCHEDULE, schedule)
            self.schedule = schedule if schedule >= 0 else None
        if tile_h is not None:
            self.set_option(_l.OPT_TILE_H, tile_h)
        if graph_iters is not None:
            self.set_option(_l.OPT_GRAPH_ITERS, graph_iters)
        if xcd_remap is not None:
            self.set_option(_l.OPT_XCD_REMAP, 1 if xcd_remap else 0)
        if flat_blocks is not None:
            self.set_option(_l.OPT_FLAT_BLOCKS, flat_blocks)
        if prefetch is not None:
            self.set_option(_l.OPT_PREFETCH, prefetch)
        if strict_impl is not None:
            self.set_option(_l.OPT_STRICT_IMPL, strict_impl)
        if tile_wgs is not None:
            self.set_option(_l.OPT_TILE_WGS, tile_wgs)
        if strict_blocks is not None:
            self.set_option(_l.OPT_STRICT_BLOCKS, strict_blocks)
        if prefetch_read is not None:
            self.set_option(_l.OPT_PREFETCH_READ, prefetch_read)
        if overlap is not None:
            self.set_option(_l.OPT_OVERLAP, 1 if overlap else 0)
        if strict_spec is not None:
            self.set_option(_l.OPT_STRICT_SPEC, 1 if strict_spec else 0)
        if comm_rehearsal is not None:
            self.set_option(_l.OPT_COMM_REHEARSAL, int(comm_rehearsal))     # True / P: rehearse the traffic of P ranks
        return self

    def set_stencil(self, coef):
        """Five coefficients (up, left, centre, right, down: entries 1, 3, 4, 5, 7 of the reference's array) or all nine."""
        _l.check(self._lib.cg_set_stencil(self._ctx, _stencil9(coef)))
        return self

    def get_option(self, key):
        v = C.c_long(0)
        _l.check(self._lib.cg_get_option(self._ctx, key, C.byref(v)))
        return v.value

    @property
    def effective_schedule(self):
        """The schedule the next solve runs (after a solve: the one it ran)."""
        return self.get_option(_l.OPT_SCHEDULE)

    @property
    def bytes_per_point(self):
        """Algorithmic HBM bytes per grid point per iteration of the schedule in effect."""
        return _l.SCHED_BYTES_PER_POINT[self.effective_schedule]

    # -- multi-GPU -------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        _l.check(_l.load().cg_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, id128):
        buf = C.create_string_buffer(bytes(id128), 128)
        _l.check(self._lib.cg_comm_init(self._ctx, buf))

    def comm_init_self(self):
        """One-rank RCCL communicator standing in for every peer of a multi-rank context (timing rehearsal on one GPU)."""
        _l.check(self._lib.cg_comm_init_self(self._ctx))

    def comm_init_host(self, transport):
        """Carry halo rows and dot-product partials over `transport` (dist.RowTransport) instead of RCCL.

        Host-staged and synchronous: for running the multi-rank path with several
        processes on one GPU, or where RCCL is unavailable.
        """
        from .dist import make_callbacks
        self._callbacks = make_callbacks(transport)          # keep the ctypes thunks alive
        ex, ag = self._callbacks
        _l.check(self._lib.cg_comm_init_host(self._ctx, C.cast(ex, C.c_void_p), C.cast(ag, C.c_void_p), None))

    def comm_selftest(self):
        """Collective check of the transport (all-gather, row exchange, loopback); raises CGError on a mismatch.

        Overwrites scratch vectors: call it before init_problem.
        """
        _l.check(self._lib.cg_comm_selftest(self._ctx))

    # -- solve -----------------------------------------------------------
    def init_problem(self):
        _l.check(self._lib.cg_init_problem(self._ctx))
        return self

    def solve(self, max_iters=200, rtol=0.0):
        r = _l.Result()
        _l.check(self._lib.cg_solve(self._ctx, int(max_iters), float(rtol), C.byref(r)))
        return SolveResult(r.iters, r.q1, r.q_init, r.seconds, bool(r.converged))

    def sync_vectors(self):
        """Apply the u / d updates the last solve left owed (collective over the ranks)."""
        _l.check(self._lib.cg_sync_vectors(self._ctx))
        return self

    def verify(self, sum_start=0.0):
        """(sequential sum of this block's u continued from sum_start, max |u - exact|) on the device."""
        p = C.c_void_p()
        _l.check(self._lib.cg_device_ptr(self._ctx, _l.VEC_U, C.byref(p)))
        return verify_u(self.n, self.row0, self.rows, p, sum_start, self.device)

    def vector(self, which):
        out = np.empty(self.count, dtype=np.float64)
        _l.check(self._lib.cg_copy_out(self._ctx, which, out.ctypes.data_as(C.c_void_p), self.count))
        return out.reshape(self.rows, self.G)

    def vector_rows(self, which, row, nrows):
        """Rows [row, row+nrows) of one of the context's vectors (local row index), without copying the rest."""
        if row < 0 or nrows < 0 or row + nrows > self.rows:
            raise ValueError("row range outside the block")
        p = C.c_void_p()
        _l.check(self._lib.cg_device_ptr(self._ctx, which, C.byref(p)))
        out = np.empty(nrows * self.G, dtype=np.float64)
        _l.check(self._lib.cg_copy_to_host(self.device, out.ctypes.data_as(C.c_void_p),
                                           C.c_void_p(p.value + 8 * row * self.G), out.size))
        return out.reshape(nrows, self.G)

    def history(self, count):
        out = np.empty(max(int(count), 1), dtype=np.float64)
        n = C.c_long(0)
        _l.check(self._lib.cg_get_history(self._ctx, out.ctypes.data_as(C.c_void_p), int(count), C.byref(n)))
        return out[:n.value].copy()

    def time_kernel(self, which, reps=20):
        """(mean ms per launch, algorithmic bytes per launch) of one hot-path kernel, HIP-event timed."""
        ms, nbytes = C.c_double(0.0), C.c_double(0.0)
        _l.check(self._lib.cg_time_kernel(self._ctx, int(which), int(reps), C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.cg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

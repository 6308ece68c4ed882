"""Row-block decomposition and a host-side transport for the multi-rank solve.

One process per GPU; rank r of P owns the contiguous rows [r*G/P, (r+1)*G/P) of
the G x G grid and exchanges only whole rows with ranks r-1 and r+1.  This
replaces the reference's sqrt(P) x sqrt(P) cartesian blocks (utils.c:408-417) and
its row/column Isend/Irecv pairs (utils.c:231-272).

The production transport is RCCL inside the library (CGSolver.comm_init).
`RowTransport` carries the same two exchanges over torch.distributed with host
tensors (gloo works without a GPU); CGSolver.comm_init_host plugs it into the
library through the callbacks of cg_comm_init_host.
"""
import ctypes as C
import traceback

import numpy as np

from . import lib as _l


def partition(G, world, rank):
    """(row0, rows) of `rank`.  The grid side must divide evenly (cf. cg.c:23-26)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    if G % world:
        raise ValueError(f"Number of rows '{G}' not divisible by number of processors '{world}'.")
    rows = G // world
    return rank * rows, rows


class RowTransport:
    """Halo rows to/from the neighbours above and below, and an all-gather of one double per rank."""

    TAG_UP, TAG_DOWN = 11, 12        # direction the message travels

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.up = self.rank - 1 if self.rank > 0 else None
        self.down = self.rank + 1 if self.rank < self.world - 1 else None

    def exchange(self, first_row, last_row):
        """Send first_row to rank-1 and last_row to rank+1; return (row above, row below) or None at the boundary."""
        torch, dist = self._torch, self._dist
        reqs, top, bot = [], None, None
        if self.up is not None:
            top = torch.empty(len(first_row), dtype=torch.float64)
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(first_row)), self.up, self.group, self.TAG_UP))
            reqs.append(dist.irecv(top, self.up, self.group, self.TAG_DOWN))
        if self.down is not None:
            bot = torch.empty(len(last_row), dtype=torch.float64)
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(last_row)), self.down, self.group, self.TAG_DOWN))
            reqs.append(dist.irecv(bot, self.down, self.group, self.TAG_UP))
        for r in reqs:
            r.wait()
        return (None if top is None else top.numpy()), (None if bot is None else bot.numpy())

    def allgather(self, value):
        """[value of rank 0, ..., value of rank P-1], identical on every rank."""
        return [float(v) for v in self.allgather_block([float(value)])]

    def allgather_block(self, values):
        """Concatenation over the ranks, in rank order, of each rank's equally long array of doubles."""
        torch, dist = self._torch, self._dist
        mine = torch.tensor(np.ascontiguousarray(values, dtype=np.float64))
        out = [torch.empty(mine.numel(), dtype=torch.float64) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        return np.concatenate([t.numpy() for t in out])


def _view(ptr, count):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,))


def make_callbacks(transport):
    """ctypes thunks (exchange, allgather) for cg_comm_init_host; keep the returned objects alive."""

    def exchange(user, send_up, send_down, recv_top, recv_bot, count):
        try:
            first = _view(send_up, count) if send_up else None
            last = _view(send_down, count) if send_down else None
            top, bot = transport.exchange(first, last)
            if recv_top:
                _view(recv_top, count)[:] = top
            if recv_bot:
                _view(recv_bot, count)[:] = bot
            return 0
        except Exception:
            traceback.print_exc()
            return 1

    def allgather(user, mine, count, out):
        try:
            vals = transport.allgather_block(_view(mine, count).copy())
            _view(out, len(vals))[:] = vals
            return 0
        except Exception:
            traceback.print_exc()
            return 1

    return _l.EXCHANGE_FN(exchange), _l.ALLGATHER_FN(allgather)

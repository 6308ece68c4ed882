"""The entry point that allocates nothing (cg_stencil_noalloc) against cg_stencil, through the C ABI.

Two calls in a row on the same GPU -- they share the library's static scratch array, which keeps the
first call's partial sums -- the second with a smaller input.  Bar: bit-exact, q and d.q alike.
Then what the entry point is for: the call recorded between hipStreamBeginCapture and hipStreamEndCapture on a
caller's stream and replayed as a graph, and the device's free memory unchanged across a call.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _both(cg, rng, G, row0, rows):
    full = rng.standard_normal((G, G))
    blk = np.ascontiguousarray(full[row0:row0 + rows])
    d = cg.DeviceVector.from_host(blk)
    t = cg.DeviceVector.from_host(np.ascontiguousarray(full[row0 - 1])) if row0 > 0 else None
    b = cg.DeviceVector.from_host(np.ascontiguousarray(full[row0 + rows])) if row0 + rows < G else None
    q0 = rng.standard_normal(rows * G)                   # boundary entries of q must survive in both
    q_ref, q_new = cg.DeviceVector.from_host(q0), cg.DeviceVector.from_host(q0)
    dq_ref = cg.stencil(G, row0, rows, d, q_ref, top=t, bot=b, want_dq=True)
    dq_dev = cg.DeviceVector.from_host(np.array([np.nan]))
    cg.stencil_noalloc(G, row0, rows, d, q_new, top=t, bot=b, dq=dq_dev)      # null stream; the copies below wait for it
    got_q, got_dq = q_new.to_host(), dq_dev.to_host()[0]
    assert np.array_equal(bits(got_q), bits(q_ref.to_host()))
    assert float(got_dq).hex() == float(dq_ref).hex()


def test_noalloc_entry_point_matches_cg_stencil_twice_in_a_row(cg):
    rng = np.random.default_rng(77)
    _both(cg, rng, 2048, 0, 2048)          # 4 strips x several row tiles: many partial sums
    _both(cg, rng, 257, 100, 57)           # smaller, odd row length (scalar path), halo rows on both sides
    _both(cg, rng, 64, 0, 64)              # smaller still: stale partials of the earlier calls lie beyond its tiles


def test_noalloc_entry_point_rejects_what_does_not_fit(cg):
    lib = cg.lib.load()
    d = cg.DeviceVector(16)
    # one row per tile cannot be forced from outside; a block with more tiles than the array holds can: 2^17 strips
    rc = lib.cg_stencil_noalloc(0, 1 << 26, 0, 1, d.ptr, None, d.ptr, d.ptr, None, None)
    assert rc != 0 and b"static scratch" in lib.cg_last_error()


def _hip():
    hip = C.CDLL("libamdhip64.so")
    for f in ("hipStreamCreateWithFlags", "hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch",
              "hipStreamSynchronize", "hipMemGetInfo", "hipGraphExecDestroy", "hipGraphDestroy", "hipStreamDestroy", "hipDeviceSynchronize"):
        getattr(hip, f).restype = C.c_int
    return hip


def _free_bytes(hip):
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_entry_point_recorded_in_a_stream_capture_and_replayed(cg):
    rng = np.random.default_rng(78)
    hip = _hip()
    G, row0, rows = 1024, 256, 512
    full = rng.standard_normal((G, G))
    d = cg.DeviceVector.from_host(np.ascontiguousarray(full[row0:row0 + rows]))
    t = cg.DeviceVector.from_host(np.ascontiguousarray(full[row0 - 1]))
    b = cg.DeviceVector.from_host(np.ascontiguousarray(full[row0 + rows]))
    q0 = rng.standard_normal(rows * G)
    q_ref, q_new = cg.DeviceVector.from_host(q0), cg.DeviceVector.from_host(q0)
    dq_ref = cg.stencil(G, row0, rows, d, q_ref, top=t, bot=b, want_dq=True)      # (also the device's first library call: outside capture)
    want = bits(q_ref.to_host())
    dq_dev = cg.DeviceVector.from_host(np.array([np.nan]))
    stream = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0                  # hipStreamNonBlocking
    # eager on the caller's stream: not a byte of device memory taken or given back
    assert hip.hipDeviceSynchronize() == 0
    before = _free_bytes(hip)
    cg.stencil_noalloc(G, row0, rows, d, q_new, top=t, bot=b, dq=dq_dev, stream=stream)
    assert _free_bytes(hip) == before
    assert hip.hipStreamSynchronize(stream) == 0
    assert np.array_equal(bits(q_new.to_host()), want) and float(dq_dev.to_host()[0]).hex() == float(dq_ref).hex()
    # recorded, not run
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 2) == 0                              # hipStreamCaptureModeRelaxed
    cg.stencil_noalloc(G, row0, rows, d, q_new, top=t, bot=b, dq=dq_dev, stream=stream)
    assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0 and graph.value
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    for rep in range(2):                                                          # the same graph twice, from spoiled outputs
        q_new.copy_from_host(q0)
        dq_dev.copy_from_host(np.array([np.nan]))
        assert hip.hipGraphLaunch(gexec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        assert np.array_equal(bits(q_new.to_host()), want), rep
        assert float(dq_dev.to_host()[0]).hex() == float(dq_ref).hex(), rep
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph); hip.hipStreamDestroy(stream)

"""The CPU oracle against the reference's own outputs (CPU only).

Every golden record was produced by running the reference's cg.c + utils.c
(oracle/_ref/cg_ref, see tests/golden/make_golden.py).  The oracle must
reproduce the single-rank records bit for bit: that is what pins it.
"""
import numpy as np
import pytest

from conftest import bits, golden_arrays, seq_sum

NP1_FIXED = [(15, 10), (20, 12), (63, 50), (100, 30), (256, 200), (255, 200), (511, 100), (1023, 200)]


@pytest.mark.parametrize("n,iters", NP1_FIXED)
def test_oracle_matches_reference_q1_and_sum_u(oracle, golden, n, iters):
    rec = golden[(n, iters, 1, 0.0)]
    r = oracle.cg(n, iters)
    assert r["iters"] == rec["iters"] == iters
    assert float(r["q1"]).hex() == float.fromhex(rec["q1_hex"]).hex()
    assert float(r["q_init"]).hex() == float.fromhex(rec["q_init_hex"]).hex()
    assert float(seq_sum(oracle, r["u"])).hex() == rec["sum_u_hex"]


@pytest.mark.parametrize("n,iters", [(15, 10), (20, 12), (63, 50)])
def test_oracle_matches_reference_arrays_bitwise(oracle, golden, n, iters):
    rec = golden[(n, iters, 1, 0.0)]
    want = golden_arrays(rec["arrays"], n)
    r = oracle.cg(n, iters)
    for k in ("u", "g", "d"):
        assert np.array_equal(bits(r[k]), bits(want[k])), k


@pytest.mark.parametrize("n", [15, 63, 255])
def test_oracle_stop_rule_matches_reference(oracle, golden, n):
    rec = golden[(n, 10000, 1, 1e-8)]
    r = oracle.cg(n, 10000, rtol=1e-8)
    assert r["iters"] == rec["iters"]
    assert float(r["q1"]).hex() == float.fromhex(rec["q1_hex"]).hex()
    assert float(seq_sum(oracle, r["u"])).hex() == rec["sum_u_hex"]


@pytest.mark.parametrize("n,blocks", [(15, 2), (15, 4), (63, 8), (255, 4), (20, 3), (20, 7)])
def test_rowblock_form_is_bit_identical(oracle, n, blocks):
    a = oracle.cg(n, 25)
    b = oracle.cg(n, 25, nblocks=blocks)
    assert a["iters"] == b["iters"]
    for k in ("u", "g", "d", "q", "hist"):
        assert np.array_equal(bits(a[k]), bits(b[k])), k


def test_rowblock_rejects_indivisible(oracle):
    with pytest.raises(ValueError):
        oracle.cg(15, 5, nblocks=3)


def test_first_stencil_closed_form(oracle):
    # A d0 = 8 h^4 wherever all four neighbours are interior points.
    n = 64
    d = oracle.init_d(n)
    q = oracle.stencil(d)
    h = 1.0 / n
    inner = q[2:-2, 2:-2]
    assert np.allclose(inner, 8 * h ** 4, rtol=1e-9, atol=0)
    assert np.all(q[0] == 0) and np.all(q[-1] == 0) and np.all(q[:, 0] == 0) and np.all(q[:, -1] == 0)


def test_init_values(oracle):
    n = 20
    d = oracle.init_d(n)
    g = oracle.init_g(d)
    h = 1.0 / n
    i, j = 3, 7
    assert d[i, j] == 2 * h * h * ((i * h) * (1 - i * h) + (j * h) * (1 - j * h))
    assert np.all(d[0] == 0) and np.all(d[:, -1] == 0)
    assert np.array_equal(bits(g), bits(-d))
    # row-block initialisation is the same function of the global row index
    blk = oracle.init_d(n, row0=7, rows=7)
    assert np.array_equal(bits(blk), bits(d[7:14]))


def test_converged_solution_is_the_analytic_one(oracle):
    # u = x(1-x) y(1-y) has zero fourth derivatives, so the 5-point stencil is exact for it.
    n = 63
    r = oracle.cg(n, 10000, rtol=1e-8)
    x = np.arange(n + 1) / n
    exact = np.outer(x * (1 - x), x * (1 - x))
    assert np.max(np.abs(r["u"] - exact)) < 5e-11


def test_history_and_stop_semantics(oracle):
    r = oracle.cg(63, 10000, rtol=1e-8)
    thr = (1e-8) ** 2 * r["q_init"]
    assert r["hist"][-1] <= thr
    assert np.all(r["hist"][:-1] > thr)
    assert len(r["hist"]) == r["iters"]


@pytest.mark.parametrize("n,chunk", [(63, 16), (255, 64), (1023, 128)])
def test_chunked_first_iteration_equals_whole_grid_oracle(oracle, n, chunk):
    # the row-chunked evaluation used to check grids that do not fit host memory
    from chunked_oracle import first_iteration
    ref = oracle.cg(n, 1)
    G = n + 1
    sample = [0, 1, chunk - 1, chunk, G // 2, G - 2, G - 1]
    got = first_iteration(oracle, n, chunk, sample, workers=4)
    assert float(got["q0"]).hex() == float(ref["q_init"]).hex()
    assert float(got["q1"]).hex() == float(ref["q1"]).hex()
    for i in sample:
        u1, g1, d1 = got["rows"][i]
        assert np.array_equal(u1.view(np.uint64), ref["u"][i].view(np.uint64))
        assert np.array_equal(g1.view(np.uint64), ref["g"][i].view(np.uint64))
        assert np.array_equal(d1.view(np.uint64), ref["d"][i].view(np.uint64))


@pytest.mark.parametrize("n,iters,np_", [(15, 10, 4), (63, 50, 4), (63, 50, 16), (255, 200, 4), (255, 200, 16), (1023, 200, 4)])
def test_oracle_in_the_references_multirank_summation_layout_matches_its_np_outputs(oracle, golden, n, iters, np_):
    # the reference run on 4 and 16 ranks (tests/golden/make_golden.py): per-block sequential sums, then
    # MPI_Allreduce; the restatement of that layout reproduces every bit of q1 and q_init
    rec = golden[(n, iters, np_, 0.0)]
    res = oracle.cg(n, iters, blocks2d=int(round(np_ ** 0.5)))
    assert res["iters"] == iters
    assert float(res["q1"]).hex() == float.fromhex(rec["q1_hex"]).hex()
    assert float(res["q_init"]).hex() == float.fromhex(rec["q_init_hex"]).hex()


STENCIL_KEYS = [(15, 10, 1, (-1, -2, 6, -2, -1)), (63, 50, 1, (-1, -2, 6, -2, -1)), (63, 50, 4, (-1, -2, 6, -2, -1)),
                (255, 100, 1, (-3, -1, 9, -1, -3)), (63, 20, 1, (-1, -2, 7, -3, -1))]


@pytest.mark.parametrize("n,iters,np_,coef", STENCIL_KEYS)
def test_oracle_with_other_stencil_coefficients_matches_reference(oracle, golden, n, iters, np_, coef):
    # the reference reads its five coefficients from the struct (utils.c:27-31); goldens made with REF_STENCIL
    rec = golden[(n, iters, np_, 0.0, coef)]
    oracle.set_stencil(coef)
    try:
        res = oracle.cg(n, iters, blocks2d=None if np_ == 1 else int(round(np_ ** 0.5)))
    finally:
        oracle.set_stencil()
    assert float(res["q1"]).hex() == float.fromhex(rec["q1_hex"]).hex()
    if "arrays" in rec:
        want = golden_arrays(rec["arrays"], n)
        for k in ("u", "g", "d"):
            assert np.array_equal(bits(res[k]), bits(want[k])), k
    if "sum_u_hex" in rec:
        assert float(seq_sum(oracle, res["u"])).hex() == rec["sum_u_hex"]


def test_nine_point_restatement_self_consistency(oracle):
    # The corner terms have no counterpart in the reference (utils.c:27-31 reads five entries): parity for them is
    # unpinned.  What can be pinned on the CPU: zero corners give the five-term bits, and the nine-term sum is the
    # loop of utils.c:25-31 continued in array order.
    rng = np.random.default_rng(3)
    d = rng.standard_normal((37, 37))
    five = oracle.stencil(d)
    oracle.set_stencil((0, -1, 0, -1, 4, -1, 0, -1, 0))
    try:
        nine0 = oracle.stencil(d)
    finally:
        oracle.set_stencil()
    assert np.array_equal(five.view(np.int64), nine0.view(np.int64))
    c = (1, -2, 3, -4, 20, -5, 2, -3, 1)
    oracle.set_stencil(c)
    try:
        got = oracle.stencil(d)
    finally:
        oracle.set_stencil()
    want = np.zeros_like(d)
    for i in range(1, 36):
        for j in range(1, 36):
            acc = 0.0
            k = 0
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    acc += float(c[k]) * d[i + di, j + dj]
                    k += 1
            want[i, j] = acc
    assert np.array_equal(got.view(np.int64), want.view(np.int64))

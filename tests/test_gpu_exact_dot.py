"""The parallel evaluation of the reference's sequential dot product must return the SAME BITS as the
CPU loop (oracle.dot, utils.c:372-379) for any input -- not just well-behaved ones.  The cases below aim
at every branch: binade crossings in both directions, ties (an addend of exactly half an ulp), zeros,
cancellation to zero and to negative values, huge dynamic range, non-zero start, odd lengths."""
import numpy as np
import pytest

from conftest import bits, seq_sum

pytestmark = pytest.mark.gpu

REPLAY, PARALLEL = 2, 3


def check(cg, oracle, v, w, start=0.0):
    v = np.ascontiguousarray(v, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    want = oracle.dot(v, w, start=start)
    dv, dw = cg.DeviceVector.from_host(v), cg.DeviceVector.from_host(w)
    got_p = cg.dot(v.size, dv, dw, strict=PARALLEL, start=start)
    got_r = cg.dot(v.size, dv, dw, strict=REPLAY, start=start)
    assert float(got_r).hex() == float(want).hex(), "literal replay"
    assert float(got_p).hex() == float(want).hex(), "parallel evaluation"


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(7)


@pytest.mark.parametrize("count", [1, 2, 63, 64, 65, 1023, 1024, 1025, 65535, 65536, 65537, 1 << 20, (1 << 22) + 3])
def test_random_mixed_sign(cg, oracle, rng, count):
    check(cg, oracle, rng.standard_normal(count), rng.standard_normal(count))


@pytest.mark.parametrize("count", [1000, 70001, (1 << 21) + 17])
def test_squares_and_start_values(cg, oracle, rng, count):
    v = rng.standard_normal(count)
    for start in (0.0, 1.0, -3.5, 1e6, -1e-9):
        check(cg, oracle, v, v, start=start)


def test_wide_dynamic_range(cg, oracle, rng):
    n = 300001
    v = 10.0 ** rng.uniform(-12, 12, n) * rng.choice([-1.0, 1.0], n)
    check(cg, oracle, v, np.ones(n))
    check(cg, oracle, np.sort(np.abs(v)), np.ones(n))            # increasing: crosses every binade upwards
    check(cg, oracle, np.sort(np.abs(v))[::-1].copy(), np.ones(n))


def test_ties_half_ulp_addends(cg, oracle):
    # accumulator 1.0 (ulp 2^-52); every further addend is exactly half an ulp: each add is a tie
    n = 5000
    v = np.full(n, 2.0 ** -53)
    v[0] = 1.0
    check(cg, oracle, v, np.ones(n))
    # ties mixed with ordinary addends, both parities of the accumulator
    v = np.tile(np.array([2.0 ** -53, 2.0 ** -52, 3 * 2.0 ** -53, 2.0 ** -54]), 4000)
    v[0] = 1.0
    check(cg, oracle, v, np.ones(v.size))
    check(cg, oracle, -v, np.ones(v.size))


def test_small_integers_cross_binades_often(cg, oracle, rng):
    n = 200003
    check(cg, oracle, rng.integers(-3, 4, n).astype(float), np.ones(n))
    check(cg, oracle, rng.integers(0, 3, n).astype(float), np.ones(n))


def test_zeros(cg, oracle, rng):
    n = 100000
    z = np.zeros(n)
    check(cg, oracle, z, z)
    v = rng.standard_normal(n)
    v[: 40000] = 0.0                                               # leading zeros (the grid's boundary row)
    check(cg, oracle, v, v)
    v[::2] = 0.0
    check(cg, oracle, v, rng.standard_normal(n))
    v[:] = 0.0
    v[77777] = 3.25
    check(cg, oracle, v, np.ones(n))


def test_cancellation_to_zero_and_negative(cg, oracle, rng):
    n = 90000
    a = rng.standard_normal(n // 2)
    v = np.empty(n)
    v[0::2], v[1::2] = a, -a                                       # accumulator returns to exactly 0 every second add
    check(cg, oracle, v, np.ones(n))
    v = np.concatenate([np.full(30000, 1e-3), np.full(60000, -1e-3)])   # rises, falls through zero, goes negative
    check(cg, oracle, v, np.ones(n))
    v = np.concatenate([np.full(50000, 1.0), np.full(50000, -1.0 + 2.0 ** -30)])
    check(cg, oracle, v, np.ones(v.size))


def test_huge_then_tiny_and_tiny_then_huge(cg, oracle, rng):
    n = 70000
    v = rng.uniform(1e-20, 2e-20, n)
    v[0] = 1e20
    check(cg, oracle, v, np.ones(n))
    v[0], v[-1] = v[1], 1e20
    check(cg, oracle, v, np.ones(n))
    v[n // 2] = -1e20
    check(cg, oracle, v, np.ones(n))


def test_subnormal_accumulator(cg, oracle, rng):
    n = 40000
    v = rng.uniform(1, 2, n) * 1e-320
    check(cg, oracle, v, np.ones(n))


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_one_crossing_in_a_chosen_piece_of_a_chunk(cg, oracle, rng, sign):
    # the walk's prepared path per marked chunk: a run of sixteen-element pieces, ONE piece added literally, a run
    # behind it -- with the crossing in the first piece (no run before), the last (no run behind), in between, at
    # the chunk's very start and end; accumulator rising through a power of two and falling through it; ragged tail
    n = 5 * 1024 + 77
    v = rng.uniform(0.5, 1.5, n)
    for m in (0, 1, 15, 16, 17, 500, 1007, 1008, 1023, 1024, 1025, 2048 + 16 * 63 - 1, 4 * 1024 + 33, n - 3):
        up = float(np.sum(v[:m]))
        check(cg, oracle, sign * v, np.ones(n), start=sign * (2.0 ** 20 - up))          # |s| grows through 2^20 near element m
        check(cg, oracle, -sign * v, np.ones(n), start=sign * (2.0 ** 20 + up))         # |s| falls through 2^20 near element m
    w = v.copy()
    w[1500:1600] *= -1.0                                                                # back and forth across the boundary inside one chunk
    check(cg, oracle, sign * w, np.ones(n), start=sign * (2.0 ** 20 - float(np.sum(v[:1490]))))


def test_smooth_fields_like_the_solver(cg, oracle):
    # the actual operands: d0 and q = A d0 on a 1024^2 grid, and g.g
    n = 1023
    d = oracle.init_d(n)
    q = oracle.stencil(d)
    g = oracle.init_g(d)
    check(cg, oracle, d.ravel(), q.ravel())
    check(cg, oracle, g.ravel(), g.ravel())


# ----------------------------------------------------------- whole solves ------

@pytest.mark.parametrize("n,iters", [(15, 10), (63, 50)])
def test_strict_solve_parallel_evaluation_small_arrays_bitwise(cg, golden, n, iters):
    from conftest import golden_arrays
    rec = golden[(n, iters, 1, 0.0)]
    want = golden_arrays(rec["arrays"], n)
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_impl=cg.STRICT_PARALLEL).init_problem()
    r = s.solve(iters)
    assert float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex()
    for k, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):
        assert np.array_equal(bits(s.vector(which)), bits(want[k])), k


@pytest.mark.parametrize("n,iters", [(255, 200), (1023, 200), (1023, 1000), (2047, 20), (4095, 200)])
def test_strict_solve_matches_reference_at_speed(cg, oracle, golden, n, iters):
    rec = golden.get((n, iters, 1, 0.0))
    if rec is None:
        pytest.skip("golden not generated")
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_impl=cg.STRICT_PARALLEL).init_problem()
    r = s.solve(iters)
    assert r.iters == iters
    assert float(r.q_init).hex() == float.fromhex(rec["q_init_hex"]).hex()
    assert float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex()
    assert float(seq_sum(oracle, s.vector(cg.VEC_U))).hex() == rec["sum_u_hex"]


@pytest.mark.parametrize("n,iters,coef", [(1023, 120, None), (2047, 40, None), (4095, 24, None),
                                          (2047, 40, (-1, -2, -1, -2, 12, -2, -1, -2, -1)),      # all nine entries
                                          (1023, 80, (0, -3, 0, -1, 8, -1, 0, -3, 0))])
def test_strict_summaries_pass_the_self_check(cg, n, iters, coef):
    # Every chunk summary the passes hand over (sums under the hypothesised binade, bounds of the running value)
    # is compared with one recomputed from the products, after every dot product of the solve, and the
    # plan-driven walk with the plain hierarchical one; the history must equal the literal replay's.
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT)
    if coef is not None:
        s.set_stencil(coef)
    s.init_problem()
    s.set_option(cg.lib.OPT_STRICT_AUDIT, 1)
    s.solve(iters)
    assert s.get_option(cg.lib.OPT_STRICT_AUDIT) == 0
    got = s.history(iters)
    s.close()
    lit = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_impl=cg.STRICT_REPLAY)
    if coef is not None:
        lit.set_stencil(coef)
    lit.init_problem()
    lit.solve(iters)
    want = lit.history(iters)
    lit.close()
    assert np.array_equal(bits(got), bits(want))


def test_headline_16384_1000_strict_with_the_self_check_on(cg, golden):
    # The headline run once more with every summary of every dot product audited (2000 dot products of 2.7e8 terms).
    rec = golden.get((16383, 1000, 1, 0.0))
    if rec is None:
        pytest.skip("16383/1000 golden not generated")
    s = cg.CGSolver(16383).configure(reduce=cg.REDUCE_STRICT).init_problem()
    s.set_option(cg.lib.OPT_STRICT_AUDIT, 1)
    r = s.solve(1000)
    assert s.get_option(cg.lib.OPT_STRICT_AUDIT) == 0
    assert float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex()


def test_headline_16384_1000_strict_is_bit_identical_to_the_reference(cg, oracle, golden):
    # The headline configuration in the reference's own summation order: 2.7e8-term dot products,
    # 1000 iterations, and every bit of q1, q_init and sum(u) equals what the reference printed
    # after 2314 s on a CPU core.
    rec = golden.get((16383, 1000, 1, 0.0))
    if rec is None:
        pytest.skip("16383/1000 golden not generated")
    s = cg.CGSolver(16383).configure(reduce=cg.REDUCE_STRICT).init_problem()
    r = s.solve(1000)
    assert r.iters == 1000
    assert float(r.q_init).hex() == float.fromhex(rec["q_init_hex"]).hex()
    assert float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex()
    assert float(seq_sum(oracle, s.vector(cg.VEC_U))).hex() == rec["sum_u_hex"]
    print(f"strict 16384^2 x 1000 iterations: {r.seconds:.2f} s = {1000 / r.seconds:.1f} it/s")


# ------------------------------------- the reference's own multi-rank results ------

@pytest.mark.parametrize("impl", ["replay", "parallel"])
@pytest.mark.parametrize("n,iters,np_", [(15, 10, 4), (63, 50, 4), (63, 50, 16), (255, 200, 4), (255, 200, 16), (1023, 200, 4)])
def test_reference_multirank_summation_layout_reproduces_its_np_outputs(cg, golden, n, iters, np_, impl):
    # `mpirun -np 4|16 ./cg n`: each rank sums its 2-D block sequentially, MPI_Allreduce combines the sums.
    # One GPU evaluating the dot products in that layout gives the reference's multi-rank bits.
    rec = golden[(n, iters, np_, 0.0)]
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_blocks=int(round(np_ ** 0.5)),
                                 strict_impl=cg.STRICT_REPLAY if impl == "replay" else cg.STRICT_PARALLEL).init_problem()
    r = s.solve(iters)
    assert r.iters == iters
    assert float(r.q_init).hex() == float.fromhex(rec["q_init_hex"]).hex()
    assert float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex()


def test_config_4096_to_1e8_is_bit_identical_to_the_reference_on_4_ranks(cg, golden):
    # The 4096^2 configuration, run to a relative residual of 1e-8.  The reference was run on 4 ranks
    # (a single rank would take hours): 6937 iterations.  Same summation layout here: same stopping
    # iteration, same last q1, bit for bit.
    rec = golden.get((4095, 100000, 4, 1e-8))
    if rec is None:
        pytest.skip("golden not generated")
    s = cg.CGSolver(4095).configure(reduce=cg.REDUCE_STRICT, strict_blocks=2).init_problem()
    r = s.solve(100000, rtol=1e-8)
    assert r.converged and r.iters == rec["iters"] == 6937
    assert float(r.q_init).hex() == float.fromhex(rec["q_init_hex"]).hex()
    assert float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex()
    print(f"4096^2 to 1e-8 in the reference's np=4 order: {r.iters} iterations in {r.seconds:.2f} s")


# ------------------------- summaries formed inside the producing passes (speculated exponents) ------

@pytest.mark.parametrize("n,iters", [(1023, 61), (2047, 40)])
def test_speculative_summaries_give_the_bits_of_the_stored_products_path(cg, oracle, n, iters):
    # lean schedule, strict order: the passes summarise their products under two hypothesised exponents per span
    # (CG_OPT_STRICT_SPEC, default) instead of storing d'*q and reading it and g' again.  Every bit must agree
    # with the path that stores the products, for every iteration, and with the CPU restatement.
    a = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_spec=True).init_problem()
    b = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_spec=False).init_problem()
    ra, rb = a.solve(iters), b.solve(iters)
    assert np.array_equal(bits(a.history(iters)), bits(b.history(iters)))
    for which in (cg.VEC_U, cg.VEC_G, cg.VEC_D):
        assert np.array_equal(bits(a.vector(which)), bits(b.vector(which))), which
    if n <= 1023:
        assert np.array_equal(bits(a.history(iters)), bits(oracle.cg(n, iters)["hist"]))
    misses = a.get_option(cg.lib.OPT_STRICT_MISSES)
    print(f"n={n}: {misses} of {(n + 1) * (n + 1) // 1024} chunks redone in the last dot product")


def test_leading_iteration_hands_its_prefix_sums_to_the_first_speculating_iteration(cg, oracle):
    # The iteration that starts a solve evaluates its dot products from stored products; the approximate prefix sums it
    # forms on the way become the hypotheses of the iteration behind it, which would otherwise redo every chunk.
    n = 2047
    chunks = (n + 1) * (n + 1) // 1024
    for iters in (2, 3):
        s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_spec=True, graph_iters=0).init_problem()
        s.solve(iters)
        misses = s.get_option(cg.lib.OPT_STRICT_MISSES)
        print(f"n={n}: {misses} of {chunks} chunks redone in the last dot product of iteration {iters}")
        assert misses < chunks // 4
        b = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_spec=False).init_problem()
        b.solve(iters)
        assert np.array_equal(bits(s.history(iters)), bits(b.history(iters)))
        for which in (cg.VEC_U, cg.VEC_G, cg.VEC_D):
            assert np.array_equal(bits(s.vector(which)), bits(b.vector(which))), which
        s.close(); b.close()


def test_solves_that_restart_from_materialised_vectors_keep_the_bits(cg, oracle):
    # Reading a vector between two solves has the owed d and u updates applied, so the next solve starts from a
    # materialised direction: with the last solve's hypotheses at hand its leading iteration runs in the lean form
    # (pass 1 on d alone, the in-place update pass with the span sums inside), with its u update applied at once or
    # left to the first pair, whichever the count of iterations behind it asks for.
    n = 1023
    counts = (3, 1, 4, 2, 7, 6, 9)
    ref = oracle.cg(n, sum(counts))
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, graph_iters=4).init_problem()
    hist, done = [], 0
    for k in counts:
        s.solve(k)
        hist.append(s.history(k))
        done += k
        mid = oracle.cg(n, done)
        for key, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):     # (materialises)
            assert np.array_equal(bits(s.vector(which)), bits(mid[key])), (key, done)
    assert np.array_equal(bits(np.concatenate(hist)), bits(ref["hist"]))
    s.close()


def test_placement_search_and_pairing_choices_do_not_touch_the_bits(cg, oracle, monkeypatch):
    # A context for a large grid draws sets of vectors, trades single vectors between them, times the lean passes per
    # direction, and cg_solve uses those times to decide which of the two kinds of pass 2 reads which buffer pair (also:
    # whether a continuing solve should start over for the other assignment).  Here the search runs on a small grid, and
    # both pairings are forced in turn: every bit must be what the restatement gives.
    n = 1023
    monkeypatch.setenv("CG_PLACEMENT_MIN_POINTS", "1")
    monkeypatch.setenv("CG_PLACEMENT_DRAWS", "3")
    ref = oracle.cg(n, 3 + 40 + 41)
    for forced in ("0", "1", None):
        s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, graph_iters=4).init_problem()
        if forced is None:
            monkeypatch.delenv("CG_LEAD_DEFER_U", raising=False)
        else:
            monkeypatch.setenv("CG_LEAD_DEFER_U", forced)
        hist = []
        s.solve(3); hist.append(s.history(3))
        monkeypatch.delenv("CG_LEAD_DEFER_U", raising=False)
        s.solve(40); hist.append(s.history(40))           # a pure continuation: may start over for the other assignment
        s.solve(41); hist.append(s.history(41))
        assert np.array_equal(bits(np.concatenate(hist)), bits(ref["hist"])), forced
        for key, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):
            assert np.array_equal(bits(s.vector(which)), bits(ref[key])), (forced, key)
        s.close()


def test_speculative_summaries_continue_across_solves_and_stop_rule(cg, oracle):
    n = 1023
    ref = oracle.cg(n, 120)
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, graph_iters=10).init_problem()
    hist = []
    for k in (1, 2, 30, 33, 54):
        s.solve(k)
        hist.append(s.history(k))
    assert np.array_equal(bits(np.concatenate(hist)), bits(ref["hist"]))
    for k, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):
        assert np.array_equal(bits(s.vector(which)), bits(ref[k])), k
    # a new problem on the same context: last solve's hypotheses are forgotten, the bits are the same again
    s.init_problem()
    s.solve(120)
    assert np.array_equal(bits(s.history(120)), bits(ref["hist"]))


@pytest.mark.parametrize("coef", [(-1, -2, 6, -2, -1), (-1, -1, -1, -1, 8, -1, -1, -1, -1)])
def test_speculative_summaries_with_other_coefficients(cg, oracle, coef):
    # the repair of missed chunks and the walk's descents recompute d'*(A d') from d and g: every coefficient set
    n, iters = 1023, 40
    oracle.set_stencil(coef)
    try:
        ref = oracle.cg(n, iters)
    finally:
        oracle.set_stencil()
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT).set_stencil(coef).init_problem()
    s.solve(iters)
    assert np.array_equal(bits(s.history(iters)), bits(ref["hist"]))
    for k, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):
        assert np.array_equal(bits(s.vector(which)), bits(ref[k])), k

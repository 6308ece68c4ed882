"""Multi-rank row-block solve on real hardware: several processes share GPU 0, the halo rows and
partial sums travel host-staged over gloo (cg_comm_init_host), everything else -- the halo-aware
kernels, the locally maintained halo rows of d in the fused schedules, the interior/edge split of
the classic schedule, the rank-ordered reduction, the stop flag -- is the production code path."""
import numpy as np
import pytest

from dist_launch import run_ranks, stack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule", ["classic", "fused", "fused_noq", "lean"])
@pytest.mark.parametrize("world,n,iters", [(2, 255, 60), (4, 255, 60), (4, 1023, 120), (3, 20, 12), (7, 20, 12)])
def test_multirank_matches_single_rank(oracle, tmp_path, schedule, world, n, iters):
    res = run_ranks(world, ["gpu", n, iters, tmp_path, schedule], tmp_path)
    ref = oracle.cg(n, iters)
    assert all(int(r["iters"]) == iters for r in res)
    q1 = float(res[0]["q1"])
    assert all(float(r["q1"]).hex() == float(q1).hex() for r in res)          # identical scalars on every rank
    assert abs(q1 - ref["q1"]) <= 5e-8 * ref["q1"]
    assert abs(float(res[0]["q_init"]) - ref["q_init"]) <= 1e-12 * ref["q_init"]
    u, g, d = stack(res, "u"), stack(res, "g"), stack(res, "d")
    assert np.max(np.abs(u - ref["u"])) <= 1e-8 * np.max(np.abs(ref["u"]))
    assert np.max(np.abs(g - ref["g"])) <= 1e-5 * np.max(np.abs(ref["g"]))
    assert np.max(np.abs(d - ref["d"])) <= 1e-5 * np.max(np.abs(ref["d"]))
    assert np.allclose(res[0]["hist"], ref["hist"], rtol=5e-8, atol=0)


@pytest.mark.parametrize("schedule", ["classic", "lean"])
@pytest.mark.parametrize("world,n,iters", [(4, 7, 6), (8, 7, 6), (8, 15, 10)])
def test_multirank_blocks_of_one_or_two_rows(oracle, tmp_path, schedule, world, n, iters):
    # every row of a block is an edge row: halo rows on both sides, no interior tiles to overlap with
    res = run_ranks(world, ["gpu", n, iters, tmp_path, schedule], tmp_path)
    ref = oracle.cg(n, iters)
    assert all(float(r["q1"]).hex() == float(res[0]["q1"]).hex() for r in res)
    live = ref["hist"] > 1e-25
    assert np.allclose(np.asarray(res[0]["hist"])[live], ref["hist"][live], rtol=1e-10, atol=0)
    assert np.max(np.abs(stack(res, "u") - ref["u"])) <= 1e-10 * np.max(np.abs(ref["u"]))


def test_multirank_stop_rule(oracle, tmp_path):
    n = 63
    res = run_ranks(4, ["gpu", n, 10000, tmp_path, "lean", 1e-8], tmp_path)
    ref = oracle.cg(n, 10000, rtol=1e-8)
    assert all(int(r["converged"]) == 1 for r in res)
    assert len({int(r["iters"]) for r in res}) == 1
    assert abs(int(res[0]["iters"]) - ref["iters"]) <= 2
    x = np.arange(n + 1) / n
    assert np.max(np.abs(stack(res, "u") - np.outer(x * (1 - x), x * (1 - x)))) < 5e-11


@pytest.mark.parametrize("order", ["strict_replay", "strict_parallel"])
@pytest.mark.parametrize("schedule,world,n,iters", [("classic", 2, 63, 50), ("fused", 4, 63, 50), ("fused", 4, 255, 200),
                                                    ("classic", 4, 1023, 40), ("fused", 2, 1023, 200),
                                                    ("lean", 4, 255, 200), ("lean", 2, 1023, 61), ("fused_noq", 4, 63, 50),
                                                    ("lean", 4, 1023, 120), ("lean", 8, 2047, 24)])
def test_multirank_strict_order_reproduces_single_rank_reference_bits(golden, oracle, tmp_path, order, schedule, world, n, iters):
    # The accumulator of every dot product is handed from rank to rank, so 2 or 4 row blocks give the
    # bits the reference gives on ONE rank (its own np=4 run differs from its np=1 run by ~1e-9).
    res = run_ranks(world, ["gpu", n, iters, tmp_path, schedule, 0.0, order], tmp_path)
    rec = golden.get((n, iters, 1, 0.0))
    ref = oracle.cg(n, iters)
    if rec is not None:
        assert float(ref["q1"]).hex() == float.fromhex(rec["q1_hex"]).hex()
    assert all(float(r["q1"]).hex() == float(ref["q1"]).hex() for r in res)
    assert all(float(r["q_init"]).hex() == float(ref["q_init"]).hex() for r in res)
    for k in ("u", "g", "d"):
        assert np.array_equal(stack(res, k).view(np.uint64), ref[k].view(np.uint64)), k
    assert np.array_equal(np.asarray(res[0]["hist"]).view(np.uint64), ref["hist"].view(np.uint64))


@pytest.mark.parametrize("world,n,iters,every", [(4, 1023, 60, 0), (4, 1023, 60, 1), (4, 1023, 61, 2), (2, 1023, 60, 3), (8, 2047, 24, 5),
                                                 (4, 2047, 41, 7)])
def test_multirank_strict_published_walk_and_its_completion_by_turns(oracle, tmp_path, monkeypatch, world, n, iters, every):
    # Strict order on several ranks, lean schedule: every rank publishes its plan's briefs and finishes the walk itself (two
    # collectives per dot product).  Where a brief does not hold for the true accumulator the dot product is completed by
    # the ranks walking in turn; the test hook declares every `every`-th dot product failed (1: all of them; 2: every g.g,
    # whose second pass has already run; 3, 5, 7: both kinds, at seams of every parity), 0 leaves the real failures only.
    # Either way: the single-rank bits, on every rank.
    if every:
        monkeypatch.setenv("CG_TEST_FAULT_EVERY", str(every))
    res = run_ranks(world, ["gpu", n, iters, tmp_path, "lean", 0.0, "strict"], tmp_path)
    ref = oracle.cg(n, iters)
    assert all(float(r["q1"]).hex() == float(ref["q1"]).hex() for r in res)
    for k in ("u", "g", "d"):
        assert np.array_equal(stack(res, k).view(np.uint64), ref[k].view(np.uint64)), k
    assert np.array_equal(np.asarray(res[0]["hist"]).view(np.uint64), ref["hist"].view(np.uint64))
    faults = {int(r["faults"]) for r in res}
    assert len(faults) == 1                                        # every rank saw the same ones
    if every:
        # (a failed d.q has the whole iteration redone by turns, its g.g included; the leading iteration of a solve call walks
        #  in turn anyway: at least every `every`-th of the remaining iterations must have gone that way)
        assert faults.pop() >= max(1, (iters - 2) // every - 1)


def test_multirank_strict_ranks_walking_in_turn_still_available(oracle, tmp_path, monkeypatch):
    monkeypatch.setenv("CG_TEST_STRICT_CHAIN", "0")
    n, iters = 1023, 40
    res = run_ranks(4, ["gpu", n, iters, tmp_path, "lean", 0.0, "strict"], tmp_path)
    ref = oracle.cg(n, iters)
    assert all(float(r["q1"]).hex() == float(ref["q1"]).hex() for r in res)
    assert np.array_equal(stack(res, "u").view(np.uint64), ref["u"].view(np.uint64))
    assert all(int(r["faults"]) == 0 for r in res)


@pytest.mark.parametrize("switch", ["classic>lean", "classic>fused", "classic>fused_noq", "lean>classic", "fused>lean"])
def test_multirank_schedule_switch_between_two_solves_keeps_the_bits(oracle, tmp_path, switch):
    # The classic schedule exchanges d's boundary rows at the start of an iteration and changes d afterwards, so
    # the locally kept halo rows of d are stale when its solve ends; a following fused / lean solve starts with an
    # iteration that reads them.  Strict order: every rank must hold the single-rank bits after the switch.
    n, iters = 255, 40
    res = run_ranks(4, ["gpu", n, iters, tmp_path, switch, 0.0, "strict"], tmp_path)
    ref = oracle.cg(n, iters)
    assert all(float(r["q1"]).hex() == float(ref["q1"]).hex() for r in res)
    for k in ("u", "g", "d"):
        assert np.array_equal(stack(res, k).view(np.uint64), ref[k].view(np.uint64)), k
    assert np.array_equal(np.asarray(res[0]["hist"]).view(np.uint64), ref["hist"].view(np.uint64))


@pytest.mark.parametrize("world", [2, 8])
def test_headline_grid_in_row_blocks_reproduces_the_single_rank_bits(cg, tmp_path, world):
    # 16384^2 split over 2 and 8 row blocks (8: the 2048-row blocks of an 8-GPU node), strict order: every block
    # must hold exactly the bits of the single-rank solve, whose 1000-iteration result the reference's own run pins.
    import hashlib
    n, iters = 16383, 24
    res = run_ranks(world, ["gpu", n, iters, tmp_path, "lean", 0.0, "strict", "digest"], tmp_path, timeout=600)
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT).init_problem()
    s.solve(iters // 2)
    r = s.solve(iters - iters // 2)
    assert all(int(x["iters"]) == iters for x in res)
    assert all(float(x["q1"]).hex() == float(r.q1).hex() for x in res)
    rows = (n + 1) // world
    for key, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):
        v = s.vector(which)
        for x in res:
            r0 = int(x["row0"])
            assert str(x[key]) == hashlib.sha256(v[r0:r0 + rows]).hexdigest(), (key, r0)
    s.close()


def test_headline_grid_in_row_blocks_fast_order(cg, tmp_path):
    # the same grid over 4 blocks in the fast order: another (fixed) order of the partial sums, same solve
    n, iters = 16383, 24
    res = run_ranks(4, ["gpu", n, iters, tmp_path, "lean", 0.0, "fast", "digest"], tmp_path, timeout=600)
    s = cg.CGSolver(n).init_problem()
    s.solve(iters // 2)
    r = s.solve(iters - iters // 2)
    assert len({float(x["q1"]).hex() for x in res}) == 1
    assert abs(float(res[0]["q1"]) - r.q1) <= 1e-9 * r.q1
    s.close()


def test_rccl_calls_on_a_single_rank_communicator(cg):
    # One GPU cannot host two RCCL ranks, but a one-rank communicator still runs the library's
    # dlopen of librccl, ncclCommInitRank, ncclAllGather and a grouped ncclSend/ncclRecv (to itself).
    s = cg.CGSolver(255)
    s.comm_init(cg.CGSolver.comm_unique_id())
    s.comm_selftest()
    r = s.init_problem().solve(20)                # the solve itself ignores a one-rank communicator
    ref = cg.CGSolver(255).init_problem().solve(20)
    assert float(r.q1).hex() == float(ref.q1).hex()


@pytest.mark.parametrize("graph", [0, 10])
@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_calls_inside_the_iteration_on_one_rank(cg, graph, overlap):
    # CG_OPT_COMM_REHEARSAL: a single-rank solve that issues the per-iteration RCCL traffic of a multi-rank one
    # (grouped send/receive of two rows to itself, the all-gathers of the partial sums) -- eagerly and captured in
    # a hipGraph, in line and on the second stream.  The traffic must not change a bit of the result.
    ref = cg.CGSolver(1023).configure(graph_iters=graph).init_problem()
    r0 = ref.solve(45)
    s = cg.CGSolver(1023)
    s.comm_init(cg.CGSolver.comm_unique_id())
    s.configure(graph_iters=graph, overlap=overlap, comm_rehearsal=True).init_problem()
    r = s.solve(45)
    assert r.iters == 45
    if not overlap:                                # same tiles, same order of partial sums
        assert float(r.q1).hex() == float(r0.q1).hex()
        assert np.array_equal(s.vector(cg.VEC_U).view(np.uint64), ref.vector(cg.VEC_U).view(np.uint64))
    else:                                          # the edge rows form tiles of their own: another (fixed) order
        assert abs(r.q1 - r0.q1) <= 1e-9 * r0.q1
        assert np.allclose(s.vector(cg.VEC_U), ref.vector(cg.VEC_U), rtol=1e-9, atol=1e-15)


def test_self_addressed_communicator_runs_a_block_of_a_larger_decomposition(cg):
    # timing rehearsal path (tools/rccl_rehearsal.py): rank 1 of 4 with a one-rank RCCL communicator for all peers
    s = cg.CGSolver(1023, row0=256, rows=256, rank=1, nranks=4)
    s.comm_init_self()
    r = s.configure(graph_iters=10).init_problem().solve(33)
    assert r.iters == 33 and r.seconds > 0

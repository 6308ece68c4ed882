"""The C driver (src/cg, the reference's command line) and the utils.h entry points on a real GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits, golden_arrays

pytestmark = pytest.mark.gpu

CG = os.path.join(ROOT, "src", "cg")
CG_MPI = os.path.join(ROOT, "src", "cg_mpi")
MPIRUN = "/opt/conda/bin/mpirun"
LINE = re.compile(r"\[CG\] q1 = (\S+) \((\S+)\) q_init = (\S+) iters=(\d+) np=(\d+) path=(\w+) order=(\w+)")


def run(args, env=None, mpi=0, timeout=240):
    cmd = [CG] + args if not mpi else [MPIRUN, "-np", str(mpi), CG_MPI] + args
    p = subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout + p.stderr
    m = LINE.search(p.stdout)
    assert m, p.stdout
    return p.stdout.splitlines(), m


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
def test_reference_output_lines_and_strict_parity(golden):
    rec = golden[(255, 200, 1, 0.0)]
    lines, m = run(["255"], env={"CG_STRICT": "1"})
    # the reference prints exactly this for ./cg 255 (cg.c:139): the same text, digit for digit
    assert lines[0] == "[INFO] norm_g^2 = %.16f" % float.fromhex(rec["q1_hex"])
    assert re.fullmatch(r"\d+\.\d{16}", lines[1])
    assert m.group(2) == rec["q1_hex"] and int(m.group(4)) == 200 and m.group(7) == "strict"


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
@pytest.mark.parametrize("path", ["solve", "calls"])
def test_driver_paths_fast_order(golden, path):
    rec = golden[(255, 200, 1, 0.0)]
    lines, m = run(["255"], env={"CG_PATH": path})
    q1_ref = float.fromhex(rec["q1_hex"])
    assert abs(float(m.group(1)) - q1_ref) <= 5e-8 * q1_ref
    assert lines[0].startswith("[INFO] norm_g^2 = 0.00000000074333")
    assert m.group(6) == path


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
def test_calls_path_strict_is_bit_identical_and_dumps_u(golden, tmp_path):
    rec = golden[(63, 50, 1, 0.0)]
    prefix = str(tmp_path / "dump")
    _, m = run(["63", "50"], env={"CG_PATH": "calls", "CG_STRICT": "1", "CG_DUMP": prefix})
    assert m.group(2) == rec["q1_hex"]
    want = golden_arrays(rec["arrays"], 63)
    for k in ("u", "g", "d"):
        got = np.fromfile(prefix + f".{k}.f64", dtype="<f8").reshape(64, 64)
        assert np.array_equal(bits(got), bits(want[k])), k


VERIFY = re.compile(r"\[VERIFY\] sum_u = (\S+) \((\S+)\) max_err = (\S+)")


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
@pytest.mark.parametrize("n,iters", [(255, 200), (1023, 200)])
@pytest.mark.parametrize("ranks,path", [(1, "solve"), (1, "calls"), (4, "solve"), (4, "calls")])
def test_verifier_and_dump_on_one_and_four_ranks(golden, tmp_path, n, iters, ranks, path):
    # CG_VERIFY: sum(u) in index order (handed from rank to rank) and max |u - exact|, formed on the GPU, against
    # the numbers the reference run printed for the goldens; CG_DUMP: the whole grid, rank blocks appended in turn.
    rec = golden[(n, iters, 1, 0.0)]
    prefix = str(tmp_path / "dump")
    env = {"CG_STRICT": "1", "CG_VERIFY": "1", "CG_DUMP": prefix, "CG_PATH": path, "CG_TRANSPORT": "host"}
    lines, m = run([str(n), str(iters)], env=env, mpi=ranks if ranks > 1 else 0)
    assert m.group(2) == rec["q1_hex"]
    v = VERIFY.search("\n".join(lines))
    assert v, lines
    assert v.group(2) == rec["sum_u_hex"]
    assert abs(float(v.group(3)) - float(rec["max_err"])) <= 2e-6 * float(rec["max_err"])
    G = n + 1
    u = np.fromfile(prefix + ".u.f64", dtype="<f8")
    assert u.size == G * G
    x = np.arange(G) / n
    assert abs(np.max(np.abs(u.reshape(G, G) - np.outer(x * (1 - x), x * (1 - x)))) - float(rec["max_err"])) <= 2e-6 * float(rec["max_err"])
    for k in ("g", "d"):
        assert np.fromfile(prefix + f".{k}.f64", dtype="<f8").size == G * G
    if "arrays" in rec:
        want = golden_arrays(rec["arrays"], n)
        for k in ("u", "g", "d"):
            assert np.array_equal(bits(np.fromfile(prefix + f".{k}.f64", dtype="<f8").reshape(G, G)), bits(want[k])), k


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
def test_driver_nine_entry_stencil(oracle):
    import sys
    coef = (-1, -1, -1, -1, 8, -1, -1, -1, -1)
    oracle.set_stencil(coef)
    try:
        ref = oracle.cg(63, 40)
    finally:
        oracle.set_stencil()
    _, m = run(["63", "40"], env={"CG_STRICT": "1", "CG_STENCIL": ",".join(str(c) for c in coef)})
    assert m.group(2) == float(ref["q1"]).hex()


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
def test_runtime_iteration_count_and_stop_rule(golden):
    rec = golden[(1023, 1000, 1, 0.0)]
    _, m = run(["1023", "1000"])
    q1_ref = float.fromhex(rec["q1_hex"])
    assert int(m.group(4)) == 1000 and abs(float(m.group(1)) - q1_ref) <= 5e-7 * q1_ref
    rec = golden[(255, 10000, 1, 1e-8)]
    _, m = run(["255", "10000", "1e-8"], env={"CG_STRICT": "1"})
    assert int(m.group(4)) == rec["iters"] and m.group(2) == rec["q1_hex"]


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
def test_bad_arguments_are_rejected():
    p = subprocess.run([CG, "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "[ERROR]" in p.stdout
    p = subprocess.run([CG], capture_output=True, text=True)
    assert p.returncode == 1


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
@pytest.mark.parametrize("ranks", [2, 4])
def test_row_blocks_over_mpi_ranks_strict_is_bit_identical(golden, ranks):
    # Several ranks sharing the one GPU of the test box: row blocks, halo rows through
    # exchange_boundaries, the accumulator of each dot product handed from rank to rank.
    # The result must equal the reference's single-rank numbers bit for bit.
    rec = golden[(255, 200, 1, 0.0)]
    _, m = run(["255"], env={"CG_PATH": "calls", "CG_STRICT": "1"}, mpi=ranks)
    assert m.group(2) == rec["q1_hex"] and int(m.group(5)) == ranks


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
def test_row_blocks_over_mpi_ranks_fast_order(golden):
    rec = golden[(255, 200, 1, 0.0)]
    _, m = run(["255"], env={"CG_PATH": "calls"}, mpi=2)
    q1_ref = float.fromhex(rec["q1_hex"])
    assert abs(float(m.group(1)) - q1_ref) <= 5e-8 * q1_ref


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
@pytest.mark.parametrize("np_", [4, 16])
def test_driver_reproduces_the_references_multirank_output(golden, np_):
    # what `mpirun -np 4|16 ./cg 255` of the reference prints, from one GPU
    rec = golden[(255, 200, np_, 0.0)]
    lines, m = run(["255"], env={"CG_STRICT": "1", "CG_STRICT_NP": str(np_)})
    assert m.group(2) == rec["q1_hex"]
    assert lines[0] == "[INFO] norm_g^2 = %.16f" % float.fromhex(rec["q1_hex"])


@pytest.mark.skipif(not os.path.exists(CG), reason="C driver not built")
@pytest.mark.parametrize("path", ["solve", "calls"])
def test_driver_with_other_stencil_coefficients(golden, path):
    rec = golden[(63, 50, 1, 0.0, (-1, -2, 6, -2, -1))]
    _, m = run(["63", "50"], env={"CG_STRICT": "1", "CG_STENCIL": "-1,-2,6,-2,-1", "CG_PATH": path})
    assert m.group(2) == rec["q1_hex"]


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
@pytest.mark.parametrize("n,iters,np_", [(15, 10, 4), (63, 50, 4), (63, 50, 16), (255, 200, 4), (1023, 200, 4)])
def test_two_dimensional_layout_reproduces_the_reference_on_np_ranks(golden, n, iters, np_):
    # The reference's own decomposition for real: np_ processes (sharing the test GPU), sqrt(np_) x sqrt(np_) chunks,
    # row and column halos, corner points, every rank's chunk summed in the reference's order, MPI_Allreduce on
    # top -- the output of `mpirun -np np_ ./cg n` of the reference itself, bit for bit.
    rec = golden[(n, iters, np_, 0.0)]
    lines, m = run([str(n), str(iters)], env={"CG_PATH": "calls", "CG_LAYOUT": "2d", "CG_STRICT": "1"}, mpi=np_)
    assert m.group(2) == rec["q1_hex"] and int(m.group(5)) == np_
    assert lines[0] == "[INFO] norm_g^2 = %.16f" % float.fromhex(rec["q1_hex"])


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
def test_two_dimensional_layout_other_coefficients_and_fast_order(golden):
    rec = golden[(63, 50, 4, 0.0, (-1, -2, 6, -2, -1))]
    _, m = run(["63", "50"], env={"CG_PATH": "calls", "CG_LAYOUT": "2d", "CG_STRICT": "1", "CG_STENCIL": "-1,-2,6,-2,-1"}, mpi=4)
    assert m.group(2) == rec["q1_hex"]
    rec = golden[(255, 200, 4, 0.0)]
    _, m = run(["255"], env={"CG_PATH": "calls", "CG_LAYOUT": "2d"}, mpi=4)
    q1_ref = float.fromhex(rec["q1_hex"])
    assert abs(float(m.group(1)) - q1_ref) <= 5e-8 * q1_ref


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
@pytest.mark.parametrize("n,iters,np_", [(5, 4, 4), (8, 6, 9), (14, 9, 9)])
def test_two_dimensional_layout_accepts_what_the_reference_accepts(oracle, n, iters, np_):
    # cg.c:21-26 wants (n+1)^2 divisible by the rank count: 6 x 6 points on 4 ranks, 9 x 9 and 15 x 15 on 9 -- grids
    # whose side is NOT divisible by the rank count, which the row-block rule would refuse.  Expected bits on 4 ranks:
    # the restatement in the reference's multi-rank summation layout (pinned against `mpirun -np 4 / 16` of the
    # reference).  Nine ranks: MPICH combines a rank count that is no power of two in another order, which the
    # restatement does not cover -- parity unpinned there, the result is held against the single-rank one at 1e-12.
    side = int(round(np_ ** 0.5))
    _, m = run([str(n), str(iters)], env={"CG_PATH": "calls", "CG_LAYOUT": "2d", "CG_STRICT": "1"}, mpi=np_)
    assert int(m.group(5)) == np_
    if np_ == 4:
        assert float.fromhex(m.group(2)) == float(oracle.cg(n, iters, blocks2d=side)["q1"])
    else:
        q1 = oracle.cg(n, iters)["q1"]
        assert abs(float(m.group(1)) - q1) <= 1e-12 * q1


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
def test_bad_rank_counts_end_with_the_references_message_and_a_clean_exit():
    # every way out passes MPI_Finalize: mpirun reports the plain exit code, no abort noise
    p = subprocess.run([MPIRUN, "-np", "2", CG_MPI, "4"], capture_output=True, text=True, timeout=120)       # 5 rows on 2 ranks
    assert p.returncode != 0 and "not divisible" in p.stdout
    assert "abort" not in (p.stdout + p.stderr).lower() and "BAD TERMINATION" not in p.stderr
    p = subprocess.run([MPIRUN, "-np", "2", CG_MPI, "7"], env=dict(os.environ, CG_PATH="calls", CG_LAYOUT="2d"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "perfect square" in p.stdout


@pytest.mark.skipif(not (os.path.exists(CG_MPI) and os.path.exists(MPIRUN)), reason="MPI driver not built")
@pytest.mark.parametrize("n,iters,np_", [(63, 30, 4), (63, 30, 16), (255, 61, 16)])
def test_nine_point_stencil_in_the_two_dimensional_layout(oracle, n, iters, np_):
    # all nine coefficients on 2 x 2 and 4 x 4 chunks: row and column halos plus the four corner values exchanged with the
    # diagonal neighbours (corner_pad, utils.h:15).  Expected bits: the single-block CPU restatement with the dot products
    # in the reference's multi-rank summation layout.  The corner terms have no counterpart in the reference's own code,
    # so against the reference this parity is unpinned; the five-point part of the same path is pinned by the goldens above.
    coef = (-1, -1, -1, -1, 8, -1, -1, -1, -1)
    side = int(round(np_ ** 0.5))
    oracle.set_stencil(coef)
    try:
        ref = oracle.cg(n, iters, blocks2d=side)
    finally:
        oracle.set_stencil()
    _, m = run([str(n), str(iters)], env={"CG_PATH": "calls", "CG_LAYOUT": "2d", "CG_STRICT": "1",
                                          "CG_STENCIL": ",".join(str(c) for c in coef)}, mpi=np_)
    assert float.fromhex(m.group(2)) == float(ref["q1"]) and int(m.group(5)) == np_      # (the same double: %a and hex() pad differently)

#!/usr/bin/env python3
"""Regenerate the golden vectors in this directory by RUNNING THE REFERENCE.

Runs oracle/_ref/cg_ref -- the reference's own cg.c + utils.c built by
`make -C oracle ref` (iteration count, stop rule and full-precision output are
supplied by oracle/ref_hook.c) -- for the cases below and records

  golden.json            one record per case: q1 as hex float and %.17e, the
                         iteration count, q_init = g0.g0 as hex float, and for
                         single-rank runs sum(u) and
                         max|u - x(1-x)y(1-y)|
  n<N>_i<K>.{u,g,d}.f64  raw little-endian fp64 dumps of the final arrays for
                         the tiny cases (bitwise targets)

Only this script needs /root/reference; the tests read the committed files.
Usage:  python tests/golden/make_golden.py [--big]   (--big adds the long runs)
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "cg_ref")
MPIRUN = "/opt/conda/bin/mpirun"

# (n, iters, np, rtol, dump arrays?)
CASES = [
    (15, 10, 1, 0.0, True), (15, 10, 4, 0.0, False),
    (20, 12, 1, 0.0, True),            # odd grid side (21): rows are not 16-byte aligned
    (63, 50, 1, 0.0, True), (63, 50, 4, 0.0, False), (63, 50, 16, 0.0, False),
    (100, 30, 1, 0.0, False),          # odd grid side (101)
    (256, 200, 1, 0.0, False),         # 257^2 grid
    (255, 200, 1, 0.0, False), (255, 200, 4, 0.0, False), (255, 200, 16, 0.0, False),
    (511, 100, 1, 0.0, False),
    (1023, 200, 1, 0.0, False), (1023, 200, 4, 0.0, False),
    (1023, 1000, 1, 0.0, False),
    (2047, 20, 1, 0.0, False),
    (15, 10000, 1, 1e-8, False), (63, 10000, 1, 1e-8, False), (255, 10000, 1, 1e-8, False),
    (1023, 10000, 1, 1e-8, False),
]
# the same with other coefficients in cg.c:33's array: (n, iters, np, (up, left, centre, right, down), dump?)
STENCIL_CASES = [
    (15, 10, 1, (-1, -2, 6, -2, -1), True),      # anisotropic, symmetric
    (63, 50, 1, (-1, -2, 6, -2, -1), False),
    (63, 50, 4, (-1, -2, 6, -2, -1), False),
    (255, 100, 1, (-3, -1, 9, -1, -3), False),
    (63, 20, 1, (-1, -2, 7, -3, -1), False),     # not symmetric: CG is not meant for it, the arithmetic still is defined
]
BIG = [
    (4095, 200, 1, 0.0, False),
    (4095, 100000, 4, 1e-8, False),
    (16383, 1000, 1, 0.0, False),      # ~40 min and 8.4 GB on one core
]

LINE = re.compile(r"\[ORACLE\] q1 = (\S+) \((\S+)\) iters=(\d+) np=(\d+) qinit = (\S+)"
                  r"(?: sum_u = (\S+) max_err = (\S+))?")


def run_case(n, iters, np_, rtol, dump, stencil=None):
    env = dict(os.environ, REF_ITERS=str(iters))
    if rtol > 0:
        env["REF_RTOL"] = repr(rtol)
    prefix = os.path.join(HERE, f"n{n}_i{iters}")
    if stencil is not None:
        env["REF_STENCIL"] = ",".join(str(x) for x in stencil)
        prefix += "_s" + "_".join(str(x).replace("-", "m") for x in stencil)
    if dump:
        env["REF_DUMP"] = prefix
    cmd = [REF, str(n)] if np_ == 1 else [MPIRUN, "-np", str(np_), REF, str(n)]
    out = subprocess.run(cmd, env=env, check=True, capture_output=True, text=True).stdout
    m = LINE.search(out)
    if not m:
        raise RuntimeError(f"no [ORACLE] line for {cmd}: {out!r}")
    rec = {"n": n, "max_iters": iters, "np": np_, "rtol": rtol,
           "q1": m.group(1), "q1_hex": m.group(2), "iters": int(m.group(3)),
           "q_init_hex": m.group(5)}
    if m.group(6):
        rec["sum_u"] = m.group(6)
        rec["sum_u_hex"] = float(m.group(6)).hex()
        rec["max_err"] = m.group(7)
    if dump:
        rec["arrays"] = os.path.basename(prefix)
    if stencil is not None:
        rec["stencil"] = list(stencil)
    return rec


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first:  make -C oracle ref")
    cases = CASES + (BIG if "--big" in sys.argv else [])
    path = os.path.join(HERE, "golden.json")
    key = lambda r: (r["n"], r["max_iters"], r["np"], r["rtol"], tuple(r.get("stencil", ())))
    old = {}
    if os.path.exists(path):
        for r in json.load(open(path))["cases"]:
            old[key(r)] = r
    recs = []
    for c in cases:
        print("running", c, flush=True)
        recs.append(run_case(*c))
        old.pop(key(recs[-1]), None)
    for n, iters, np_, stencil, dump in STENCIL_CASES:
        print("running", (n, iters, np_, stencil), flush=True)
        recs.append(run_case(n, iters, np_, 0.0, dump, stencil))
        old.pop(key(recs[-1]), None)
    recs += list(old.values())          # keep long-run records made earlier with --big
    recs.sort(key=key)
    doc = {"generator": "tests/golden/make_golden.py (oracle/_ref/cg_ref = reference cg.c+utils.c, "
                        "gcc -std=c99 -O3 -march=x86-64-v3, MPICH 3.3.2)",
           "cases": recs}
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path, len(recs), "cases")


if __name__ == "__main__":
    main()

"""How often does the published walk give up on the benchmark grid in 8 (or P) row blocks?  Ranks share GPU 0 over the
host transport (as the multi-rank GPU tests do): python tools/t_faults8.py [world] [iters]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dist_launch import run_ranks
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
with tempfile.TemporaryDirectory() as tmp:
    res = run_ranks(world, ["gpu", 16383, iters, tmp, "lean", 0.0, "strict", "digest"], tmp)
print(f"{world} ranks, {iters} iterations: q1={float(res[0]['q1']).hex()}  dot products completed by the ranks walking in turn: "
      f"{[int(r['faults']) for r in res]}", flush=True)

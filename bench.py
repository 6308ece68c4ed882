#!/usr/bin/env python3
"""Headline benchmark: fp64 CG iterations/sec on the 16384^2 grid (one GPU, or row blocks over N GPUs).

  python bench.py --gpus N --steps K --warmup W

A step is one CG iteration (stencil matvec + both dot products + the three
vector updates) over the whole grid.  N = 1 runs the 16384 x 16384 grid on one
MI355X; N > 1 (launched by torch.distributed.run, one rank per GPU) row-partitions
the same grid over the ranks (strong scaling), halo rows and dot products going
through RCCL inside the library.  Inputs are the reference's own closed-form
right-hand side (no RNG), generated on the device before the timed region.

Rank 0 prints ONE JSON line.  `value` is whole-job iterations/sec over the reference's timed region
(cg.c:50-55,131-134: the dot product g.g, K iterations, every vector final) in the reference's own summation
order, whose results are bit-identical to the reference's; `other_order` gives the same for the documented
parallel order.  `roofline`
describes the dominant kernel (HIP-event timed on the solver's own stream inside
this run); `cpu_baseline` is the reference's CPU path timed on this box's host
cores on a bounded sample (N = 1 only).
"""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak


def cpu_baseline(n, budget_s=40.0):
    """The reference's CPU path on this box's host cores, bounded sample of the same workload."""
    ref = os.path.join(ROOT, "oracle", "_ref", "cg_ref")
    mpirun = "/opt/conda/bin/mpirun"
    G = n + 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    # the reference accepts perfect-square rank counts whose root divides the grid side (cg.c:15,23);
    # the sample is sized for roughly 15-25 s of CPU work at the rates measured on such hosts
    for np_ in (64, 16, 4, 1):
        iters = {64: 400, 16: 200, 4: 50, 1: 12}[np_]
        root = int(round(np_ ** 0.5))
        if np_ > cores or G % root:
            continue
        if not os.path.exists(ref) or (np_ > 1 and not os.path.exists(mpirun)):
            continue
        cmd = [ref, str(n)] if np_ == 1 else [mpirun, "-np", str(np_), ref, str(n)]
        try:
            out = subprocess.run(cmd, env=dict(os.environ, REF_ITERS=str(iters)), capture_output=True, text=True,
                                 timeout=budget_s * 4).stdout
            m = re.search(r"^\s*([0-9]+\.[0-9]+)\s*$", out, flags=re.M)
            if m and "[ORACLE]" in out:
                sec = float(m.group(1))
                return {"value": iters / sec, "unit": "it/s", "cores": np_, "kind": "reference",
                        "sample": f"{iters} iterations of the {G}x{G} grid, reference cg.c+utils.c, mpirun -np {np_}"}
        except Exception:
            pass
    port = os.path.join(ROOT, "oracle", "cg_oracle")
    iters = 6
    if os.path.exists(port):
        try:
            out = subprocess.run([port, str(n), str(iters)], capture_output=True, text=True, timeout=budget_s * 4).stdout
            m = re.search(r"^\s*([0-9]+\.[0-9]+)\s*$", out, flags=re.M)
            if m:
                sec = float(m.group(1))
                return {"value": iters / sec, "unit": "it/s", "cores": 1, "kind": "port",
                        "sample": f"{iters} iterations of the {G}x{G} grid, CPU restatement oracle/cg_oracle, 1 thread"}
        except Exception:
            pass
    return None


def parity_run(cg, s, n):
    """The same grid once more in the reference's summation order, for the iteration count the reference
    itself was run with (tests/golden/golden.json): untimed side information, the proof that the numbers
    above come from a solver that reproduces the reference bit for bit when asked to."""
    try:
        cases = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]
        rec = max((c for c in cases if c["n"] == n and c["np"] == 1 and c["rtol"] == 0.0 and "stencil" not in c),
                  key=lambda c: c["max_iters"], default=None)
        if rec is None:
            return None
        s.configure(reduce=cg.REDUCE_STRICT, schedule=-1).init_problem()
        r = s.solve(rec["max_iters"])
        return {"order": "strict (one accumulator, index order: utils.c:372-379)", "iters": r.iters,
                "q1_hex": float(r.q1).hex(), "reference_q1_hex": float.fromhex(rec["q1_hex"]).hex(),
                "bit_identical": float(r.q1).hex() == float.fromhex(rec["q1_hex"]).hex(),
                "it_per_s": r.iters / r.seconds,
                "note": "a fresh solve: initial g.g and the first iterations (which have no hypotheses for the "
                        "sequential sums yet) included, the final d / u updates not"}
    except Exception as e:                      # side information must never take the benchmark down
        return {"error": str(e)}


def source_id():
    """Identity of the kernel sources this tree holds (sha256 over csrc/ and the C ABI header)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "conjugate-gradient_amd", "csrc")
    for path in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, "include", "cg_hip.h")]:
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel, G, schedule, reduce="fast"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/prof.sh), or None.

    The counters cannot be collected from inside this process; a profiles/*_traffic.json is reported only if
    it was taken for the same grid, schedule and reduction order AND is stamped with the identity of the kernel
    sources in this tree (FETCH_SIZE doubled, both counters x1024: gfx950 units) -- numbers measured on
    other kernels are not this run's traffic.
    """
    best = None
    pdir = os.path.join(ROOT, "profiles")
    sid = source_id()
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_traffic.json"):
            try:
                doc = json.load(open(os.path.join(pdir, name)))
                if doc.get("grid") == G and doc.get("schedule") == schedule and doc.get("reduce", "fast") == reduce and \
                        doc.get("source_id") == sid and kernel in doc.get("kernels", {}):
                    best = doc["kernels"][kernel]["hbm_bytes_per_launch"]
            except Exception:
                pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--grid-n", dest="n", type=int, default=16383,
                    help="argv n of the reference: grid side is n+1 (--grid-n: the spelling that survives torch.distributed.run)")
    ap.add_argument("--schedule", choices=["classic", "fused", "fused_noq", "lean"], default="lean")
    ap.add_argument("--reduce", choices=["fast", "strict"], default="strict",
                    help="reduction order of the timed run: the reference's own (one accumulator, index order: results "
                         "bit-identical to the reference; default), or the documented parallel order")
    ap.add_argument("--no-side", action="store_true", help="skip the timed region in the other reduction order (side figure)")
    ap.add_argument("--tile-h", type=int, default=0)
    ap.add_argument("--graph-iters", type=int, default=None,
                    help="iterations per captured hipGraph (default: 10 on one GPU; eager launches with RCCL)")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="N > 1: RCCL inside the library, or host-staged rows over a gloo group (diagnostic)")
    ap.add_argument("--prefetch", type=int, default=0, help="rows of loads in flight per thread (0 = per-kernel defaults)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the strict-order run of the golden configuration (N = 1 only, ~6 s)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0, gloo process group, host-staged transport")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    cg = importlib.import_module("conjugate-gradient_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the CG hot path has no CPU fallback")
    if args.share_gpu:
        local_rank, args.transport = 0, "host"
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if args.share_gpu else "cuda"       # where the bench's own tiny collectives live
    if world > 1:
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n, G = args.n, args.n + 1
    if G % world:
        raise SystemExit(f"grid side {G} not divisible by {world} ranks")
    rows = G // world
    sched = {"classic": cg.SCHED_CLASSIC, "fused": cg.SCHED_FUSED, "fused_noq": cg.SCHED_FUSED_NOQ,
             "lean": cg.SCHED_LEAN}[args.schedule]
    s = cg.CGSolver(n, device=local_rank, row0=rank * rows, rows=rows, rank=rank, nranks=world)
    reduce = cg.REDUCE_STRICT if args.reduce == "strict" else cg.REDUCE_FAST
    s.configure(reduce=reduce, schedule=sched, tile_h=args.tile_h, graph_iters=args.graph_iters, prefetch=args.prefetch)
    transport = "none"
    if world > 1:
        transport, why = args.transport, ""
        if transport == "rccl":
            # The library opens its own RCCL communicator (id broadcast over the job's process group).
            # If that fails on any rank, every rank falls back to host-staged rows: slower, but the
            # run still measures the multi-rank solver instead of dying.
            ok = 1
            try:
                ids = [cg.CGSolver.comm_unique_id() if rank == 0 else None]
            except cg.CGError as e:
                ids, ok, why = [None], 0, str(e)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is None:
                ok = 0
            else:
                try:
                    s.comm_init(ids[0])
                    s.comm_selftest()        # tagged all-gather + row exchange: wrong data is caught here, not in q1
                except cg.CGError as e:
                    ok, why = 0, str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                transport = "host"
                if rank == 0:
                    print(f"bench.py: RCCL transport unavailable ({why or 'failed on another rank'}); "
                          f"using host-staged rows", file=sys.stderr, flush=True)
        if transport == "host":
            from importlib import import_module
            cgdist = import_module("conjugate-gradient_amd.dist")
            s.comm_init_host(cgdist.RowTransport(group=None if args.share_gpu else dist.new_group(backend="gloo")))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_region(order):
        """W warm-up iterations, then the reference's timed region (cg.c:50-55,131-134) around exactly K iterations on the
        warm vectors: the dot product q0 = g.g, the K iterations, and the d / u updates the lean schedule's last iteration
        still owes (cg_sync_vectors), device-synchronised at both ends, max over ranks.  Returns (seconds, result, steady):
        `steady` is the library's own clock around a further K iterations that neither begin with g.g nor end with the
        flush -- the per-iteration cost a long solve converges to."""
        s.configure(reduce=cg.REDUCE_STRICT if order == "strict" else cg.REDUCE_FAST, schedule=sched)
        s.init_problem()
        s.set_option(cg.lib.OPT_RESUM_Q0, 1)
        if args.warmup > 0:
            s.solve(args.warmup)                 # untimed
            s.sync_vectors()
            s.solve(args.steps)                  # untimed rehearsal of the timed call: captures the hipGraph for the buffer
            s.sync_vectors()                     # roles that call starts in, sizes the history buffer
        fence()
        t0 = time.perf_counter()
        res = s.solve(args.steps)                # g.g, then exactly K iterations
        s.sync_vectors()                         # ... and what the last of them left owed
        fence()
        elapsed = time.perf_counter() - t0
        s.set_option(cg.lib.OPT_RESUM_Q0, 0)
        # A solve long enough for the library to settle which pass keeps which direction of the buffer pairs for good
        # (cg_solve decides that at its leading iteration from the count of iterations behind it); the K iterations that
        # continue it are the steady state.
        s.solve(args.steps + 2)
        st = s.solve(args.steps)
        steady = st.seconds / max(st.iters, 1)
        if world > 1:
            t = torch.tensor([elapsed, steady], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, steady = float(t[0].item()), float(t[1].item())
        return elapsed, res, steady

    def tune_multirank(order):
        """Several ranks: whether the halo rows travel beside the interior rows (CG_OPT_OVERLAP) and whether the iterations
        run from a captured hipGraph are settled by a short timing of each combination, not by a default: 20 iterations
        each, the slowest rank's time (so every rank sees the same numbers and takes the same decision)."""
        s.configure(reduce=cg.REDUCE_STRICT if order == "strict" else cg.REDUCE_FAST, schedule=sched)
        graphs = [args.graph_iters] if args.graph_iters is not None else ([0, 10] if transport == "rccl" else [0])
        table = {}
        for gi in graphs:
            for ov in (0, 1):
                s.configure(overlap=ov, graph_iters=gi)
                s.init_problem()
                s.solve(12)                          # warm: the leading iteration, the graph capture, the hypotheses
                fence()
                r = s.solve(20)
                t = torch.tensor([r.seconds / max(r.iters, 1)], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                table[(gi, ov)] = 1e3 * float(t.item())
        best = min(table, key=table.get)
        s.configure(overlap=best[1], graph_iters=best[0])
        return {"overlap": best[1], "graph_iters": best[0],
                "overlap_ms": {str(ov): table[(best[0], ov)] for ov in (0, 1)},
                "all_ms": {f"graph{gi}_overlap{ov}": ms for (gi, ov), ms in table.items()}}

    tuning = tune_multirank(args.reduce) if world > 1 else None
    side_order = "fast" if args.reduce == "strict" else "strict"
    side = None if args.no_side else timed_region(side_order)
    elapsed, res, steady = timed_region(args.reduce)      # the headline order runs last: its state is what the roofline timings follow

    out = None
    if rank == 0:
        bpp = s.bytes_per_point
        its = args.steps / elapsed
        out = {
            "metric": "cg_iterations_per_sec", "value": its, "unit": "it/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{G}x{G} fp64 grid, fixed-count CG, 5-point stencil "
                                   f"(BASELINE.json configs[{2 if world == 1 else 3}])",
                       "grid": G, "rows_per_gpu": rows, "schedule": args.schedule, "reduce": args.reduce,
                       "bytes_per_point_per_iter": bpp,
                       "graph_iters": s.get_option(cg.lib.OPT_GRAPH_ITERS), "overlap": s.get_option(cg.lib.OPT_OVERLAP) if world > 1 else 0,
                       # several ranks: ms per iteration with the halo exchange in line ("0") and beside the interior rows ("1"),
                       # measured before the timed region; the faster setting is the one in effect
                       "overlap_ms": tuning["overlap_ms"] if tuning else None, "tuning_ms": tuning["all_ms"] if tuning else None,
                       "transport": transport, "parallelism": f"rowblocks{world}",
                       # sets of vectors cg_create tries before it keeps the fastest placement (blocks of >= 2^27 points)
                       "placement_draws": (int(os.environ.get("CG_PLACEMENT_DRAWS", "8")) if rows * G >= 1 << 27 else 1)},
            "hbm_gbs_per_gpu": bpp * rows * G * its / 1e9,
            "hbm_frac_of_peak": bpp * rows * G * its / 1e9 / HBM_PEAK_GBS,
            "q1": res.q1, "iters_total": args.warmup + 3 * args.steps + 2,
            "timed_region": "q0 = g.g, K iterations, the owed d / u updates (cg.c:50-55,131-134), warm vectors",
            "steady_state": {"ms_per_step": 1e3 * steady, "it_per_s": 1.0 / steady,
                             "hbm_frac_of_peak": bpp * rows * G / steady / 1e9 / HBM_PEAK_GBS,
                             "note": "K further iterations without the leading g.g and the trailing updates"},
            "source_id": source_id(),
        }
        if side is not None:
            out["other_order"] = {"reduce": side_order, "it_per_s": args.steps / side[0], "ms_per_step": 1e3 * side[0] / args.steps,
                                  "steady_ms_per_step": 1e3 * side[2], "q1": side[1].q1}
    # per-kernel roofline: HIP events on the solver's stream, this run, this GPU (rank 0's row block)
    if rank == 0:
        names = {cg.SCHED_CLASSIC: ("stencil", "update", "dupdate"), cg.SCHED_FUSED: ("stencil", "update"),
                 cg.SCHED_FUSED_NOQ: ("stencil", "update"),
                 # lean: pass 1 every iteration; pass 2 alternates between the plain form and the one that
                 # also applies two iterations' u updates
                 cg.SCHED_LEAN: ("pass1_dq", "pass2_g", "pass2_g_u")}[sched]

        def kernel_roofline(order):
            kern = {}
            for which, name in enumerate(names):
                ms, nbytes = s.time_kernel(which, reps=20)
                share = 0.5 if (sched == cg.SCHED_LEAN and which > 0) else 1.0
                kern[name] = {"ms": ms, "bytes": nbytes, "gbs": nbytes / ms / 1e6, "launches_per_iter": share}
            dom = max(kern, key=lambda k: kern[k]["ms"] * kern[k]["launches_per_iter"])
            roof = {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["gbs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kern[dom]["gbs"] / HBM_PEAK_GBS,
                    "traffic": profiled_traffic(dom, G, args.schedule, order) if world == 1 else None,
                    "algorithmic_bytes_per_launch": kern[dom]["bytes"], "ms_per_launch": kern[dom]["ms"]}
            return roof, kern

        out["roofline"], out["kernels"] = kernel_roofline(args.reduce)
        out["roofline"]["whole_iteration_frac"] = out["steady_state"]["hbm_frac_of_peak"]
        out["roofline"]["timed_region_frac"] = out["hbm_frac_of_peak"]
        # the same grid from a fresh start in the reference's summation order for the iteration count the reference
        # itself was run with: q1 must equal the reference's bit for bit (untimed proof; its rate is side information)
        out["parity"] = parity_run(cg, s, n) if (world == 1 and not args.no_parity) else None
        out["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(n)
    if world > 1:
        dist.barrier()                       # no rank tears its communicator down while another still solves
    s.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

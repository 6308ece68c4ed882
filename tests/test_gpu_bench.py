"""The benchmark's one-line JSON contract, on a small grid so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "2047", "--steps", "7", "--warmup", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "it/s" and d["dtype"] == "f64" and d["value"] > 0
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the golden set has a single-rank record for this grid: the parity side run must reproduce it
    assert d["parity"]["bit_identical"] is True


def test_bench_two_ranks_sharing_the_gpu():
    # the N > 1 code path of the benchmark (row blocks, transport self-test, max-over-ranks timing), rehearsed
    # with two processes on one GPU over the host-staged transport
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--grid-n", "2047", "--steps", "6", "--warmup", "2", "--share-gpu"],
                       capture_output=True, text=True, timeout=400, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rows_per_gpu"] == 1024 and d["config"]["transport"] == "host"
    assert d["scaling"] == "strong" and d["value"] > 0 and d["cpu_baseline"] is None
    # overlap of the halo exchange: measured both ways before the timed region, the faster one in effect
    om = d["config"]["overlap_ms"]
    assert set(om) == {"0", "1"} and all(v > 0 for v in om.values())
    assert d["config"]["overlap"] == int(min(om, key=om.get))
    assert d["config"]["reduce"] == "strict" and d["other_order"]["reduce"] == "fast"

import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference arithmetic (checker only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    mod = importlib.import_module("oracle")
    mod.build()
    mod.load()
    return mod


@pytest.fixture(scope="session")
def cg():
    """The product package (directory name has a hyphen)."""
    return importlib.import_module("conjugate-gradient_amd")


@pytest.fixture(scope="session")
def golden():
    """Records produced by running the reference itself (tests/golden/make_golden.py)."""
    doc = json.load(open(os.path.join(GOLDEN_DIR, "golden.json")))
    # default coefficients: (n, iters, np, rtol); other coefficient sets append the 5-tuple
    return {(r["n"], r["max_iters"], r["np"], r["rtol"]) + ((tuple(r["stencil"]),) if "stencil" in r else ()): r
            for r in doc["cases"]}


def golden_arrays(stem, n):
    G = n + 1
    return {k: np.fromfile(os.path.join(GOLDEN_DIR, f"{stem}.{k}.f64"), dtype="<f8").reshape(G, G)
            for k in ("u", "g", "d")}


def bits(a):
    """View fp64 data as integers so that comparisons are bit-exact (distinguishes -0.0, NaN payloads)."""
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def seq_sum(oracle_mod, a):
    """Strictly sequential sum, the order the reference's own loops use."""
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    return oracle_mod.dot(a, np.ones_like(a))

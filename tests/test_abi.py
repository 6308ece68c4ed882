"""The C-ABI library loads and exports every symbol include/*.h declares (CPU only, no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"\btypedef\b[^;{]*(\{[^}]*\})?[^;]*;", "", text)      # struct and function-pointer typedefs
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions("cg_hip.h")
    for must in ("cg_create", "cg_solve", "cg_stencil", "cg_dot", "cg_axpy", "cg_xpay", "cg_init_d", "cg_init_g",
                 "cg_comm_init", "cg_time_kernel"):
        assert must in names


def test_library_exports_every_declared_symbol(cg):
    lib = cg.lib.load()
    for name in declared_functions("cg_hip.h"):
        assert hasattr(lib, name), f"{name} declared in include/cg_hip.h but not exported"
    # and the Python binding covers the same set
    assert sorted(cg.lib.SIGNATURES) == declared_functions("cg_hip.h")


def test_compat_header_symbols_exported_by_driver_object():
    # include/utils.h re-exports the reference's nine names (utils.h:34-42); they are
    # implemented in src/utils_hip.c on top of the C ABI.
    names = declared_functions("utils.h")
    want = {"apply_stencil_serial", "apply_stencil_parallel", "exchange_boundaries", "init_locald", "init_localg",
            "print_local2dmesh", "dot", "init_mpi_settings", "free_struct_elems"}
    assert want <= set(names)
    src = open(os.path.join(ROOT, "src", "utils_hip.c")).read()
    for n in want:
        assert re.search(r"\b%s\s*\(" % n, src), n


def test_reference_driver_compiles_against_the_dropin_header():
    # The reference's own cg.c (its call sites cg.c:30-146 of the nine utils.h entry points, its struct
    # member accesses) must compile unchanged against include/utils.h.
    ref = "/root/reference/cg.c"
    mpi_inc = "/opt/conda/include"
    if not (os.path.exists(ref) and os.path.exists(os.path.join(mpi_inc, "mpi.h"))):
        pytest.skip("reference sources or an MPI header are not present on this machine")
    import subprocess
    p = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-I", mpi_inc,
                        "-include", "mpi.h", ref], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_unequal_row_blocks_are_rejected(cg):
    # the all-gathers of the partial sums need the same count on every rank: equal blocks in rank order
    lib = cg.lib.load()
    ctx = ctypes.c_void_p()
    assert lib.cg_create(ctypes.byref(ctx), 0, 20, 0, 10, 0, 2) != 0       # 21 rows over 2 ranks
    assert b"equal blocks" in lib.cg_last_error()
    assert lib.cg_create(ctypes.byref(ctx), 0, 19, 5, 10, 0, 2) != 0       # right size, wrong place
    assert b"equal blocks" in lib.cg_last_error()


def test_no_gpu_means_loud_failure(cg):
    # On a machine without a GPU every compute entry point must fail, never fall back.
    if cg.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(cg.CGError):
        cg.CGSolver(15)
    with pytest.raises(cg.CGError):
        cg.DeviceVector(16)


def test_argument_errors_are_reported(cg):
    lib = cg.lib.load()
    ctx = ctypes.c_void_p()
    assert lib.cg_create(ctypes.byref(ctx), 0, 1, 0, 2, 0, 1) != 0      # n < 2
    assert b"n must be" in lib.cg_last_error()
    assert lib.cg_set_option(None, 1, 0) != 0
    assert lib.cg_solve(None, 1, 0.0, None) != 0

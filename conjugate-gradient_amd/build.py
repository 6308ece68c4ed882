"""In-tree build of libcg_hip.so (hipcc, gfx950 only) and of the C driver."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcg_hip.so")

# -ffp-contract=off: the reference rounds every product before the add that
# follows it; hipcc's default would fuse them into v_fma_f64.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-I/opt/rocm/include", "-Wall", "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_lib(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "cg_hip.h")]
    if force or _newer(LIB, srcs):
        cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, "cg_lib.hip"), "-o", LIB, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_driver(force=False, verbose=False):
    """The C driver (src/cg.c + src/utils_hip.c): plain C, linked against libcg_hip.so."""
    src = os.path.join(ROOT, "src")
    out = os.path.join(src, "cg")
    files = [os.path.join(src, f) for f in ("cg.c", "utils_hip.c")]
    if not all(os.path.exists(f) for f in files):
        return None
    deps = files + [os.path.join(ROOT, "include", "utils.h"), os.path.join(ROOT, "include", "cg_hip.h"), LIB]
    if force or _newer(out, deps):
        common = ["-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O2", "-Wall", "-I", os.path.join(ROOT, "include")] + files
        link = ["-L", HERE, "-lcg_hip", "-Wl,-rpath,$ORIGIN/../conjugate-gradient_amd", "-lm"]
        cmd = ["gcc"] + common + ["-o", out] + link
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        # one rank per GPU under mpirun, where an MPI is installed (MPICH under /opt/conda in this image)
        # (gcc directly, not the mpicc wrapper: the wrapper's -L/opt/conda/lib would make the linker
        # resolve the HIP runtime's libstdc++ against conda's older copy)
        mpi = "/opt/conda"
        if os.path.exists(os.path.join(mpi, "lib", "libmpi.so")):
            cmd = ["gcc", "-DCG_USE_MPI", "-I", os.path.join(mpi, "include")] + common + ["-o", out + "_mpi"] + link + \
                  [os.path.join(mpi, "lib", "libmpi.so"), "-Wl,--enable-new-dtags", "-Wl,-rpath," + os.path.join(mpi, "lib"),
                   "-Wl,--allow-shlib-undefined"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_driver(force=True, verbose=True))

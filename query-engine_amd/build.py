"""Builds libqe_gpu.so (HIP kernels + host runtime + C ABI) for gfx950 with hipcc.

The library is built in-tree (query-engine_amd/libqe_gpu.so) so that it travels
with the source tree to a GPU box; objects are rebuilt only when a source or
header is newer.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libqe_gpu.so")
SOURCES = ["qe_kernels.hip", "qe_static.hip", "qe_core.cpp", "qe_exec.cpp"]
# -ffp-contract=off: `a + b*c` must round twice like the reference's separate mul and add passes.
# -munsafe-fp-atomics: native ds_add_f64 / global_atomic_add_f64 instead of compare-swap loops.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    objs = []
    relink = force or not os.path.exists(LIB)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
    if relink or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

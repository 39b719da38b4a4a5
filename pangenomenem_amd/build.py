"""Build the HIP shared library in-tree: pangenomenem_amd/lib/libnem_mi355x.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off and no fast-math are part of the numeric
contract: the reference is plain SSE2 arithmetic with separate multiply/add.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("nem_kernels.hip", "nem_engine.hip", "nem_io.cpp", "nem_capi.cpp")]
HDR = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".hpp")) + \
      [os.path.join(HERE, "..", "include", "nem_mi355x.h")]
LIB = os.path.join(HERE, "lib", "libnem_mi355x.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-std=c++17", "-ldl",
         "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRC + HDR + [os.path.abspath(__file__)])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("NEM_EXTRA_HIPCC_FLAGS", "").split() + ["-o", LIB] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Build the HIP shared library in-tree: pangenomenem_amd/lib/libnem_mi355x.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off and no fast-math are part of the numeric
contract: the reference is plain SSE2 arithmetic with separate multiply/add.

The translation units are compiled side by side (one hipcc per unit, objects under lib/obj/, only the units
whose sources or headers are newer than their object) and linked: the kernel units are most of the minute a
single command took.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
UNITS = ("nem_kernels.hip", "nem_sweep.hip", "nem_chunks.hip", "nem_engine.hip", "nem_io.cpp", "nem_capi.cpp")
SRC = [os.path.join(HERE, "csrc", f) for f in UNITS]
HDR = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".hpp")) + \
      [os.path.join(HERE, "..", "include", "nem_mi355x.h")]
LIB = os.path.join(HERE, "lib", "libnem_mi355x.so")
OBJ = os.path.join(HERE, "lib", "obj")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++17",
          "-Wall", "-Wno-unused-function"]
LFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared", "-ldl"]


def _extra():
    return os.environ.get("NEM_EXTRA_HIPCC_FLAGS", "").split()


def _stamp():
    """what an object depends on besides its source: the headers, this file, the extra flags"""
    return max(os.path.getmtime(p) for p in HDR + [os.path.abspath(__file__)])


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src) + ".o")


def _flags_file():
    return os.path.join(OBJ, "flags.txt")


def _stale_units():
    flags = " ".join(CFLAGS + _extra())
    try:
        same_flags = open(_flags_file()).read() == flags
    except OSError:
        same_flags = False
    stamp = _stamp()
    out = []
    for s in SRC:
        o = _obj(s)
        if not same_flags or not os.path.isfile(o) or os.path.getmtime(o) < max(os.path.getmtime(s), stamp):
            out.append(s)
    return out


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRC + HDR + [os.path.abspath(__file__)])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stale = SRC if force else _stale_units()

    def compile_one(src):
        cmd = [hipcc] + CFLAGS + _extra() + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(stale), os.cpu_count() or 1, 6))) as pool:
        list(pool.map(compile_one, stale))
    with open(_flags_file(), "w") as f:
        f.write(" ".join(CFLAGS + _extra()))
    cmd = [hipcc] + LFLAGS + ["-o", LIB + ".tmp"] + [_obj(s) for s in SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

#!/usr/bin/env python3
"""Where k_finish spends its time (development probe; needs a build with NEM_EXTRA_HIPCC_FLAGS=-DNEM_PHASE_PROF):
block 0 stamps the 100 MHz wall clock at its phase boundaries.

    NEM_EXTRA_HIPCC_FLAGS=-DNEM_PHASE_PROF python pangenomenem_amd/build.py --force
    python3 profiles/finish_phases.py 200000 5000
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, load_library


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    lib = load_library()
    x, _ = synth.ushaped_pa_matrix(n, d, seed=1)
    nei = synth.contiguity_graph(n, seed=1)
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x)
    eng.set_graph(nei)
    eng.set_params(*synth.default_init(d))
    eng.configure(algo="ncem", beta=0.5, it_max=4, cvtest="none")
    eng.run()
    lib.nemgpu_debug_phases.argtypes = [C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * 32)()
    rc = lib.nemgpu_debug_phases(out)
    t = [int(v) for v in out]
    names = {0: "entry", 1: "centres", 2: "dispersion done", 3: "flags", 4: "table entries", 5: "ff tables",
             8: "  closed-form test", 9: "  staged", 10: "  chains", 11: "    inertia chain", 12: "    N_KD closed form"}
    base = t[0]
    for i in (0, 1, 8, 9, 12, 11, 10, 2, 3, 4, 5):
        print("%-22s +%.2f us" % (names[i], (t[i] - base) / 100.0))
    return rc


if __name__ == "__main__":
    sys.exit(main())

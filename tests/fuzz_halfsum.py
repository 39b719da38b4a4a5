#!/usr/bin/env python3
"""Extended CPU fuzz of the piecewise inertia chain (csrc/nem_halfsum.hpp): the host emulation of the device procedure
against the plain float loop on random chains of non-negative multiples of 1/2.

    python tests/fuzz_halfsum.py [chains] > profiles/r03_halfsum_fuzz.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import build, engine  # noqa: E402


def main():
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
    build.build()
    lib = engine.load_library()
    lib.nemgpu_halfsum_host.restype = C.c_float
    lib.nemgpu_halfsum_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.Generator(np.random.PCG64(20261005))
    bad, stepped_total, total = 0, 0, 0
    for t in range(chains):
        n = int(rng.integers(1, 8193))
        kind = t % 6
        if kind == 0:                                  # class counts: min(s1, N_k - s1), some N_k / 2
            nk = int(rng.integers(2, 1 << 24))
            s1 = rng.integers(0, nk + 1, size=n)
            x = np.minimum(s1, nk - s1).astype(np.float64)
            if t % 12 == 0:
                x[rng.random(n) < 0.1] = nk / 2.0
        elif kind == 1:
            x = rng.integers(0, 1 << int(rng.integers(1, 25)), size=n) / 2.0
        elif kind == 2:
            x = rng.integers(0, 1 << int(rng.integers(1, 25)), size=n) / 2.0
            x[rng.random(n) < rng.random()] = 0
        elif kind == 3:                                # powers of two: ties on every grid
            x = np.ldexp(1.0, rng.integers(-1, 23, size=n))
        elif kind == 4:                                # every element an exact tie at one level
            j = int(rng.integers(1, 10))
            x = np.minimum((rng.integers(0, 1 << 14, size=n) * 2 + 1) * (1 << (j - 1)) / 2.0, 2 ** 24 - 1)
        else:                                          # huge and tiny addends mixed
            x = np.concatenate([rng.integers(1 << 20, 1 << 24, size=n // 2), rng.integers(0, 4, size=n - n // 2) / 2.0])
            rng.shuffle(x)
        x = np.ascontiguousarray(x, np.float32)
        st = C.c_int(0)
        waves = (16, 1, 2, 7, 13)[t % 5]
        want = lib.nemgpu_halfsum_host(x.ctypes.data, n, 0, None)
        got = lib.nemgpu_halfsum_host(x.ctypes.data, n, waves, C.byref(st))
        stepped_total += st.value
        total += n
        if np.float32(want).view(np.uint32) != np.float32(got).view(np.uint32):
            bad += 1
    print(json.dumps(dict(chains=chains, mismatches=bad, elements=total, stepped_fraction=stepped_total / max(total, 1),
                          wavefronts_emulated=[16, 1, 2, 7, 13],
                          what="halfsum_host (the device procedure, thread by thread) against the plain float loop"), indent=1))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

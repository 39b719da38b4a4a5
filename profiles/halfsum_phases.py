#!/usr/bin/env python3
"""Where the piecewise inertia chain (csrc/nem_halfsum.hpp) spends its time, from the kernel's own clock."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import build, engine  # noqa: E402

build.build()
lib = engine.load_library()
lib.nemgpu_halfsum_profile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
rng = np.random.Generator(np.random.PCG64(5))
out = []
for n, nk in ((5000, 70000), (1000, 17000), (8192, 200000), (500, 7000)):
    s1 = rng.integers(0, nk + 1, size=n)
    x = np.ascontiguousarray(np.minimum(s1, nk - s1), np.float32)
    for waves in (16, 1):
        us = (C.c_double * 5)()
        rc = lib.nemgpu_halfsum_profile(x.ctypes.data, n, waves, 0, us)
        out.append(dict(n=n, class_size=nk, wavefronts=waves, rc=rc, prefix_us=us[0], walk_us=us[1], scan_us=us[2], ordered_pass_us=us[3],
                        pieces_total_us=us[0] + us[1] + us[2] + us[3], plain_chain_us=us[4]))
print(json.dumps(out, indent=1))

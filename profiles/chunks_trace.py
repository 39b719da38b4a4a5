#!/usr/bin/env python3
"""One nemgpu_solve_many job for a kernel trace: P configs[1]-sized problems as bit rows, 8 workers, groups of 32."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth  # noqa: E402
from pangenomenem_amd.batch import solve_many  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = synth.make_config("C2")
rng = np.random.default_rng(0)
d = cfg["x"].shape[1]
pad = (-((d + 7) // 8)) % 4
probs = []
for p in range(P):
    x = cfg["x"][:, rng.permutation(d)]
    b = np.pad(np.packbits(x, axis=1, bitorder="little"), ((0, 0), (0, pad)))
    probs.append((np.ascontiguousarray(b).view(np.uint32), cfg["nei"], 3, cfg["prop"], cfg["center"], cfg["disp"]))
solve_many(probs[:64], W, algo="ncem", beta=0.5, disper="sk_")
for rep in range(3):
    t0 = time.perf_counter()
    solve_many(probs, W, algo="ncem", beta=0.5, disper="sk_")
    dt = time.perf_counter() - t0
    print("rep %d: %.2f ms, %.0f problems/s" % (rep, dt * 1e3, P / dt), flush=True)

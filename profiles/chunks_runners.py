#!/usr/bin/env python3
"""nemgpu_solve_many on one GPU with one, two or three concurrent lock-step runners (a device list that names the
device more than once): P configs[1]-sized problems as bit rows."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth  # noqa: E402
from pangenomenem_amd.batch import solve_many  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.make_config("C2")
rng = np.random.default_rng(0)
d = cfg["x"].shape[1]
pad = (-((d + 7) // 8)) % 4
probs = []
for p in range(P):
    x = cfg["x"][:, rng.permutation(d)]
    b = np.pad(np.packbits(x, axis=1, bitorder="little"), ((0, 0), (0, pad)))
    probs.append((np.ascontiguousarray(b).view(np.uint32), cfg["nei"], 3, cfg["prop"], cfg["center"], cfg["disp"]))
kw = dict(algo="ncem", beta=0.5, disper="sk_")
want = solve_many(probs[:64], 8, **kw)
out = {}
for runners in (1, 2, 3):
    for workers in (8, 12):
        devs = None if runners == 1 else [0] * runners
        got = solve_many(probs[:64], workers, devices=devs, **kw)
        assert all(np.array_equal(a["c"], b["c"]) and a["iters"] == b["iters"] for a, b in zip(want, got))
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            solve_many(probs, workers, devices=devs, **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["runners %d workers %d" % (runners, workers)] = round(P / best)
        print(runners, workers, round(P / best), flush=True)
print(json.dumps(out))

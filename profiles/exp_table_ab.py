#!/usr/bin/env python3
"""exp(beta * context) from the per-beta table against evaluated (NEM_MI355X_EXP_TABLE=0), on graphs with the edge weights
PPanGGOLiN writes (synth 'adjacency': counts of organisms): one engine (bench-style restart cycles) and 64 problems in
lock step.  Run once per setting; prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth  # noqa: E402
from pangenomenem_amd.engine import NemEngine, Result  # noqa: E402


def main():
    n, d = 20000, 500
    x, _ = synth.ushaped_pa_matrix(n, d, 2)
    nei = synth.contiguity_graph(n, 2, weights="adjacency", counts=x.sum(axis=1))
    prop, center, disp = synth.default_init(d)
    out = dict(table=os.environ.get("NEM_MI355X_EXP_TABLE", "1") != "0", families=n, organisms=d)
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp)
    eng.configure(algo="ncem", beta=0.5, disper="sk_", cvtest="clas", it_max=100)
    first = eng.run()
    cyc = first["iters"]
    eng.configure(algo="ncem", beta=0.5, disper="sk_", cvtest="none", it_max=100)
    eng.set_graph_policy(True)
    for _ in range(20):
        eng.restart_iterate(cyc)
    blocks = []
    for _ in range(15):
        t0 = time.perf_counter()
        for _ in range(10):
            eng.restart_iterate(cyc)
        blocks.append((time.perf_counter() - t0) / (10 * cyc))
    out["solo_ms_per_iteration"] = sorted(blocks)[len(blocks) // 2] * 1e3
    out["iters_to_converge"] = cyc
    out["labels_sum"] = int(first["c"].argmax(1).sum())
    eng.close()
    B = 64
    rng = np.random.default_rng(0)
    engs = []
    for p in range(B):
        xp = np.ascontiguousarray(x[:, rng.permutation(d)])
        e = NemEngine(n, d, 3)
        e.set_matrix(xp); e.set_graph(nei); e.set_params(prop, center, disp)
        e.configure(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=1)
        engs.append(e)
    lib = engs[0].lib
    handles = (C.c_void_p * B)(*[e._h for e in engs])
    res = (Result * B)()
    best = None
    for _ in range(6):
        t0 = time.perf_counter()
        assert lib.nemgpu_run_many(handles, B, res) == 0
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    iters = sum(r.iters for r in res)
    out["lockstep_64_us_per_problem_iteration"] = best * 1e6 / iters
    out["lockstep_64_em_iterations"] = iters
    for e in engs:
        e.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

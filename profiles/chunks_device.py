#!/usr/bin/env python3
"""Chunk throughput from a RESIDENT master: P samples of 500 organisms out of one 20 000 x 5 000 pangenome,
(a) formed on the device and solved by nemgpu_solve_chunks, (b) formed on the host (numpy, not timed) and solved by
nemgpu_solve_many from bit rows (uploads included) -- whole problems per second, best of 3 jobs each.  One JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth  # noqa: E402
from pangenomenem_amd.batch import solve_many  # noqa: E402
from pangenomenem_amd.chunks import Master, form_chunk_host, pack_rows  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n, d, dc = 20000, 5000, 500
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    t0 = time.perf_counter()
    x, (ptr, idx), eb = synth.master_pangenome(n, d, 2)
    t_gen = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    subs = [rng.permutation(d)[:dc] for _ in range(P)]
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=1)
    t0 = time.perf_counter()
    m = Master(x, ptr, idx, eb)
    t_master = time.perf_counter() - t0
    out = dict(master=[n, d], sample_organisms=dc, problems=P, workers=workers, master_edges=int(len(idx)), master_upload_s=t_master,
               master_generation_s=t_gen, device={}, host_formed={})
    m.solve_chunks(subs[:64], workers=workers, group=32, want_params=False, **cfg)          # (fills the library's pools)
    for group in (32, 64):
        best, lib_best, res = None, None, None
        for _ in range(3):
            t0 = time.perf_counter()
            res = m.solve_chunks(subs, workers=workers, group=group, want_params=False, **cfg)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            lib_best = m.last_call_seconds if lib_best is None else min(lib_best, m.last_call_seconds)
        out["device"]["group_%d" % group] = dict(seconds=best, chunks_per_s=P / best, library_call_seconds=lib_best, library_chunks_per_s=P / lib_best,
                                                 em_iterations=int(sum(r["iters"] for r in res)),
                                                 mean_families=float(np.mean([r["n"] for r in res])), mean_edges=float(np.mean([r["nnz"] for r in res])))
    # the same job cut off after ONE EM iteration: what is left is the pipeline around the EM -- forming (or uploading) a
    # problem, its start, its results
    short = dict(cfg, it_max=1)
    best = None
    for _ in range(3):
        m.solve_chunks(subs, workers=workers, group=32, want_params=False, **short)
        best = m.last_call_seconds if best is None else min(best, m.last_call_seconds)
    out["device"]["one_iteration"] = dict(library_call_seconds=best, library_chunks_per_s=P / best)
    # the same samples formed on the host, through nemgpu_solve_many (a quarter of them: forming one in numpy takes ~50 ms)
    Q = min(P, 64)
    prop, center, disp = synth.default_init(dc)
    t0 = time.perf_counter()
    host = [form_chunk_host(x, ptr, idx, eb, s) for s in subs[:Q]]
    out["host_formation_s_per_chunk_numpy"] = (time.perf_counter() - t0) / Q
    probs = [(pack_rows(xc), nei, 3, prop, center, disp) for xc, nei, _ in host]
    solve_many(probs, workers, group=32, **cfg)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        want = solve_many(probs, workers, group=32, **cfg)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["host_formed"] = dict(problems=Q, seconds=best, chunks_per_s=Q / best, note="formation not timed; bit rows, graph and parameters uploaded per chunk")
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        solve_many(probs, workers, group=32, **short)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["host_formed"]["one_iteration"] = dict(seconds=best, chunks_per_s=Q / best)
    got = m.solve_chunks(subs[:Q], workers=workers, group=32, **cfg)
    out["identical_to_host_formed"] = bool(all(np.array_equal(g["labels"], w["c"].argmax(1)) and g["iters"] == w["iters"] and
                                               np.array_equal(g["disp"], w["disp"]) for g, w in zip(got, want)))
    m.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

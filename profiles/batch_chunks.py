#!/usr/bin/env python3
"""Chunk throughput on one GPU, whole problems (upload included): P independent configs[1]-sized problems through the
drop-in nem() (files on /tmp) on 1..16 worker threads (pangenomenem_amd.batch.nem_many), and in memory through
solve_many = ONE nemgpu_solve_many call -- engines built on `w` threads of the library, groups of 32 in lock step
(nemgpu_run_many), results fetched while the next groups are built -- from byte matrices and from bit rows.
Prints one JSON object."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import nemfiles, synth  # noqa: E402
from pangenomenem_amd.batch import nem_many, solve_many  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = synth.make_config("C2")
    root = tempfile.mkdtemp(prefix="nemchunks_")
    rng = np.random.default_rng(0)
    calls, problems = [], []
    for p in range(P):
        x = np.ascontiguousarray(cfg["x"][:, rng.permutation(cfg["x"].shape[1])])      # another sample of organisms
        if p < 16:                                                                      # (0.6 s of Python per set of files)
            base = nemfiles.write_nem_inputs(os.path.join(root, str(p)), x, cfg["nei"], cfg["prop"], cfg["center"], cfg["disp"])
            calls.append(dict(Fname=base.encode(), nk=3, algo=b"ncem", beta=0.5, convergence=b"clas", convergence_th=1e-8,
                              format=b"fuzzy", it_max=100, dolog=True, model_family=b"bern", proportion=b"pk",
                              dispersion=b"sk_", init_mode=2))
        problems.append((x, cfg["nei"], 3, cfg["prop"], cfg["center"], cfg["disp"]))
    nem_many(calls[:2], 1)                                                              # load, context, page cache
    out = dict(problems=P, problems_as_files=len(calls), shape=list(cfg["x"].shape), host_cores=os.cpu_count(), files={}, in_memory={}, in_memory_bit_rows={})
    for w in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        rcs = nem_many(calls, w)
        dt = time.perf_counter() - t0
        assert all(rc == 0 for rc in rcs), (w, rcs, open(calls[0]['Fname'].decode() + '.stderr').read()[-400:])
        out["files"][str(w)] = dict(seconds=dt, problems_per_s=len(calls) / dt)
    ref_uf = open(calls[0]["Fname"].decode() + ".uf", "rb").read()
    nem_many(calls[:1], 1)
    assert open(calls[0]["Fname"].decode() + ".uf", "rb").read() == ref_uf          # same answer alone and in a crowd
    solo = solve_many(problems[:1], 1, algo="ncem", beta=0.5, disper="sk_")[0]
    t0 = time.perf_counter()
    solve_many(problems, 8, algo="ncem", beta=0.5, disper="sk_")                        # (fills the library's pools)
    dt = time.perf_counter() - t0
    out["in_memory_first_call"] = dict(workers=8, seconds=dt, problems_per_s=P / dt,
                                       note="the first job of the process: empty resource pool, no captured batch")
    d = cfg["x"].shape[1]
    pad = (-((d + 7) // 8)) % 4
    as_bits = [(np.ascontiguousarray(np.pad(np.packbits(p[0], axis=1, bitorder="little"), ((0, 0), (0, pad))).view(np.uint32)),)
               + p[1:] for p in problems]
    for key, probs in (("in_memory", problems), ("in_memory_bit_rows", as_bits)):
        for w in (1, 2, 4, 8, 16):
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                res = solve_many(probs, w, algo="ncem", beta=0.5, disper="sk_")
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            assert all(r["status"] == 0 for r in res)
            assert np.array_equal(res[0]["c"], solo["c"]) and np.array_equal(res[0]["disp"], solo["disp"])
            out[key][str(w)] = dict(seconds=best, problems_per_s=P / best)
    shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Chunk throughput on one GPU, whole problems (upload included): P independent configs[1]-sized problems through the
drop-in nem() (files on /tmp) on 1..16 worker threads (pangenomenem_amd.batch.nem_many), and in memory through
solve_many -- engines built on `w` host threads, then ONE lock-step batch (nemgpu_run_many).  Prints one JSON object."""
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
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfg = synth.make_config("C2")
    root = tempfile.mkdtemp(prefix="nemchunks_")
    rng = np.random.default_rng(0)
    calls, problems = [], []
    for p in range(P):
        x = np.ascontiguousarray(cfg["x"][:, rng.permutation(cfg["x"].shape[1])])      # another sample of organisms
        base = nemfiles.write_nem_inputs(os.path.join(root, str(p)), x, cfg["nei"], cfg["prop"], cfg["center"], cfg["disp"])
        calls.append(dict(Fname=base.encode(), nk=3, algo=b"ncem", beta=0.5, convergence=b"clas", convergence_th=1e-8,
                          format=b"fuzzy", it_max=100, dolog=True, model_family=b"bern", proportion=b"pk",
                          dispersion=b"sk_", init_mode=2))
        problems.append((x, cfg["nei"], 3, cfg["prop"], cfg["center"], cfg["disp"]))
    nem_many(calls[:2], 1)                                                              # load, context, page cache
    out = dict(problems=P, shape=list(cfg["x"].shape), host_cores=os.cpu_count(), files={}, in_memory={})
    for w in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        rcs = nem_many(calls, w)
        dt = time.perf_counter() - t0
        assert all(rc == 0 for rc in rcs), (w, rcs, open(calls[0]['Fname'].decode() + '.stderr').read()[-400:])
        out["files"][str(w)] = dict(seconds=dt, problems_per_s=P / dt)
    ref_uf = open(calls[0]["Fname"].decode() + ".uf", "rb").read()
    nem_many(calls[:1], 1)
    assert open(calls[0]["Fname"].decode() + ".uf", "rb").read() == ref_uf          # same answer alone and in a crowd
    solve_many(problems[:2], 1, algo="ncem", beta=0.5, disper="sk_")
    for w in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        res = solve_many(problems, w, algo="ncem", beta=0.5, disper="sk_")
        dt = time.perf_counter() - t0
        assert all(r["status"] == 0 for r in res)
        out["in_memory"][str(w)] = dict(seconds=dt, problems_per_s=P / dt)
    shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

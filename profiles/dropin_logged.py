#!/usr/bin/env python3
"""The drop-in nem() as PPanGGOLiN calls it (dolog=True: the full per-iteration log) on configs[1]-sized files whose
solve takes 7 EM iterations (the U-shaped family-frequency data of bench.py), and with NEM_MI355X_LOG=0 (header-only
log, fully pipelined run); the same for init_mode = 1, the 50 random starts.  Prints the whole-call times and the
library's own split."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import nemfiles, synth  # noqa: E402
from pangenomenem_amd.nem import nem  # noqa: E402


def main():
    n, d = 20000, 500
    x, _ = synth.ushaped_pa_matrix(n, d, 1)
    nei = synth.contiguity_graph(n, 1)
    prop, center, disp = synth.default_init(d)
    root = tempfile.mkdtemp()
    base = nemfiles.write_nem_inputs(os.path.join(root, "p"), x, nei, prop, center, disp)
    kw = dict(Fname=base.encode(), nk=3, algo=b"ncem", beta=0.5, convergence=b"clas", convergence_th=1e-8, format=b"fuzzy",
              it_max=100, dolog=True, model_family=b"bern", proportion=b"pk", dispersion=b"sk_", init_mode=2)
    out = {}
    for key, env in (("full_log", None), ("header_only_log", "0")):
        if env is None:
            os.environ.pop("NEM_MI355X_LOG", None)
        else:
            os.environ["NEM_MI355X_LOG"] = env
        nem(**kw)
        best = None
        for rep in range(5):
            t0 = time.perf_counter()
            rc = nem(**kw)
            dt = time.perf_counter() - t0
            assert rc == 0
            best = dt if best is None else min(best, dt)
        out[key] = dict(whole_call_ms=best * 1e3, log_bytes=os.path.getsize(base + ".log"),
                        phases=[l.strip() for l in open(base + ".stderr").read().splitlines() if "[engine]" in l])
    # init_mode = 1 (what partition_shell uses): 50 random starts, with the reference's log of every start (the starts one
    # after the other, one logged step per host round trip) and with NEM_MI355X_LOG=0 (the starts in lock step)
    os.environ["NEM_MI355X_SEED"] = "4242"
    kw["init_mode"] = 1
    for key, env in (("random_starts_full_log", None), ("random_starts_header_only_log", "0")):
        if env is None:
            os.environ.pop("NEM_MI355X_LOG", None)
        else:
            os.environ["NEM_MI355X_LOG"] = env
        nem(**kw)
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            rc = nem(**kw)
            dt = time.perf_counter() - t0
            assert rc == 0
            best = dt if best is None else min(best, dt)
        out[key] = dict(whole_call_ms=best * 1e3, log_bytes=os.path.getsize(base + ".log"),
                        phases=[l.strip() for l in open(base + ".stderr").read().splitlines() if "[engine]" in l])
    os.environ.pop("NEM_MI355X_LOG", None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

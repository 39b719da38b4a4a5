#!/usr/bin/env python3
"""Where a fused sweep launch's time goes: the device's 100 MHz clock at the phase boundaries of k_sweep_fused
(NEM_MI355X_SWEEP_PROF=1), first and last block, for the last sweep of a few EM iterations at configs[1] size.

    NEM_MI355X_SWEEP_PROF=1 NEM_MI355X_GRAPHS=0 python profiles/sweep_phases.py [families organisms]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NEM_MI355X_SWEEP_PROF", "1")
os.environ.setdefault("NEM_MI355X_GRAPHS", "0")

from pangenomenem_amd import engine, synth  # noqa: E402

NAMES = {0: "entry", 1: "prologue done (loads landed, exp table in LDS)"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    x, _ = synth.ushaped_pa_matrix(n, d, 2)
    nei = synth.contiguity_graph(n, 2)
    prop, center, disp = synth.default_init(d)
    eng = engine.NemEngine(n, d, 3)
    eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp)
    eng.configure(algo="ncem", beta=0.5, disper="sk_", cvtest="none", it_max=100)
    lib = engine.load_library()
    lib.nemgpu_sweep_phases.argtypes = [C.POINTER(C.c_ulonglong)]
    out = []
    for it in range(1, iters + 1):
        eng.restart_iterate(it)
        buf = (C.c_ulonglong * 64)()
        assert lib.nemgpu_sweep_phases(buf) == 0
        rec = {}
        for blk, off in (("first_block", 0), ("last_block", 32)):
            t = [int(buf[off + i]) for i in range(32)]
            t0 = t[0]
            rec[blk] = {str(i): round((v - t0) / 100.0, 2) for i, v in enumerate(t) if v}
        out.append(dict(iteration=it, us_since_entry=rec))
    print(json.dumps(dict(families=n, organisms=d, legend="0 entry, 1 prologue done, then per round r: 2+4r local steps done, "
                          "3+4r published, 4+4r met, 5+4r neighbours re-read; 31 end", sweeps=out, counters=eng.sweep_counters()), indent=1))
    eng.close()


if __name__ == "__main__":
    main()

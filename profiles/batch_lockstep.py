"""Throughput of B configs[1]-sized problems: solo runs back to back vs one lock-step batch (nemgpu_run_many).
Times are the library's own loop clocks (EM only: init + iterations), results are not fetched inside the timed part."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, Result

def build(B, n, d, spectrum):
    out = []
    for p in range(B):
        gen = synth.ushaped_pa_matrix if spectrum == "ushape" else synth.bernoulli_pa_matrix
        x, _ = gen(n, d, 100 + p)
        nei = synth.contiguity_graph(n, 100 + p)
        prop, center, disp = synth.default_init(d)
        e = NemEngine(n, d, 3)
        e.set_matrix(x); e.set_graph(nei); e.set_params(prop, center, disp)
        e.configure(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=1)
        out.append(e)
    return out

def run_many_raw(engs):
    lib = engs[0].lib
    handles = (C.c_void_p * len(engs))(*[e._h for e in engs])
    res = (Result * len(engs))()
    t0 = time.perf_counter()
    rc = lib.nemgpu_run_many(handles, len(engs), res)
    dt = time.perf_counter() - t0
    assert rc == 0
    return dt, res

def run_solo_raw(engs):
    r = Result()
    t0 = time.perf_counter()
    for e in engs:
        assert e.lib.nemgpu_run(e._h, C.byref(r)) == 0
    return time.perf_counter() - t0

def main():
    n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 500
    sizes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 16, 32, 64]
    rec = dict(n=n, d=d, rows=[])
    for spectrum in ("ushape", "latent3"):
        for B in sizes:
            engs = build(B, n, d, spectrum)
            run_solo_raw(engs); run_solo_raw(engs)            # warm (graphs captured on the second pass)
            t_solo = min(run_solo_raw(engs) for _ in range(3))
            run_many_raw(engs)
            best = None
            for _ in range(3):
                dt, res = run_many_raw(engs)
                best = dt if best is None else min(best, dt)
            iters = sum(r.iters for r in res)
            rec["rows"].append(dict(spectrum=spectrum, B=B, solo_s=t_solo, lockstep_s=best, speedup=t_solo / best,
                                    problems_per_s=B / best, em_iterations=iters, us_per_problem_iteration=best * 1e6 / max(iters, 1),
                                    cells_per_s=iters * n * d / best))
            print(rec["rows"][-1], file=sys.stderr)
            for e in engs: e.close()
    print(json.dumps(rec))
main()

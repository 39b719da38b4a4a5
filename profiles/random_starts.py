"""RandNemAlgo's 50 starts (init_mode = INIT_RANDOM): one after the other vs in lock step."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, Result
rec = []
for (n, d, tie, gen) in [(5000, 500, "hash", "ushape"), (20000, 500, "hash", "ushape"), (20000, 500, "libc", "ushape"),
                         (20000, 500, "libc", "latent3")]:
    # (ushape under libc: every start's sweeps draw tie-breaks -- the data hold identical rows; latent3: hardly any do)
    x, _ = (synth.ushaped_pa_matrix if gen == "ushape" else synth.bernoulli_pa_matrix)(n, d, 9)
    nei = synth.contiguity_graph(n, 9)
    row = dict(n=n, d=d, tie=tie, data=gen)
    for mode in ("0", "1"):
        os.environ["NEM_MI355X_BATCH_STARTS"] = mode
        eng = NemEngine(n, d, 3)
        eng.set_matrix(x); eng.set_graph(nei)
        eng.configure(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=100, tie=tie, seed=3)
        r, best = Result(), C.c_int(-1)
        def go(starts):
            t0 = time.perf_counter()
            assert eng.lib.nemgpu_run_random(eng._h, starts, C.c_uint32(3), C.byref(r), C.byref(best)) == 0
            return time.perf_counter() - t0
        go(50); go(1)
        one = min(go(1) for _ in range(3))
        fifty = min(go(50) for _ in range(2))                  # (lock step: plain launches, as in a call of its own)
        row["one_start_s" if mode == "0" else "one_start_s_lockstep_build"] = one
        row["fifty_sequential_s" if mode == "0" else "fifty_lockstep_s"] = fifty
        if mode == "1":                                         # the batch shapes' graphs exist from the fourth run on
            go(50); go(50)
            row["fifty_lockstep_replayed_s"] = min(go(50) for _ in range(3))
        row["best_%s" % mode] = best.value
        row["tie_draws_%s" % mode] = int(r.tie_draws)
        if mode == "1": row["how"] = eng.random_start_counters()   # lock-step rounds, starts in them, starts alone, starts redone
        eng.close()
    row["lockstep_over_one_start"] = row["fifty_lockstep_s"] / row["one_start_s"]
    row["speedup"] = row["fifty_sequential_s"] / row["fifty_lockstep_s"]
    print(row, file=sys.stderr)
    rec.append(row)
print(json.dumps(rec))

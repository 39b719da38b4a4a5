"""One lock-step run of RandNemAlgo's 50 starts with the batch driver's phase timings (NEM_MI355X_BATCH_PROF=1)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NEM_MI355X_BATCH_PROF"] = "1"
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, Result
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 500
tie = sys.argv[2] if len(sys.argv) > 2 else "hash"
gen = sys.argv[3] if len(sys.argv) > 3 else "ushape"
x, _ = (synth.ushaped_pa_matrix if gen == "ushape" else synth.bernoulli_pa_matrix)(n, d, 9)
nei = synth.contiguity_graph(n, 9)
eng = NemEngine(n, d, 3)
eng.set_matrix(x); eng.set_graph(nei)
eng.configure(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=100, tie=tie, seed=3)
r, best = Result(), C.c_int(-1)
for rep in range(3):
    print("---- run", rep, file=sys.stderr)
    t0 = time.perf_counter()
    assert eng.lib.nemgpu_run_random(eng._h, 50, C.c_uint32(3), C.byref(r), C.byref(best)) == 0
    print("total %.0f us, iterations of the best start %d, %s" % ((time.perf_counter() - t0) * 1e6, r.iters, eng.random_start_counters()), file=sys.stderr)

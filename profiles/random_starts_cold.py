"""RandNemAlgo's 50 starts on a FRESH engine per call (what a drop-in nem() call with init_mode = 1 does): first call of the
process, then three more engines; lock step against one after the other, with the batch driver's phase timings."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, Result
n, d = 20000, 500
gen = sys.argv[1] if len(sys.argv) > 1 else "ushape"
x, _ = (synth.ushaped_pa_matrix if gen == "ushape" else synth.bernoulli_pa_matrix)(n, d, 9)
nei = synth.contiguity_graph(n, 9)
for mode in ("1", "0", "1", "0", "1", "0"):
    os.environ["NEM_MI355X_BATCH_STARTS"] = mode
    t0 = time.perf_counter()
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x); eng.set_graph(nei)
    eng.configure(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=100, tie="libc", seed=3)
    t1 = time.perf_counter()
    r, best = Result(), C.c_int(-1)
    assert eng.lib.nemgpu_run_random(eng._h, 50, C.c_uint32(3), C.byref(r), C.byref(best)) == 0
    t2 = time.perf_counter()
    eng.close()
    t3 = time.perf_counter()
    print("%s: engine + upload %.2f ms, 50 starts %.2f ms, close %.2f ms" % ("lock step" if mode == "1" else "sequential", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), file=sys.stderr)

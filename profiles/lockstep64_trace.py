"""64 configs[1]-sized problems in lock step (nemgpu_run_many), three calls: the workload of bench.py's
also.lockstep_64_configs1, alone in a process for a kernel trace."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, Result
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x0, _ = synth.ushaped_pa_matrix(20000, 500, 100)
nei0 = synth.contiguity_graph(20000, 100)
prop, center, disp = synth.default_init(500)
rng = np.random.default_rng(0)
engs = []
for p in range(B):
    e = NemEngine(20000, 500, 3)
    e.set_matrix(np.ascontiguousarray(x0[:, rng.permutation(500)])); e.set_graph(nei0); e.set_params(prop, center, disp)
    e.configure(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=1)
    engs.append(e)
lib = engs[0].lib
handles = (C.c_void_p * B)(*[e._h for e in engs])
res = (Result * B)()
for rep in range(3):
    t0 = time.perf_counter()
    assert lib.nemgpu_run_many(handles, B, res) == 0
    dt = time.perf_counter() - t0
    it = sum(r.iters for r in res)
    print("call %d: %.3f ms, %d EM iterations, %.2f us per problem-iteration" % (rep, dt * 1e3, it, dt * 1e6 / it), file=sys.stderr)

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine
n,d=20000,500
x,_=synth.ushaped_pa_matrix(n,d,2); nei=synth.contiguity_graph(n,2); p,c,dd=synth.default_init(d)
eng=NemEngine(n,d,3); eng.set_matrix(x); eng.set_graph(nei); eng.set_params(p,c,dd)
eng.configure(algo="ncem",beta=0.5,disper="sk_",propor="pk",cvtest="clas",cvthres=1e-8,it_max=100)
r=eng.run(); print("run", r["iters"], r["converged"], r["sweep_rounds"], eng.graph_counters())
eng.configure(algo="ncem",beta=0.5,disper="sk_",propor="pk",cvtest="none",it_max=100)
for m in (0,1,2,3,7,7,7):
    r=eng.restart_iterate(m); print(m, r["iters"], r["sweep_rounds"], eng.graph_counters())
lab=eng.labels(); print(np.bincount(lab))

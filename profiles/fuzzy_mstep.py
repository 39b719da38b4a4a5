"""The fuzzy M-step alone (EstimPara on a float partition, nem_mod.c:415-469) on configs[1]: microseconds per call for a
one-hot, a real (three fuzzy EM iterations), a random and a constant partition (launches and the wait included).
NEM_MI355X_FUZZY_CHAINS=0 times the one-lane-per-chain kernels of round 1, =1 the wave-per-chain kernels, default (2):
the producer / consumer kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine
n,d=20000,500
x,_=synth.ushaped_pa_matrix(n,d,2); nei=synth.contiguity_graph(n,2); p,c,dd=synth.default_init(d)
eng=NemEngine(n,d,3); eng.set_matrix(x); eng.set_graph(nei); eng.set_params(p,c,dd)
eng.configure(algo="nem",beta=0.5,disper="sk_",propor="pk",cvtest="none",it_max=3)
r=eng.run()
real=r["c"].copy()
rng=np.random.default_rng(1)
lab=rng.integers(0,3,n); onehot=np.zeros((n,3),np.float32); onehot[np.arange(n),lab]=1
rnd=rng.random((n,3)).astype(np.float32); rnd/=rnd.sum(1,keepdims=True)
half=np.full((n,3),0.25,np.float32); half[:,0]=0.5
for name,cc in (("onehot",onehot),("real",real),("random",rnd),("const",half)):
    eng.set_partition(cc.astype(np.float32))
    eng.mstep()
    t0=time.perf_counter()
    for _ in range(5): eng.mstep()
    print(name, (time.perf_counter()-t0)/5*1e6, "us per M-step")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from pangenomenem_amd import distributed as nd
job = nd.ShardedNem.synthetic(20000, 500, 3, 0.5, 0, 1, 0)
orig = job.st.end
log = []
def end():
    r = orig(); log.append(dict(r)); return r
job.st.end = end
cyc = job.iters_to_converge()
print("cycle", cyc, "native", job.native)
log.clear()
for rep in range(3):
    job.run_steps(cyc, cyc)
    print("cycle run", rep, [(r["iters"], r["commits"], r["need_rounds"], r["status"], r["sweep_rounds"]) for r in log]); log.clear()
print(job.eng.graph_counters())
dist.destroy_process_group()

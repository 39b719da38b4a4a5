"""The sharded driver with the real GPU stepper on the one-GPU box: (a) world_size 1 over nccl (= RCCL)
exercises the exact code path bench.py --gpus N uses; (b) world_size 2 over gloo, both ranks on GPU 0
with host-staged collectives, checks shards + kernels + protocol against the oracle."""
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, backend, initfile, n, d, beta, outdir):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from pangenomenem_amd import synth
    from pangenomenem_amd.distributed import Comm, GpuStepper, ShardedNem, shard_bounds, slice_graph, slot_layout
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="file://" + initfile, rank=rank, world_size=world, **kw)
    try:
        x, _ = synth.bernoulli_pa_matrix(n, d, 1)
        nei = synth.contiguity_graph(n, 1)
        prop, center, disp = synth.default_init(d)
        lo, hi, _ = shard_bounds(n, world, rank)
        blk, stride = slot_layout(n, world, 3 + 3 * d)
        cfg = dict(algo="ncem", beta=beta, disper="sk_", propor="pk", cvtest="clas", seed=11)
        st = GpuStepper(x[lo:hi], slice_graph(nei, lo, hi, blk, stride), 3, n, world, rank, prop, center, disp, 0, cfg)
        job = ShardedNem(st, Comm(), n, beta, cvtest="clas", cvthres=1e-8)
        res = job.run(100)
        labels = job.global_labels().copy()
        params = {k: v.copy() for k, v in st.params().items()}
        # the bench helpers too
        cyc = job.iters_to_converge()
        job.run_steps(7, cyc)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), labels=labels, iters=res["iters"],
                 converged=res["converged"], status=res["status"], cycle=cyc, **params)
    finally:
        dist.destroy_process_group()


def _run(world, backend, n, d, beta):
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemgdist_")
    mp.spawn(_worker, args=(world, backend, os.path.join(outdir, "rdv"), n, d, beta, outdir), nprocs=world, join=True)
    return [np.load(os.path.join(outdir, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,backend,beta", [(1, "nccl", 0.5), (2, "gloo", 0.5), (1, "nccl", 0.0), (2, "gloo", 0.0)])
def test_gpu_sharded_matches_oracle(gpu_lib, oracle, world, backend, beta):
    from pangenomenem_amd import synth
    n, d = 4096, 15
    outs = _run(world, backend, n, d, beta)
    x, _ = synth.bernoulli_pa_matrix(n, d, 1)
    prop, center, disp = synth.default_init(d)
    want = oracle.run(x, synth.contiguity_graph(n, 1), 3, prop, center, disp, algo="ncem", beta=beta, tie="hash", seed=11)
    for o in outs:
        assert int(o["status"]) == want["status"] and int(o["iters"]) == want["iters"]
        assert bool(o["converged"]) == want["converged"]
        assert int(o["cycle"]) == max(5, want["iters"])
        assert np.array_equal(o["labels"], want["c"].argmax(1))
        assert np.array_equal(o["center"], want["center"])
        assert np.array_equal(o["disp"], want["disp"]) and np.array_equal(o["prop"], want["prop"])

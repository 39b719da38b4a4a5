"""Fuzzy NEM (algo = "nem") over several ranks with the real GPU stepper (pangenomenem_amd/distributed.py:
ShardedFuzzyNem / FuzzyGpuStepper, the nemgpu_shard_fuzzy_* C ABI): the E-step sharded over families, the M-step's
i-ordered chains over organisms.  World 1 over nccl (= RCCL) and world 2 / 3 over gloo with every rank on the one GPU
of the box (host-staged collectives).  The sharded run must equal the single engine bit for bit -- same kernels, same
chains, the same order of every sum -- and the oracle within the tolerance of north_star."""
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-6


def _worker(rank, world, backend, initfile, n, d, beta, disper, it_max, outdir):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from pangenomenem_amd import synth
    from pangenomenem_amd.distributed import Comm, ShardedFuzzyNem
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="file://" + initfile, rank=rank, world_size=world, **kw)
    try:
        x, _ = synth.ushaped_pa_matrix(n, d, 4)
        nei = synth.contiguity_graph(n, 4)
        prop, center, disp = synth.default_init(d)
        job = ShardedFuzzyNem.from_problem(x, nei, 3, prop, center, disp, beta, rank, world, 0, disper=disper)
        res = job.run(it_max)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), c=job.memberships(), iters=res["iters"], converged=res["converged"],
                 status=res["status"], emptyk=res["emptyk"], rounds=res["sweep_rounds"], **job.params())
    finally:
        dist.destroy_process_group()


def _run(world, backend, n, d, beta, disper, it_max):
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemgfz_")
    mp.spawn(_worker, args=(world, backend, os.path.join(outdir, "rdv"), n, d, beta, disper, it_max, outdir), nprocs=world, join=True)
    return [np.load(os.path.join(outdir, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,backend,n,d,beta,disper,it_max", [(1, "nccl", 3000, 40, 0.5, "sk_", 10), (2, "gloo", 3000, 40, 0.5, "sk_", 10),
                                                                  (3, "gloo", 2500, 70, 1.0, "skd", 10), (2, "gloo", 1501, 33, 0.0, "sk_", 10),
                                                                  (2, "gloo", 1000, 20, 0.5, "sk_", 0), (2, "gloo", 20000, 500, 0.5, "sk_", 3)])
def test_gpu_sharded_fuzzy_equals_the_single_engine_and_the_oracle(gpu_lib, oracle, world, backend, n, d, beta, disper, it_max):
    from pangenomenem_amd import synth
    from pangenomenem_amd.engine import solve
    from tests.util import maxdiff
    outs = _run(world, backend, n, d, beta, disper, it_max)
    x, _ = synth.ushaped_pa_matrix(n, d, 4)
    nei = synth.contiguity_graph(n, 4)
    prop, center, disp = synth.default_init(d)
    one = solve(x, nei, 3, prop, center, disp, algo="nem", beta=beta, disper=disper, it_max=it_max)
    want = oracle.run(x, nei, 3, prop, center, disp, algo="nem", beta=beta, disper=disper, it_max=it_max)
    for o in outs:
        assert int(o["status"]) == one["status"] == want["status"] and int(o["iters"]) == one["iters"] == want["iters"]
        assert bool(o["converged"]) == bool(one["converged"])
        # the same kernels summing the same numbers in the same order: bit for bit
        assert np.array_equal(o["c"].view(np.uint32), one["c"].view(np.uint32))
        assert np.array_equal(o["center"], one["center"])
        assert np.array_equal(o["disp"].view(np.uint32), one["disp"].view(np.uint32))
        assert np.array_equal(o["prop"].view(np.uint32), one["prop"].view(np.uint32))
        assert np.array_equal(o["nbobs_k"].view(np.uint32), one["nbobs_k"].view(np.uint32))
        # ... and the reference's arithmetic within the tolerance of north_star
        assert np.array_equal(o["c"].argmax(1), want["c"].argmax(1))
        assert maxdiff(o["c"], want["c"]) <= TOL and np.array_equal(o["center"], want["center"])
        assert maxdiff(o["disp"], want["disp"]) <= TOL and maxdiff(o["prop"], want["prop"]) <= TOL


def test_sharded_fuzzy_rejects_an_ncem_engine(gpu_lib):
    import ctypes as C
    from pangenomenem_amd import synth
    from pangenomenem_amd.engine import NemEngine
    x, _ = synth.bernoulli_pa_matrix(256, 10, 1)
    e = NemEngine(256, 10, 3)
    e.set_matrix(x)
    e.set_graph(None)
    e.set_params(*synth.default_init(10))
    e.configure(algo="ncem")
    assert e.lib.nemgpu_shard_fuzzy_layout(e._h, 256) != 0
    assert b"algo = nem" in e.lib.nemgpu_last_error()

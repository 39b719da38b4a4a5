"""The reference's tie stream (NEMGPU_TIE_LIBC: kmaxes[random() % (nequal + 1)], nem_alg.c:617-637 -> nem_rnd.c:53-61) in
the family-sharded path: a tied site draws random() number `draws before the sweep + sites below it that drew in this
sweep`, the sites of the ranks below come first, every rank's draws of a round ride in its block's tail with the
all-gathered labels.  The two golden cases of the UNMODIFIED reference that tie, and a matrix all of whose densities
underflow (every site of every sweep draws), through ShardedNem at world 1 (RCCL), 2 and 3 (gloo, the ranks sharing the
one GPU of the box): labels equal the reference's fixtures / the oracle's, iteration for iteration."""
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, backend, initfile, spec, outdir):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from pangenomenem_amd import synth
    from pangenomenem_amd.distributed import Comm, GpuStepper, ShardedNem, shard_bounds, slice_graph, slot_layout
    from tests.golden_util import load_case
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="file://" + initfile, rank=rank, world_size=world, **kw)
    try:
        if spec["kind"] == "golden":
            case = load_case(spec["name"])
            x, nei, k = case["x"], case["nei"], case["k"]
            prop, center, disp = case["prop"], case["center"], case["disp"]
            c = case["cfg"]
            cfg = dict(algo="ncem", beta=c["beta"], disper=c["disper"], propor=c["propor"], cvtest=c["cvtest"], cvthres=c["cvthres"],
                       tie="libc", seed=case["meta"]["libc_seed"])
            beta, cvtest, cvthres, it_max = c["beta"], c["cvtest"], c["cvthres"], c["it_max"]
        else:
            n, d = spec["n"], spec["d"]
            x, _ = synth.bernoulli_pa_matrix(n, d, 5)
            nei = synth.contiguity_graph(n, 5)
            prop, center, disp = synth.default_init(d)
            k, beta, cvtest, cvthres, it_max = 3, spec["beta"], "clas", 1e-8, spec["it_max"]
            cfg = dict(algo="ncem", beta=beta, disper="sk_", propor="pk", cvtest="clas", tie="libc", seed=3)
        n, d = x.shape
        if nei is None:
            nei = (np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        lo, hi, _ = shard_bounds(n, world, rank)
        blk, stride = slot_layout(n, world, k + k * d)
        st = GpuStepper(x[lo:hi], slice_graph(nei, lo, hi, blk, stride), k, n, world, rank, prop, center, disp, 0, cfg)
        job = ShardedNem(st, Comm(), n, beta, cvtest=cvtest, cvthres=cvthres)
        assert job.libc
        res = job.run(it_max)
        labels = job.global_labels().copy()
        params = {kk: v.copy() for kk, v in st.params().items()}
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), labels=labels, iters=res["iters"], converged=res["converged"],
                 status=res["status"], **params)
    finally:
        dist.destroy_process_group()


def _run(world, backend, spec):
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemglibc_")
    mp.spawn(_worker, args=(world, backend, os.path.join(outdir, "rdv"), spec, outdir), nprocs=world, join=True)
    return [np.load(os.path.join(outdir, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo"), (3, "gloo")])
@pytest.mark.parametrize("name", ["ties_two_equal_classes", "k10_ncem_skd"])
def test_sharded_tie_cases_equal_the_reference_fixtures(gpu_lib, name, world, backend):
    from tests.golden_util import load_case
    exp = load_case(name)["expected"]
    for o in _run(world, backend, dict(kind="golden", name=name)):
        assert int(o["status"]) == int(exp["status"]) and int(o["iters"]) == int(exp["iters"])
        assert bool(o["converged"]) == bool(exp["converged"])
        assert np.array_equal(o["labels"], exp["c"].argmax(1))
        assert np.array_equal(o["center"], exp["center"])
        assert float(np.max(np.abs(o["disp"] - exp["disp"]))) <= 1e-6 and float(np.max(np.abs(o["prop"] - exp["prop"]))) <= 1e-6


@pytest.mark.parametrize("world,backend,beta", [(2, "gloo", 0.5), (3, "gloo", 0.0)])
def test_sharded_every_density_underflows_with_the_reference_tie_stream(gpu_lib, oracle, world, backend, beta):
    """every site of every sweep draws: the draw tables grow, the ranks' totals are thousands, sweeps take many rounds"""
    from pangenomenem_amd import synth
    n, d, it_max = 6000, 2800, 2
    x, _ = synth.bernoulli_pa_matrix(n, d, 5)
    prop, center, disp = synth.default_init(d)
    want = oracle.run(x, synth.contiguity_graph(n, 5), 3, prop, center, disp, algo="ncem", beta=beta, disper="sk_", propor="pk",
                      it_max=it_max, tie="libc", seed=3)
    assert want["n_zero_density"] > n // 2
    for o in _run(world, backend, dict(kind="underflow", n=n, d=d, beta=beta, it_max=it_max)):
        assert int(o["iters"]) == want["iters"] and int(o["status"]) == want["status"]
        assert np.array_equal(o["labels"], want["c"].argmax(1))
        assert np.array_equal(o["center"], want["center"])

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


def _worker(rank, world, backend, initfile, n, d, beta, outdir, gen="latent3", seed=1, it_max=100, bench_helpers=True):
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
        lo, hi, _ = shard_bounds(n, world, rank)
        if gen == "ushape":
            x_local, _ = synth.ushaped_pa_matrix(n, d, seed, rows=(lo, hi))     # (this rank's rows of the same matrix)
        else:
            x_local = synth.bernoulli_pa_matrix(n, d, seed)[0][lo:hi]
        nei = synth.contiguity_graph(n, seed)
        prop, center, disp = synth.default_init(d)
        blk, stride = slot_layout(n, world, 3 + 3 * d)
        cfg = dict(algo="ncem", beta=beta, disper="sk_", propor="pk", cvtest="clas", seed=11)
        st = GpuStepper(x_local, slice_graph(nei, lo, hi, blk, stride), 3, n, world, rank, prop, center, disp, 0, cfg)
        job = ShardedNem(st, Comm(), n, beta, cvtest="clas", cvthres=1e-8)
        res = job.run(it_max)
        labels = job.global_labels().copy()
        params = {k: v.copy() for k, v in st.params().items()}
        cyc = 0
        if bench_helpers:                             # the bench helpers too
            cyc = job.iters_to_converge()
            job.run_steps(7, cyc)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), labels=labels, iters=res["iters"],
                 converged=res["converged"], status=res["status"], cycle=cyc, **params)
    finally:
        dist.destroy_process_group()


def _run(world, backend, n, d, beta, **kw):
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemgdist_")
    extra = (kw.get("gen", "latent3"), kw.get("seed", 1), kw.get("it_max", 100), kw.get("bench_helpers", True))
    mp.spawn(_worker, args=(world, backend, os.path.join(outdir, "rdv"), n, d, beta, outdir) + extra, nprocs=world, join=True)
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
        assert int(o["cycle"]) == max(1, want["iters"])
        assert np.array_equal(o["labels"], want["c"].argmax(1))
        assert np.array_equal(o["center"], want["center"])
        assert np.array_equal(o["disp"], want["disp"]) and np.array_equal(o["prop"], want["prop"])


def test_sharded_driver_at_configs2_size_matches_the_oracle(gpu_lib, oracle):
    """BASELINE configs[2] in its own form: 50 000 families x 1 000 organisms, K=3, beta=0.5, families sharded over two
    ranks (gloo, both on the one GPU of the box) -- the workload bench.py shards -- against the CPU oracle's whole run:
    labels bit-exact, centres, dispersions and proportions equal, same iteration count."""
    from pangenomenem_amd import synth
    n, d = 50000, 1000
    outs = _run(2, "gloo", n, d, 0.5, gen="ushape", seed=3, bench_helpers=False)
    x, _ = synth.ushaped_pa_matrix(n, d, 3)
    prop, center, disp = synth.default_init(d)
    want = oracle.run(x, synth.contiguity_graph(n, 3), 3, prop, center, disp, algo="ncem", beta=0.5, tie="hash", seed=11)
    assert want["iters"] >= 5
    for o in outs:
        assert int(o["status"]) == want["status"] and int(o["iters"]) == want["iters"] and bool(o["converged"]) == want["converged"]
        assert np.array_equal(o["labels"], want["c"].argmax(1))
        assert np.array_equal(o["center"], want["center"])
        assert np.array_equal(o["disp"], want["disp"]) and np.array_equal(o["prop"], want["prop"])


def test_sharded_driver_at_configs3_size_equals_the_single_engine(gpu_lib):
    """BASELINE configs[3]'s 200 000 x 5 000 matrix sharded over two ranks: every density underflows at this width (as
    in the reference), every site is an exact tie, and the labels are whatever the tie rule draws -- the
    sharded run's labels and parameters equal the single engine's on the same matrix, iteration for iteration."""
    from pangenomenem_amd import synth
    from pangenomenem_amd.engine import solve
    n, d, iters = 200000, 5000, 3
    outs = _run(2, "gloo", n, d, 0.5, gen="ushape", seed=2, it_max=iters, bench_helpers=False)
    x, _ = synth.ushaped_pa_matrix(n, d, 2)
    prop, center, disp = synth.default_init(d)
    one = solve(x, synth.contiguity_graph(n, 2), 3, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=11,
                it_max=iters)
    del x
    assert one["n_zero_density"] > 0                      # (the stress regime: zero densities everywhere)
    for o in outs:
        assert int(o["iters"]) == one["iters"] == iters and int(o["status"]) == one["status"]
        assert np.array_equal(o["labels"], one["c"].argmax(1))
        assert np.array_equal(o["center"], one["center"]) and np.array_equal(o["disp"], one["disp"])
        assert np.array_equal(o["prop"], one["prop"])


def _bench(*argv):
    """python bench.py ... as the driver starts it: one JSON line on stdout"""
    import json
    import subprocess
    env = dict(os.environ)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, check=True,
                         stdout=subprocess.PIPE, timeout=900).stdout.decode()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(gpu_lib):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks itself (both on the one
    GPU of the box, host-staged gloo collectives) and prints the one JSON line; the default multi-GPU workload is
    the strong-scaling split of ONE problem."""
    rec = _bench("--gpus", "2", "--backend", "gloo", "--steps", "12", "--warmup", "3", "--families", "6000",
                 "--organisms", "200", "--repeats", "3", "--extras-strong-shape", "9000x300")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["steps"] == 12 and rec["warmup"] == 3
    assert rec["config"]["families_total"] == 6000 and rec["config"]["families_per_gpu"] == 3000
    assert rec["value"] > 0 and rec["iters_to_converge"] >= 1 and rec["repeats"] == 3
    assert rec["ms_per_step_min"] <= rec["ms_per_step"] <= rec["ms_per_step_max"]
    # an N-rank line says what its collectives cost, what one GPU does with the same problem, and what else N GPUs can
    # do with this path: independent problems (no collective) and the strong scaling of a larger problem
    assert rec["collective"]["per_iteration"] == 2 and rec["collective"]["per_iteration_ms"] > 0
    assert rec["single_gpu_same_workload"]["ms_per_step"] > 0 and rec["speedup_vs_single_gpu_same_workload"] > 0
    also = rec["also"]
    assert also["replicas_20000x500_per_gpu"]["value"] > 0 and also["replicas_20000x500_per_gpu"]["collectives_per_iteration"] == 0
    big = also["strong_9000x300"]
    assert big["value"] > 0 and big["single_gpu_same_workload"]["ms_per_step"] > 0 and big["collective"]["per_iteration"] == 2


@pytest.mark.parametrize("scaling", ["replicas", "weak"])
def test_bench_other_multi_gpu_modes(gpu_lib, scaling):
    """--scaling replicas: N independent problems, one per rank, no collective on the data path (the reference's own
    parallel form); --scaling weak: one problem of N x --families families, sharded.  Two ranks on the one GPU, gloo."""
    rec = _bench("--gpus", "2", "--backend", "gloo", "--scaling", scaling, "--steps", "14", "--warmup", "7",
                 "--families", "5000", "--organisms", "150", "--repeats", "3", "--no-extras")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["families_total"] == 10000 and rec["config"]["families_per_gpu"] == 5000
    if scaling == "replicas":
        assert rec["collective"]["per_iteration"] == 0 and rec["whole_solves_per_block"] > 0 and rec["graphs_primed"]
        assert "no collective" in rec["config"]["parallelism"]
    else:
        assert rec["collective"]["per_iteration"] == 2 and "sharded" in rec["config"]["parallelism"]


def test_bench_sharded_driver_agrees_with_single_engine(gpu_lib):
    """N = 1 through the sharded driver (--dist, RCCL group of one rank) against the single engine on the same
    workload: within 1.3x of each other at 50 000 x 1 000; and a short run (--steps 20 --warmup 5) reports the same ms_per_step as a
    long one -- every batch shape is captured before the clock starts."""
    common = ["--families", "20000", "--organisms", "500", "--no-cpu-baseline", "--no-north-star"]
    short = _bench("--steps", "20", "--warmup", "5", *common)
    long_ = _bench("--steps", "700", "--warmup", "70", *common)
    assert short["graphs_primed"] and long_["graphs_primed"]
    assert short["iters_to_converge"] == long_["iters_to_converge"] >= 5
    assert short["ms_per_step"] < 1.6 * long_["ms_per_step"], (short["ms_per_step"], long_["ms_per_step"])
    # the ratio at BASELINE configs[2]'s size (the shape the sharded path is for)
    common = ["--families", "50000", "--organisms", "1000", "--no-cpu-baseline", "--no-north-star", "--repeats", "7"]
    long_ = _bench("--steps", "220", "--warmup", "22", *common)
    dist1 = _bench("--dist", "--steps", "220", "--warmup", "22", *common)
    assert long_["iters_to_converge"] == dist1["iters_to_converge"] >= 5 and dist1["batch_graphs"]
    # (a rank alone skips the all-gathers; the sharded iteration keeps one launch more than the single engine's, its
    #  loop control: round 3 brought the ratio from 1.5-1.6 to under 1.2 at 50 000 x 1 000)
    assert 0.5 < dist1["ms_per_step"] / long_["ms_per_step"] < 1.3, (dist1["ms_per_step"], long_["ms_per_step"])


def _replay_worker(rank, world, initfile, outdir):
    """a 1-rank sharded job whose restart batch has been captured, then set_params (a lazy reset), then the
    batch REPLAYED: the host half of the restart must run on a replay too, or the next look at the parameters copies
    the initial ones over the estimated ones."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from pangenomenem_amd import synth
    from pangenomenem_amd.distributed import Comm, GpuStepper, ShardedNem, slice_graph, slot_layout
    from pangenomenem_amd.engine import solve
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="file://" + initfile, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        n, d = 6000, 40
        x, _ = synth.ushaped_pa_matrix(n, d, 9)
        nei = synth.contiguity_graph(n, 9)
        prop, center, disp = synth.default_init(d)
        blk, stride = slot_layout(n, 1, 3 + 3 * d)
        cfg = dict(algo="ncem", beta=0.5, disper="sk_", propor="pk", cvtest="clas", seed=11)
        st = GpuStepper(x, slice_graph(nei, 0, n, blk, stride), 3, n, 1, 0, prop, center, disp, 0, cfg)
        job = ShardedNem(st, Comm(), n, 0.5, cvtest="clas", cvthres=1e-8)
        runs = []
        for _ in range(3):                                   # plain, captured, replayed
            res = job.run(100)
            runs.append((res["iters"], job.global_labels().copy(), {k: v.copy() for k, v in st.params().items()}))
        counters = st.eng.graph_counters()
        # other initial parameters: a lazy reset is pending when the (replayed) restart batch starts
        disp2 = disp.copy(); disp2[0] = 0.2; disp2[2] = 0.05
        st.eng.set_params(prop, center, disp2)
        res2 = job.run(100)
        got = (res2["iters"], job.global_labels().copy(), {k: v.copy() for k, v in st.params().items()})
        want = solve(x, nei, 3, prop, center, disp2, **dict(cfg, tie="hash"))
        first = solve(x, nei, 3, prop, center, disp, **dict(cfg, tie="hash"))
        np.savez(os.path.join(outdir, "replay.npz"), replayed=counters["replayed"], captured=counters["captured"],
                 iters=[r[0] for r in runs], same_labels=all(np.array_equal(r[1], runs[0][1]) for r in runs),
                 same_disp=all(np.array_equal(r[2]["disp"], runs[0][2]["disp"]) for r in runs),
                 first_ok=bool(first["iters"] == runs[0][0] and np.array_equal(first["c"].argmax(1), runs[0][1]) and np.array_equal(first["disp"], runs[0][2]["disp"])),
                 iters2=got[0], want_iters2=want["iters"], labels2_ok=bool(np.array_equal(got[1], want["c"].argmax(1))),
                 disp2_ok=bool(np.array_equal(got[2]["disp"], want["disp"])), prop2_ok=bool(np.array_equal(got[2]["prop"], want["prop"])),
                 changed=bool(not np.array_equal(want["disp"], first["disp"])))
    finally:
        dist.destroy_process_group()


def test_replayed_restart_batch_runs_its_host_half(gpu_lib):
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemgreplay_")
    mp.spawn(_replay_worker, args=(1, os.path.join(outdir, "rdv"), outdir), nprocs=1, join=True)
    o = np.load(os.path.join(outdir, "replay.npz"))
    assert int(o["replayed"]) > 0 and int(o["captured"]) > 0            # (the third run did replay a captured batch)
    assert bool(o["same_labels"]) and bool(o["same_disp"]) and bool(o["first_ok"])
    assert bool(o["changed"])                                            # (the second parameter set gives another answer)
    assert int(o["iters2"]) == int(o["want_iters2"]) and bool(o["labels2_ok"]) and bool(o["disp2_ok"]) and bool(o["prop2_ok"])

"""The edge weights the reference's own caller writes (ppanggolin.py:866-878: `distance_score = coverage`, the number of
selected organisms that carry the adjacency -- up to D, not the 1..8 of SURVEY.md §8d's synthetic graphs).  With them
beta * sum(w) passes 88, where criterion Z's float exp overflows (nem_alg.c:2740-2751: M = -inf, labels untouched), and
709, where the site's own double exp does (nem_alg.c:2581-2601: inf * (1 / inf) = NaN rows; ComputeMAP's NaN rules
nem_alg.c:603-637 under NCEM; NaN class sizes -> "Class k empty" -> status 2 -> nem() returns 1 and writes nothing under
fuzzy NEM).  The fixtures (tests/golden/cov*, nan_ring*, straddle709*, mid88_709*, edge88*) come from the UNMODIFIED
reference; here they go through a lock-step batch and the family-sharded driver (the solo engine and the drop-in take
them in tests/test_gpu_golden.py, which iterates over the manifest), and a differential fuzz with two more weight
kinds reaches the same branches on random problems."""
import os

import numpy as np
import pytest

from tests.golden_util import load_case
from tests.util import assert_crit_close, maxdiff
from pangenomenem_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-6

HEAVY = ["cov60_ncem_sk", "cov60_ncem_skd", "cov60_nem_sk", "cov60_nem_skd", "cov500_ncem_sk", "cov500_ncem_skd",
         "cov500_nem_sk", "cov500_w1500_ncem", "cov500_w1500_ncem_beta1", "nan_ring_ncem", "nan_ring_nem",
         "straddle709_ncem", "straddle709_nem", "straddle709_ncem_skd", "mid88_709_nem", "mid88_709_ncem", "edge88_nem",
         "edge88_ncem"]


def check_against_fixture(got, case, name):
    cfg, exp = case["cfg"], case["expected"]
    assert got["status"] == int(exp["status"]), name
    assert got["iters"] == int(exp["iters"]), name
    if got["status"] == 2:
        assert got["emptyk"] > 0, name
        return
    assert got["converged"] == bool(exp["converged"]), name
    assert np.array_equal(np.isnan(got["c"]), np.isnan(exp["c"])), name
    if cfg["algo"] == "ncem":
        assert np.array_equal(got["c"], exp["c"]), name
    assert maxdiff(got["c"], exp["c"]) <= TOL, name
    assert np.array_equal(got["center"], exp["center"]), name
    assert maxdiff(got["disp"], exp["disp"]) <= TOL and maxdiff(got["prop"], exp["prop"]) <= TOL, name
    assert_crit_close(got["crit"], exp["crit"], 1e-6, name)
    assert (got["n_zero_density"] > 0) == bool(exp["zero_density"]), name


def test_fixtures_do_reach_the_overflow_branches():
    """(what the cases are for: M = -inf with finite labels; NaN rows; an emptied class)"""
    crit = {n: load_case(n)["expected"]["crit"] for n in HEAVY}
    status = {n: int(load_case(n)["expected"]["status"]) for n in HEAVY}
    assert np.isneginf(crit["cov500_ncem_sk"][3]) and np.isfinite(crit["cov500_ncem_sk"][[0, 1, 2, 4]]).all()
    assert np.isfinite(crit["cov60_ncem_sk"]).all()
    assert status["nan_ring_ncem"] == 2 and status["nan_ring_nem"] == 2 and status["straddle709_nem"] == 2
    assert np.isnan(crit["nan_ring_nem"][[1, 2, 3, 5]]).all()
    assert bool(load_case("cov500_w1500_ncem")["expected"]["zero_density"])
    assert not load_case("cov500_w1500_ncem")["meta"]["converged"]


def test_heavy_fixtures_in_a_lockstep_batch(gpu_lib):
    """All eighteen cases as ONE lock-step batch (nemgpu_run_many: different sizes, algorithms, dispersion models, members
    that stop with an emptied class at their first or third iteration while the others go on for 30): every member
    equals the reference's fixture and its own solo run bit for bit, the non-finite criteria included."""
    from pangenomenem_amd.engine import NemEngine, run_many
    cases = [load_case(n) for n in HEAVY]
    engines, solo = [], []
    for case in cases:
        cfg = case["cfg"]
        eng = NemEngine(case["x"].shape[0], case["x"].shape[1], case["k"])
        eng.set_matrix(case["x"]); eng.set_graph(case["nei"]); eng.set_params(case["prop"], case["center"], case["disp"])
        eng.configure(algo=cfg["algo"], beta=cfg["beta"], disper=cfg["disper"], propor=cfg["propor"], cvtest=cfg["cvtest"],
                      cvthres=cfg["cvthres"], it_max=cfg["it_max"], param_fix=cfg["param_fix"], tie="libc",
                      seed=case["meta"]["libc_seed"])
        solo.append(eng.run())
        engines.append(eng)
    many = run_many(engines)
    for name, case, a, b in zip(HEAVY, cases, solo, many):
        check_against_fixture(a, case, name + " (solo)")
        check_against_fixture(b, case, name + " (lock step)")
        assert a["iters"] == b["iters"] and a["status"] == b["status"] and a["tie_draws"] == b["tie_draws"], name
        for key in ("c", "center", "disp", "prop", "nbobs_k", "crit"):
            assert np.array_equal(a[key], b[key], equal_nan=True), (name, key)
    for e in engines:
        e.close()


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo")])
@pytest.mark.parametrize("name", [n for n in HEAVY if "_ncem" in n])
def test_heavy_fixtures_family_sharded(gpu_lib, name, world, backend):
    """ShardedNem (families in contiguous blocks, two all-gathers per iteration) under the reference's tie stream: the
    NaN rows' ties are broken by random() in the reference's order across the ranks."""
    from tests.test_gpu_distributed_libc import _run
    exp = load_case(name)["expected"]
    for o in _run(world, backend, dict(kind="golden", name=name)):
        assert int(o["status"]) == int(exp["status"]) and int(o["iters"]) == int(exp["iters"]), name
        if int(exp["status"]) == 2:
            continue
        assert bool(o["converged"]) == bool(exp["converged"])
        assert np.array_equal(o["labels"], exp["c"].argmax(1))
        assert np.array_equal(o["center"], exp["center"])
        assert maxdiff(o["disp"], exp["disp"]) <= TOL and maxdiff(o["prop"], exp["prop"]) <= TOL


def _fuzzy_worker(rank, world, initfile, name, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from pangenomenem_amd.distributed import Comm, ShardedFuzzyNem
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        case = load_case(name)
        cfg = case["cfg"]
        job = ShardedFuzzyNem.from_problem(case["x"], case["nei"], case["k"], case["prop"], case["center"], case["disp"], cfg["beta"],
                                           rank, world, 0, disper=cfg["disper"])
        res = job.run(cfg["it_max"])
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), c=job.memberships(), iters=res["iters"], status=res["status"],
                 emptyk=res["emptyk"], **job.params())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["cov60_nem_sk", "cov500_nem_sk", "mid88_709_nem", "edge88_nem", "straddle709_nem"])
def test_heavy_fuzzy_fixtures_sharded_over_families_and_organisms(gpu_lib, name):
    """fuzzy NEM on two ranks (E-step over families, M-step's i-ordered chains over organisms): the reference's
    memberships within 1e-6 where beta * sum(w) is between 88 and 709, its emptied class where a NaN row gets in"""
    import tempfile
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemghf_")
    mp.spawn(_fuzzy_worker, args=(2, os.path.join(outdir, "rdv"), name, outdir), nprocs=2, join=True)
    exp = load_case(name)["expected"]
    for r in range(2):
        o = np.load(os.path.join(outdir, "rank%d.npz" % r))
        assert int(o["status"]) == int(exp["status"]) and int(o["iters"]) == int(exp["iters"]), name
        if int(exp["status"]) == 2:
            assert int(o["emptyk"]) > 0
            continue
        assert maxdiff(o["c"], exp["c"]) <= TOL and np.array_equal(o["c"].argmax(1), exp["c"].argmax(1))
        assert np.array_equal(o["center"], exp["center"]) and maxdiff(o["disp"], exp["disp"]) <= TOL and maxdiff(o["prop"], exp["prop"]) <= TOL


def heavy_problem(seed):
    """tests/test_gpu_fuzz.py's random problem with a graph whose weights are (kind 0) integers U[1, 600] -- the
    coverage weights of a 600-organism chunk -- or (kinds 1, 2) chosen per site so that beta * sum(w) falls within +-2 of
    709.78 (log(DBL_MAX): the double exp of nem_alg.c:2584) or of 88.72 (log(FLT_MAX): the float zi of nem_alg.c:2740)."""
    from tests.test_gpu_fuzz import random_problem
    x, nei, k, prop, center, disp, cfg = random_problem(5000 + seed)
    rng = np.random.Generator(np.random.PCG64(90000 + seed))
    n = x.shape[0]
    if cfg["beta"] == 0.0:
        cfg["beta"] = float(rng.choice([0.3, 0.5, 1.0]))
    deg = rng.integers(1, 6, size=n) if n > 1 else np.zeros(n, np.int64)
    ptr = np.zeros(n + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    idx = rng.integers(0, max(n, 1), size=int(ptr[-1])).astype(np.int32)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        w = rng.integers(1, 601, size=len(idx)).astype(np.float32)
    else:
        target = 709.78 if kind == 1 else 88.72
        share = rng.dirichlet(np.ones(8), size=n)                       # how a site's weight sum splits over its edges
        w = np.zeros(len(idx), np.float32)
        for i in range(n):
            t = (target + rng.uniform(-2.0, 2.0)) / cfg["beta"]
            s = share[i, :deg[i]] / share[i, :deg[i]].sum()
            w[ptr[i]:ptr[i + 1]] = (t * s).astype(np.float32)
    cfg["it_max"] = int(rng.choice([1, 3, 7]))
    return x, (ptr, idx, w), k, prop, center, disp, cfg


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("NEM_FUZZ_HEAVY_SEEDS", "90")))))
def test_heavy_weight_problem(gpu_lib, oracle, seed):
    from pangenomenem_amd.engine import solve
    x, nei, k, prop, center, disp, cfg = heavy_problem(seed)
    want = oracle.run(x, nei, k, prop, center, disp, **cfg)
    got = solve(x, nei, k, prop, center, disp, **cfg)
    ctx = (seed, x.shape, k, cfg)
    assert got["status"] == want["status"], ctx
    assert got["iters"] == want["iters"] and got["converged"] == want["converged"], ctx
    if want["status"] == 2:
        assert got["emptyk"] == want["emptyk"], ctx
        return
    assert np.array_equal(np.isnan(got["c"]), np.isnan(want["c"])), ctx     # NaN rows (exp overflow): the same places
    if cfg["algo"] == "ncem":
        assert np.array_equal(got["c"], want["c"]), ctx
    assert maxdiff(got["c"], want["c"]) <= TOL, ctx
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, (key, ctx)
    assert np.array_equal(np.nan_to_num(got["center"], nan=-7), np.nan_to_num(want["center"], nan=-7)), ctx
    assert got["n_zero_density"] == want["n_zero_density"], ctx
    assert_crit_close(got["crit"], want["crit"], 1e-5, ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,beta,seed", [(300, 600, 1.0, 4), (300, 600, 0.5, 4), (400, 200, 1.0, 5)])
def test_heavy_weight_random_starts_on_the_reference_tie_stream(gpu_lib, oracle, n, d, beta, seed):
    """RandNemAlgo on a graph with the caller's coverage weights (U[1, d]) under TIE_LIBC.  Where beta * sum(w) passes 709
    a site's row is NaN and ComputeMAP's NaN rules redraw it in EVERY sweep: the starts draw behind their initial sweeps
    (case 1: all of them, case 2: three of six -- the lock-step bet is lost mid-round and the starts behind the one that
    drew are redone, case 3: none), and the chosen criterion M is -inf for every start, so the first start stays the
    best (nem_alg.c:1676-1697).  Engine == oracle."""
    from pangenomenem_amd.engine import NemEngine
    x, _ = synth.bernoulli_pa_matrix(n, d, seed)
    nei = synth.contiguity_graph(n, seed, weights="coverage", d=d)
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x)
    eng.set_graph(nei)
    eng.configure(algo="ncem", beta=beta, disper="sk_", propor="pk", it_max=8, tie="libc", seed=seed)
    got = eng.run_random(n_starts=6, rng_seed=seed)
    how = eng.random_start_counters()
    want = oracle.run_random(x, nei, 3, n_starts=6, rng_seed=seed, algo="ncem", disper="sk_", beta=beta, it_max=8, tie="libc")
    assert how["in_lockstep"] + how["alone"] == 6
    if (d, beta) == (600, 0.5):
        assert how["redone"] > 0, how
    assert got["status"] == want["status"] and got["best_start"] == want["best_start"] and got["iters"] == want["iters"]
    assert np.array_equal(got["c"], want["c"])
    assert np.array_equal(got["center"], want["center"])
    for key in ("disp", "prop"):
        assert np.max(np.abs(got[key] - want[key])) <= 1e-6, key
    assert_crit_close(got["crit"], want["crit"], 1e-6, "criteria")
    eng.close()

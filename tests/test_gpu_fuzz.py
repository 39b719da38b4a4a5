"""Differential fuzzing of the whole loop: seeded random problems (sizes, class counts, algorithms, dispersion and
proportion models, beta, graph density and weights, hand-made initial parameters with 0 / 0.5 / > 0.5 dispersions and
half centres, fixed parameters, tie rules) solved by the HIP engine and by the oracle; everything the reference
defines must agree (labels bit-exact, posteriors / epsilon / pi within 1e-6, centres, iteration count, status)."""
import os

import numpy as np
import pytest

from tests.util import maxdiff

pytestmark = pytest.mark.gpu
TOL = 1e-6


def random_problem(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(rng.choice([1, 2, 3, 17, 64, 65, 130, 257, 400, 777, 1200]))
    d = int(rng.choice([1, 2, 5, 31, 32, 33, 64, 97, 128, 200, 333]))
    k = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 8, 12]))
    groups = max(k, 2)
    z = rng.integers(0, groups, size=n)
    profile = rng.random((groups, d)) < rng.uniform(0.1, 0.9)
    noise = rng.random((n, d)) < rng.uniform(0.0, 0.3)
    x = (profile[z] ^ noise).astype(np.uint8)
    if rng.random() < 0.2:
        x[:, rng.integers(0, d)] = rng.integers(0, 2)                  # a constant organism
    if rng.random() < 0.2 and n > 4:
        x[1] = x[0]; x[2] = x[0]                                       # identical families (ties)
    nei = None
    if n > 1 and rng.random() < 0.8:
        deg = rng.integers(0, 5, size=n)
        if rng.random() < 0.3:
            deg[rng.integers(0, n, size=max(1, n // 4))] = 0          # isolated families
        ptr = np.zeros(n + 1, np.int32)
        ptr[1:] = np.cumsum(deg)
        idx = rng.integers(0, n, size=int(ptr[-1])).astype(np.int32)  # self-loops and repeats allowed, as in a .nei
        wkind = rng.integers(0, 3)
        w = (np.ones(len(idx)) if wkind == 0 else rng.integers(1, 9, size=len(idx)) if wkind == 1
             else rng.uniform(0.05, 3.0, size=len(idx))).astype(np.float32)
        nei = (ptr, idx, w)
    prop = rng.dirichlet(np.ones(k)).astype(np.float32) if rng.random() < 0.5 else np.full(k, 1.0 / k, np.float32)
    center = rng.choice(np.array([0.0, 1.0, 0.5], np.float32), size=(k, d), p=[0.45, 0.45, 0.10]).astype(np.float32)
    kind = rng.integers(0, 4)
    if kind == 0:
        disp = np.repeat(rng.uniform(0.02, 0.5, size=(k, 1)), d, axis=1)
    elif kind == 1:
        disp = rng.uniform(0.01, 0.6, size=(k, d))
    elif kind == 2:
        disp = np.repeat(rng.choice([0.5, 0.1, 0.7, 1e-3], size=(k, 1)), d, axis=1)
    else:
        disp = rng.uniform(0.05, 0.45, size=(k, d))
        disp[rng.integers(0, k), rng.integers(0, d)] = 0.0             # a null dispersion (zero densities)
    cfg = dict(algo=str(rng.choice(["ncem", "nem"])), beta=float(rng.choice([0.0, 0.3, 1.0, 2.5])),
               disper=str(rng.choice(["sk_", "skd", "s__", "s_d"])), propor=str(rng.choice(["pk", "p_"])),
               it_max=int(rng.choice([0, 1, 3, 7, 15])), param_fix=bool(rng.random() < 0.15),
               tie=str(rng.choice(["hash", "first", "libc"])), seed=int(rng.integers(0, 1000)))
    return x, nei, k, prop, center, disp.astype(np.float32), cfg


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("NEM_FUZZ_SEEDS", "120")))))
def test_random_problem(gpu_lib, oracle, seed):
    from pangenomenem_amd.engine import solve
    x, nei, k, prop, center, disp, cfg = random_problem(seed)
    want = oracle.run(x, nei, k, prop, center, disp, **cfg)
    got = solve(x, nei, k, prop, center, disp, **cfg)
    ctx = (seed, x.shape, k, cfg)
    assert got["status"] == want["status"], ctx
    assert got["iters"] == want["iters"] and got["converged"] == want["converged"], ctx
    if want["status"] == 2:
        assert got["emptyk"] == want["emptyk"], ctx
    finite = np.isfinite(want["c"]).all()
    if finite:
        assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1)), ctx
        if cfg["algo"] == "ncem":
            assert np.array_equal(got["c"], want["c"]), ctx
        assert maxdiff(got["c"], want["c"]) <= TOL, ctx
    else:                                                     # NaN posteriors (exp overflow): same places
        assert np.array_equal(np.isnan(got["c"]), np.isnan(want["c"])), ctx
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, (key, ctx)
    assert np.array_equal(np.nan_to_num(got["center"], nan=-7), np.nan_to_num(want["center"], nan=-7)), ctx
    assert got["n_zero_density"] == want["n_zero_density"], ctx


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("NEM_FUZZ_LARGE_SEEDS", "10")))))
def test_random_large_problem(gpu_lib, oracle, seed):
    """Sizes where the other code paths live: sums above 2^24 (sequential dispersion chains in k_finish), long
    chains (fast-forward over many binades), several density tiles per class, 1024-site sweep blocks."""
    from pangenomenem_amd import synth
    from pangenomenem_amd.engine import solve
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    n = int(rng.integers(18000, 80000))
    d = int(rng.choice([260, 500, 777, 1024, 1025, 1500, 2300]))
    k = int(rng.choice([2, 3, 3, 4]))
    x, _ = synth.grouped_pa_matrix(n, d, 1000 + seed, groups=max(k, 3), p_in=float(rng.uniform(0.7, 0.97)),
                                   p_out=float(rng.uniform(0.02, 0.3)))
    nei = synth.contiguity_graph(n, seed, chord_frac=float(rng.uniform(0.0, 0.3)))
    prop, center, disp = synth.kclass_init(x, k, eps=float(rng.uniform(0.05, 0.4)))
    cfg = dict(algo=str(rng.choice(["ncem", "ncem", "nem"])), beta=float(rng.choice([0.0, 0.5, 1.0])),
               disper=str(rng.choice(["sk_", "sk_", "skd", "s__", "s_d"])), propor="pk", it_max=3, tie="hash", seed=seed)
    want = oracle.run(x, nei, k, prop, center, disp, **cfg)
    got = solve(x, nei, k, prop, center, disp, **cfg)
    ctx = (seed, x.shape, k, cfg)
    assert got["status"] == want["status"] and got["iters"] == want["iters"], ctx
    assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1)), ctx
    if cfg["algo"] == "ncem":
        assert np.array_equal(got["c"], want["c"]), ctx
    assert maxdiff(got["c"], want["c"]) <= TOL, ctx
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, (key, ctx)
    assert np.array_equal(got["center"], want["center"]), ctx


@pytest.mark.parametrize("n,d,tie", [(12000, 2800, "hash"), (66000, 2800, "hash"), (12000, 2800, "libc")])
def test_every_density_underflows(gpu_lib, oracle, n, d, tie):
    """Wide matrices drive every density to zero (exp(-dk) underflows): every site takes the uniform-posterior
    branch of nem_alg.c:2603-2613 and is counted.  Grids of 47 (256-site) and 65 (1024-site) blocks: the tally
    goes through the last-block counters from the second sweep on, and must still equal the reference's."""
    from pangenomenem_amd import synth
    from pangenomenem_amd.engine import solve
    x, _ = synth.bernoulli_pa_matrix(n, d, 5)
    nei = synth.contiguity_graph(n, 5)
    prop, center, disp = synth.default_init(d)
    # (tie = libc: every site of every sweep draws from the reference's stream -- the draw table slides batch by batch)
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=3, tie=tie, seed=3)
    want = oracle.run(x, nei, 3, prop, center, disp, **cfg)
    got = solve(x, nei, 3, prop, center, disp, **cfg)
    assert want["n_zero_density"] > n // 2               # the regime this test is about
    assert got["n_zero_density"] == want["n_zero_density"]
    assert got["status"] == want["status"] and got["iters"] == want["iters"]
    assert np.array_equal(got["c"], want["c"])
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, key

"""k_sweep_fused -- the first relaxation rounds of an E2 sweep (ComputePartitionNEM's in-place site sweep,
nem_alg.c:2330-2405, as rounds towards its unique fixed point) in ONE launch whose blocks meet between rounds -- against
the one-launch-per-round form and the oracle: the same labels, the same number of rounds to the fixed point, on graphs
where changes run across many blocks, on both block geometries, and when a launch's rounds are not enough."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import maxdiff, random_hard_partition

pytestmark = pytest.mark.gpu


@contextmanager
def env(**kw):
    """the engine reads these when it is created"""
    old = {k: os.environ.get(k) for k in kw}
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_both(x, nei, k, prop, center, disp, fused_rounds=None, **cfg):
    from pangenomenem_amd.engine import NemEngine
    out = []
    for fused in ("0", "1"):
        with env(NEM_MI355X_FUSED_SWEEP=fused, NEM_MI355X_FUSED_ROUNDS=fused_rounds):
            eng = NemEngine(x.shape[0], x.shape[1], k)
        eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp); eng.configure(**cfg)
        res = eng.run()
        res["counters"] = eng.sweep_counters()
        eng.close()
        out.append(res)
    return out


@pytest.mark.parametrize("n,d,k,weights,tie", [(20000, 60, 3, "small", "hash"), (20000, 60, 3, "coverage", "hash"),
                                               (9000, 33, 5, "small", "first"), (70000, 40, 3, "small", "hash"),
                                               (131072, 20, 2, "coverage", "hash"), (5000, 24, 10, "small", "hash")])
def test_fused_rounds_equal_one_launch_per_round(gpu_lib, oracle, n, d, k, weights, tie):
    x, _ = synth.ushaped_pa_matrix(n, d, 3)
    nei = synth.contiguity_graph(n, 3, chord_frac=0.3, weights=weights, d=d)
    if k == 3:
        prop, center, disp = synth.default_init(d)
    else:
        prop, center, disp = synth.kclass_init(x, k)
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", it_max=12, tie=tie, seed=5)
    classic, fused = run_both(x, nei, k, prop, center, disp, **cfg)
    assert classic["counters"]["fused_launches"] == 0
    assert fused["counters"]["fused_launches"] > 0 and fused["counters"]["fused_failed"] == 0 and fused["counters"]["fused_on"]
    for key in ("iters", "status", "converged", "sweep_rounds", "n_zero_density"):
        assert classic[key] == fused[key], key                # (sweep_rounds: the same rounds to every fixed point)
    for key in ("c", "center", "disp", "prop", "nbobs_k", "crit"):
        assert np.array_equal(classic[key], fused[key], equal_nan=True), key
    want = oracle.run(x, nei, k, prop, center, disp, **cfg)
    assert want["iters"] == fused["iters"] and np.array_equal(want["c"], fused["c"])
    assert np.array_equal(want["center"], fused["center"]) and maxdiff(want["disp"], fused["disp"]) <= 1e-6


def domino(n):
    """every site reads only its left neighbour; two identical classes (flat densities), everybody in class 1, site 0
    has no neighbour and ties to class 0 under the 'first' rule: the flip runs down the whole path, 64 sites per round
    inside a block (the cap of the block-local steps) and one block boundary per round at most"""
    x = np.zeros((n, 4), np.uint8); x[:, 0] = 1
    ptr = np.zeros(n + 1, np.int32); ptr[2:] = np.arange(1, n)
    idx = np.arange(0, n - 1, dtype=np.int32)
    w = np.full(n - 1, 4.0, np.float32)
    prop = np.array([0.5, 0.5], np.float32)
    center = np.tile(np.array([1, 0, 0, 0], np.float32), (2, 1))
    disp = np.full((2, 4), 0.2, np.float32)
    return x, (ptr, idx, w), prop, center, disp


@pytest.mark.parametrize("n,fused_rounds", [(900, 16), (5000, 4), (5000, 16), (70000, 16)])
def test_domino_runs_through_the_blocks(gpu_lib, oracle, n, fused_rounds):
    """One sweep that needs n/64 rounds and more: inside one fused launch (900 sites, 16 rounds), and with the host going
    on round by round where the launch's rounds are not enough -- the labels are the sequential sweep's."""
    from pangenomenem_amd.engine import NemEngine
    x, nei, prop, center, disp = domino(n)
    c0 = np.zeros((n, 2), np.float32); c0[:, 1] = 1.0
    pk, _, _ = oracle.density(x, prop, center, disp)
    assert np.all(pk[:, 0] == pk[:, 1])
    want, _ = oracle.sweep(c0, nei, 1.0, pk, True, tie="first")
    assert np.all(want[:, 0] == 1.0)
    rounds = {}
    for fused in ("0", "1"):
        with env(NEM_MI355X_FUSED_SWEEP=fused, NEM_MI355X_FUSED_ROUNDS=fused_rounds):
            eng = NemEngine(n, 4, 2)
        eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp)
        eng.configure(algo="ncem", beta=1.0, disper="sk_", tie="first")
        eng.density()
        eng.set_partition(c0)
        rounds[fused] = eng.sweep(1.0)
        got = eng.partition()
        cnt = eng.sweep_counters()
        eng.close()
        assert np.array_equal(got, want), fused
        assert (cnt["fused_launches"] > 0) == (fused == "1") and cnt["fused_failed"] == 0
    # (beyond the 64-slot flag window the host's round numbers skip some values to keep the buffers' parity: the two
    #  counts are comparable only below it)
    assert min(rounds.values()) >= n // 64
    if n == 900:
        assert rounds["0"] == rounds["1"] <= 16               # (all inside the one launch)


def test_engines_side_by_side_keep_meeting(gpu_lib, oracle):
    """Several engines on threads of their own, each with its own fused launches in flight next to the others' kernels:
    every launch's blocks still meet (79 blocks of 256 threads per engine: the grids are resident together), results
    equal the engine alone."""
    import threading
    from pangenomenem_amd.engine import NemEngine
    n, d = 20000, 48
    probs = []
    for s in range(6):
        x, _ = synth.ushaped_pa_matrix(n, d, 40 + s)
        probs.append((x, synth.contiguity_graph(n, 40 + s, chord_frac=0.2)))
    prop, center, disp = synth.default_init(d)
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", it_max=20, tie="hash", seed=1)
    alone = []
    for x, nei in probs:
        eng = NemEngine(n, d, 3); eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp); eng.configure(**cfg)
        alone.append(eng.run()); eng.close()
    got = [None] * len(probs)
    cnt = [None] * len(probs)

    def work(i):
        x, nei = probs[i]
        eng = NemEngine(n, d, 3); eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp); eng.configure(**cfg)
        for _ in range(5):
            got[i] = eng.run()
        cnt[i] = eng.sweep_counters()
        eng.close()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(probs))]
    with env(NEM_MI355X_FUSED_SWEEP="1"):                    # (off by default since the A/B of round 4; read at engine creation)
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    for a, b, c in zip(alone, got, cnt):
        assert c["fused_launches"] > 0 and c["fused_failed"] == 0
        assert a["iters"] == b["iters"] and np.array_equal(a["c"], b["c"]) and np.array_equal(a["crit"], b["crit"], equal_nan=True)

"""The algorithmic claim behind the GPU E2 kernel, checked on CPU with the oracle: Jacobi-style
relaxation rounds over "new for j < i, old for j >= i" converge, bit for bit, to the reference's
sequential in-place (Gauss-Seidel, UPDATE_SEQ) sweep -- and a plain Jacobi sweep (UPDATE_PARA) does not."""
import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import random_fuzzy_partition, random_hard_partition


def relax_to_fixed_point(oracle, nei, beta, pk, ncem, c_old, max_rounds=10000):
    n, k = c_old.shape
    guess = c_old.copy()
    rounds = 0
    while True:
        out = np.zeros_like(c_old)
        changed = oracle.relax_round(0, n, nei, beta, pk, ncem, c_old, guess, out, tie="hash", seed=3, sweep_id=7)
        rounds += 1
        if changed == 0:
            return out, rounds
        guess = out
        assert rounds < max_rounds


@pytest.mark.parametrize("ncem", [True, False])
@pytest.mark.parametrize("n,d", [(3000, 15), (1500, 40)])
def test_rounds_reach_sequential_sweep(oracle, ncem, n, d):
    x, _ = synth.bernoulli_pa_matrix(n, d, 17)
    nei = synth.contiguity_graph(n, 17, chord_frac=0.3)
    prop, center, disp = synth.default_init(d)
    pk, _, _ = oracle.density(x, prop, center, disp)
    c0 = random_hard_partition(n, 3, 2) if ncem else random_fuzzy_partition(n, 3, 2)
    want, _ = oracle.sweep(c0, nei, 0.5, pk, ncem, tie="hash", seed=3, sweep_id=7)
    got, rounds = relax_to_fixed_point(oracle, nei, 0.5, pk, ncem, c0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert 2 <= rounds < n
    # a single round from the old partition is the PARALLEL update (UPDATE_PARA) -- a different algorithm
    jac = np.zeros_like(c0)
    oracle.relax_round(0, n, nei, 0.5, pk, ncem, c0, c0, jac, tie="hash", seed=3, sweep_id=7)
    if d == 15:                                    # weak densities: the neighbourhood term matters
        assert not np.array_equal(jac, want)


def test_domino_worst_case_still_exact(oracle):
    """A path graph on which one label flip propagates site by site: the rounds need ~n iterations but
    still end on the sequential answer (the fixed point is unique: the system is lower-triangular)."""
    n, k = 60, 2
    ptr = np.zeros(n + 1, np.int32); idx = []; w = []
    for i in range(n):
        if i > 0:
            idx.append(i - 1); w.append(4.0)
        ptr[i + 1] = len(idx)
    nei = (ptr, np.array(idx, np.int32), np.array(w, np.float32))
    pk = np.full((n, k), 0.5)                      # flat densities: the left neighbour decides
    pk[0] = [0.9, 0.1]
    c0 = np.zeros((n, k), np.float32); c0[:, 1] = 1.0      # everybody starts in class 1
    want, _ = oracle.sweep(c0, nei, 1.0, pk, True, tie="first")
    assert np.all(want[:, 0] == 1.0)               # the flip at site 0 runs down the whole path
    guess, rounds = c0.copy(), 0
    while True:
        out = np.zeros_like(c0)
        ch = oracle.relax_round(0, n, nei, 1.0, pk, True, c0, guess, out, tie="first")
        rounds += 1
        if ch == 0:
            break
        guess = out
    assert np.array_equal(out, want) and rounds == n + 1


def test_sharded_rounds_equal_global_round(oracle):
    """Running a round shard by shard (what each GPU does) equals running it over all sites."""
    n, d = 2000, 15
    x, _ = synth.bernoulli_pa_matrix(n, d, 5)
    nei = synth.contiguity_graph(n, 5)
    prop, center, disp = synth.default_init(d)
    pk, _, _ = oracle.density(x, prop, center, disp)
    c0 = random_hard_partition(n, 3, 1)
    g = random_hard_partition(n, 3, 9)
    whole = np.zeros_like(c0)
    oracle.relax_round(0, n, nei, 0.5, pk, True, c0, g, whole)
    from pangenomenem_amd.distributed import shard_bounds, slice_graph
    parts = np.zeros_like(c0)
    for r in range(3):
        lo, hi, _ = shard_bounds(n, 3, r)
        oracle.relax_round(lo, hi, slice_graph(nei, lo, hi), 0.5, pk[lo:hi], True, c0, g, parts)
    assert np.array_equal(parts, whole)

"""Pins the oracle to the compiled, unmodified reference (oracle/_ref) on inputs beyond the committed
fixtures.  Runs where /root/reference (or a prebuilt oracle/_ref) exists; skipped elsewhere."""
import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import bits_equal, random_fuzzy_partition, random_hard_partition

CASES = [
    (900, 33, 0.5, "ncem", "sk_", "pk", 40),
    (900, 33, 0.5, "nem", "skd", "pk", 15),
    (700, 130, 1.5, "nem", "sk_", "p_", 15),
    (1300, 8, 0.2, "ncem", "s_d", "pk", 40),
    (500, 257, 0.5, "nem", "s__", "pk", 10),
]


@pytest.mark.parametrize("n,d,beta,algo,disper,propor,it_max", CASES)
def test_full_loop_bit_exact(oracle, reference, n, d, beta, algo, disper, propor, it_max):
    x, _ = synth.bernoulli_pa_matrix(n, d, n + d)
    nei = synth.contiguity_graph(n, n + d, chord_frac=0.1)
    prop, center, disp = synth.default_init(d)
    a = oracle.run(x, nei, 3, prop, center, disp, algo=algo, beta=beta, disper=disper, propor=propor, it_max=it_max,
                   tie="libc", seed=777)
    b = reference.classify(x, nei, 3, prop, center, disp, algo=algo, beta=beta, disper=disper, propor=propor,
                           it_max=it_max, seed=777)
    assert a["status"] == b["status"] and a["iters"] == b["iters"] and a["converged"] == b["converged"]
    for key in ("c", "prop", "center", "disp", "nbobs_k", "crit"):
        assert bits_equal(a[key], b[key]), key


@pytest.mark.parametrize("hard", [True, False])
@pytest.mark.parametrize("disper", ["s__", "sk_", "s_d", "skd"])
def test_mstep_bit_exact(oracle, reference, hard, disper):
    n, d, k = 2500, 48, 4
    x, _ = synth.grouped_pa_matrix(n, d, 3, groups=4)
    c = random_hard_partition(n, k, 1) if hard else random_fuzzy_partition(n, k, 1)
    prop = np.full(k, 0.25, np.float32)
    center = np.full((k, d), 0.5, np.float32)
    disp = np.full((k, d), 0.3, np.float32)
    a = oracle.mstep(x, c, disper, "pk", prop, center, disp)
    b = reference.estim_para(x, c, disper, "pk", prop, center, disp)
    assert a["status"] == b["status"] and a["emptyk"] == b["emptyk"]
    for key in ("prop", "center", "disp", "nbobs_k", "nbobs_kd", "iner"):
        assert bits_equal(a[key], b[key]), key


def test_reference_files_equal_reference_memory(reference, tmp_path):
    """nem() on the five ASCII files and ClassifyByNem on the in-memory structs agree to the print
    precision of .uf/.mf -- i.e. our input writer emits what the reference's readers expect."""
    from pangenomenem_amd import nemfiles
    n, d = 600, 21
    x, _ = synth.bernoulli_pa_matrix(n, d, 6)
    nei = synth.contiguity_graph(n, 6)
    prop, center, disp = synth.default_init(d)
    base = nemfiles.write_nem_inputs(str(tmp_path), x, nei, prop, center, disp)
    rc = reference.nem(base, 3, algo=b"nem", it_max=10)
    assert rc == 0
    mem = reference.classify(x, nei, 3, prop, center, disp, algo="nem", it_max=10)
    uf = np.loadtxt(base + ".uf", dtype=np.float64)
    assert np.max(np.abs(uf - mem["c"])) <= 5.01e-4
    labels, params, m_crit, _ = nemfiles.read_nem_outputs(str(tmp_path), d)
    assert abs(m_crit - mem["crit"][3]) <= 1e-5 * abs(mem["crit"][3])
    assert len(labels) == n and set(labels) <= {"P", "S", "C"}


@pytest.mark.parametrize("n,d,k,algo,disper,starts,seed", [
    (600, 20, 3, "ncem", "sk_", 8, 1), (600, 20, 3, "nem", "skd", 8, 7), (1000, 33, 4, "ncem", "skd", 6, 3),
    (300, 12, 2, "ncem", "s__", 10, 5), (400, 9, 3, "nem", "s_d", 5, 11), (1, 6, 2, "ncem", "sk_", 3, 2)])
def test_random_starts_bit_exact(oracle, reference, n, d, k, algo, disper, starts, seed):
    """init_mode = INIT_RANDOM (RandNemAlgo): same libc stream (srandom(seed)) for the centre draws and the
    tie-breaks, best start by M, final EstimPara -- everything bit for bit."""
    x, _ = synth.bernoulli_pa_matrix(n, d, seed)
    nei = synth.contiguity_graph(n, seed) if n > 1 else None
    a = oracle.run_random(x, nei, k, n_starts=starts, rng_seed=seed, algo=algo, disper=disper, beta=0.5, it_max=30)
    b = reference.classify_random(x, nei, k, n_starts=starts, rng_seed=seed, algo=algo, disper=disper, beta=0.5,
                                  it_max=30)
    assert a["status"] == b["status"]
    if b["status"] == 0:
        assert a["best_start"] == b["best_start"]
        for key in ("c", "prop", "center", "disp", "nbobs_k", "crit"):
            assert bits_equal(a[key], b[key]), key


@pytest.mark.parametrize("n,d,algo,disper,thres", [(2048, 15, "ncem", "sk_", 1e-4), (2048, 15, "nem", "sk_", 1e-3),
                                                   (1500, 40, "nem", "skd", 1e-4), (1500, 40, "ncem", "skd", 0.02),
                                                   (900, 25, "ncem", "s__", 1e-6)])
def test_crit_convergence_bit_exact(oracle, reference, n, d, algo, disper, thres):
    """convergence = "crit" (HasConverged's CVTEST_CRIT, nem_alg.c:2090-2105): the relative move of criterion M
    between iterations, from Criteria = {0} in a run without a log (nem_exe.c:264) -- iteration count and everything
    else bit for bit."""
    x, _ = synth.ushaped_pa_matrix(n, d, 4)
    nei = synth.contiguity_graph(n, 4)
    prop, center, disp = synth.default_init(d)
    a = oracle.run(x, nei, 3, prop, center, disp, algo=algo, beta=0.5, disper=disper, cvtest="crit", cvthres=thres,
                   it_max=60, tie="libc", seed=12345)
    b = reference.classify(x, nei, 3, prop, center, disp, algo=algo, beta=0.5, disper=disper, cvtest="crit",
                           cvthres=thres, it_max=60, seed=12345)
    assert a["iters"] == b["iters"] and a["converged"] == b["converged"] and 1 < a["iters"] < 60
    for key in ("c", "prop", "center", "disp", "crit"):
        assert bits_equal(a[key], b[key]), key

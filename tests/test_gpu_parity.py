"""GPU parity tests proper: the HIP engine (through the C ABI, via ctypes) against the CPU oracle
on the same seeded inputs.  Bar (BASELINE.json north_star): integer labels bit-exact; posteriors,
epsilon, pi within 1e-6.  In practice every float below is compared bit-for-bit except the two
places a device libm call enters (exp -> pk*fk, exp -> MRF factor), which get an explicit bound."""
import ctypes as C

import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import assert_crit_close, bits_equal, maxdiff, random_fuzzy_partition, random_hard_partition, ulp_diff64

pytestmark = pytest.mark.gpu

TOL = 1e-6   # north_star tolerance for posteriors / epsilon / pi


def make_engine(x, nei, k, prop, center, disp, **cfg):
    from pangenomenem_amd.engine import NemEngine
    n, d = x.shape
    eng = NemEngine(n, d, k)
    eng.set_matrix(x)
    eng.set_graph(nei)
    eng.set_params(prop, center, disp)
    eng.configure(**cfg)
    return eng


@pytest.mark.parametrize("n,d,seed", [(2048, 15, 1), (1000, 64, 2), (777, 333, 3), (300, 1100, 4)])
def test_density_matches_oracle(gpu_lib, oracle, n, d, seed):
    x, _ = synth.bernoulli_pa_matrix(n, d, seed)
    prop, center, disp = synth.default_init(d)
    rng = np.random.Generator(np.random.PCG64(seed))
    disp = (disp * rng.uniform(0.5, 1.5, size=disp.shape)).astype(np.float32)   # per-(k,d) epsilons
    eng = make_engine(x, None, 3, prop, center, disp)
    pk, lp = eng.density()
    opk, olp, _ = oracle.density(x, prop, center, disp)
    # the float log-density chain is IEEE arithmetic only: bit-exact
    assert bits_equal(lp, olp)
    # pk*fk goes through the device exp(): allow 2 ulp (double)
    assert ulp_diff64(pk, opk) <= 2
    eng.close()


@pytest.mark.parametrize("n,d,seed", [(1500, 15, 1), (900, 500, 2), (400, 3100, 3), (700, 97, 4)])
def test_density_uniform_dispersion_fast_forward(gpu_lib, oracle, n, d, seed):
    """One epsilon per class (sk_, s__, the default .m): the kernel replaces runs of chain steps inside a
    float binade by integer adds on the bit pattern.  Must stay bit-identical to the step-by-step chain
    for any epsilon in (0, 1/2], tiny ones, 1/2 itself, centres 0 / 1/2 / 1, and thousands of organisms."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x, _ = synth.bernoulli_pa_matrix(n, d, seed)
    for trial in range(6):
        k = [3, 2, 5, 10, 3, 4][trial]
        eps = rng.uniform(1e-3, 0.5, size=k).astype(np.float32)
        if trial == 1:
            eps[:] = [0.5, 1e-7][:k]
        if trial == 4:
            eps[:] = [1e-19, 0.25, 0.4999999]
        center = rng.choice(np.array([0.0, 0.5, 1.0], np.float32), size=(k, d))
        center[0] = 1.0
        disp = np.repeat(eps[:, None], d, axis=1)
        prop = np.full(k, 1.0 / k, np.float32)
        opk, olp, _ = oracle.density(x, prop, center, disp)
        for mode in (1, 0):                           # fast-forward forced on / plain stepping
            eng = make_engine(x, None, k, prop, center, disp)
            eng.set_fast_forward(mode)
            pk, lp = eng.density()
            assert bits_equal(lp, olp), (trial, mode, eps)
            assert ulp_diff64(pk, opk) <= 2
            eng.close()


def test_density_fast_forward_falls_back_for_dispersions_above_half(gpu_lib, oracle):
    """eps > 1/2 makes log((1-eps)/eps) negative: the chain is no longer monotone, the kernel must step it."""
    n, d = 600, 900
    x, _ = synth.bernoulli_pa_matrix(n, d, 21)
    eps = np.array([0.7, 0.1, 0.9999], np.float32)
    center = np.stack([np.ones(d), np.zeros(d), np.full(d, 0.5)]).astype(np.float32)
    disp = np.repeat(eps[:, None], d, axis=1)
    prop = np.full(3, 1.0 / 3, np.float32)
    opk, olp, _ = oracle.density(x, prop, center, disp)
    eng = make_engine(x, None, 3, prop, center, disp)
    eng.set_fast_forward(1)
    pk, lp = eng.density()
    assert bits_equal(lp, olp)
    assert ulp_diff64(pk, opk) <= 2
    eng.close()


@pytest.mark.parametrize("algo,disper", [("ncem", "sk_"), ("nem", "sk_"), ("ncem", "s__"), ("ncem", "skd")])
def test_full_run_with_fast_forward_forced(gpu_lib, oracle, algo, disper):
    """Whole EM with the fast-forwarded E1 (the automatic mode would step a problem this small)."""
    from pangenomenem_amd.engine import solve
    n, d = 3000, 420
    x, _ = synth.bernoulli_pa_matrix(n, d, 33)
    nei = synth.contiguity_graph(n, 33)
    prop, center, disp = synth.default_init(d)
    cfg = dict(algo=algo, beta=0.5, disper=disper, propor="pk", it_max=12, tie="hash", seed=5)
    ref = oracle.run(x, nei, 3, prop, center, disp, **cfg)
    for mode in (1, 0):
        got = solve(x, nei, 3, prop, center, disp, fast_forward=mode, **cfg)
        assert got["iters"] == ref["iters"]
        if algo == "ncem":
            assert np.array_equal(got["c"], ref["c"])
        else:
            assert maxdiff(got["c"], ref["c"]) <= TOL
        assert maxdiff(got["disp"], ref["disp"]) <= TOL and maxdiff(got["prop"], ref["prop"]) <= TOL
        assert bits_equal(got["center"], ref["center"])


def test_density_null_dispersion_and_half_centres(gpu_lib, oracle):
    n, d = 512, 70
    x, _ = synth.bernoulli_pa_matrix(n, d, 9)
    prop, center, disp = synth.default_init(d)
    disp[0, ::7] = 0.0          # eps = 0 with mu = 1: zero density whenever x = 0 there (nem_mod.c:662-666)
    disp[2, 5] = 1e-30          # below EPSILON
    eng = make_engine(x, None, 3, prop, center, disp)
    pk, lp = eng.density()
    opk, olp, _ = oracle.density(x, prop, center, disp)
    assert bits_equal(lp, olp)
    assert np.array_equal(pk == 0.0, opk == 0.0)
    assert ulp_diff64(pk, opk) <= 2
    eng.close()


@pytest.mark.parametrize("algo", ["ncem", "nem"])
@pytest.mark.parametrize("disper", ["sk_", "skd", "s__", "s_d"])
def test_mstep_matches_oracle(gpu_lib, oracle, algo, disper):
    n, d, k = 3000, 97, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 21)
    prop, center, disp = synth.default_init(d)
    c = random_hard_partition(n, k, 5) if algo == "ncem" else random_fuzzy_partition(n, k, 5)
    if algo == "ncem":
        # force exact median ties (centre 0.5) in a few columns: as many zeros as ones inside class 1
        members = np.flatnonzero(c[:, 1] == 1.0)
        members = members[: len(members) // 2 * 2]
        c[np.setdiff1d(np.flatnonzero(c[:, 1] == 1.0), members), :] = [1, 0, 0]
        x[members[0::2], 3] = 0
        x[members[1::2], 3] = 1
    eng = make_engine(x, None, k, prop, center, disp, algo=algo, disper=disper)
    eng.set_partition(c)
    rc, ek = eng.mstep()
    got = eng.params()
    want = oracle.mstep(x, c, disper, "pk", prop, center, disp)
    assert rc == want["status"] and ek == want["emptyk"]
    for key in ("center", "disp", "prop", "nbobs_k"):
        assert bits_equal(got[key], want[key]), key
    if algo == "ncem":
        assert (got["center"][1, 3] == 0.5)
    eng.close()


def test_mstep_empty_class(gpu_lib, oracle):
    n, d, k = 500, 40, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 3)
    prop, center, disp = synth.default_init(d)
    c = np.zeros((n, k), np.float32)
    c[:, 0] = 1.0
    c[::3, :] = [0, 0, 1]
    eng = make_engine(x, None, k, prop, center, disp, algo="ncem")
    eng.set_partition(c)
    rc, ek = eng.mstep()
    want = oracle.mstep(x, c, "sk_", "pk", prop, center, disp)
    assert rc == want["status"] == 2 and ek == want["emptyk"] == 2
    got = eng.params()
    for key in ("center", "disp", "prop", "nbobs_k"):
        assert bits_equal(got[key], want[key]), key
    eng.close()


@pytest.mark.parametrize("algo", ["ncem", "nem"])
def test_sweep_is_gauss_seidel(gpu_lib, oracle, algo):
    """One E2 sweep on a path+chords graph: must equal the SEQUENTIAL in-place sweep, not a Jacobi one."""
    n, d, k = 4000, 15, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 31)
    nei = synth.contiguity_graph(n, 31)
    prop, center, disp = synth.default_init(d)
    c0 = random_hard_partition(n, k, 8) if algo == "ncem" else random_fuzzy_partition(n, k, 8)
    eng = make_engine(x, nei, k, prop, center, disp, algo=algo, beta=0.5)
    eng.density()
    eng.set_partition(c0)
    rounds = eng.sweep(0.5)
    got = eng.partition()
    pk, _, _ = oracle.density(x, prop, center, disp)
    want, _ = oracle.sweep(c0, nei, 0.5, pk, algo == "ncem", tie="hash", seed=0, sweep_id=0)
    assert rounds >= 2
    if algo == "ncem":
        assert np.array_equal(got, want)
    else:
        assert maxdiff(got, want) <= TOL
        assert np.array_equal(got.argmax(1), want.argmax(1))
    eng.close()


CASES = [
    # n, d, beta, algo, disper, it_max
    (2048, 15, 0.0, "ncem", "sk_", 100),     # BASELINE configs[0]
    (2048, 15, 0.5, "ncem", "sk_", 100),
    (2048, 15, 0.5, "nem", "sk_", 30),
    (2048, 15, 0.5, "nem", "skd", 30),
    (3000, 64, 0.5, "ncem", "skd", 100),
    (3000, 64, 0.5, "nem", "s__", 40),
    (3000, 64, 0.5, "nem", "s_d", 40),
    (5000, 40, 0.5, "nem", "sk_", 40),
    (5000, 40, 1.0, "ncem", "sk_", 100),
]


@pytest.mark.parametrize("n,d,beta,algo,disper,it_max", CASES)
def test_full_run_matches_oracle(gpu_lib, oracle, n, d, beta, algo, disper, it_max):
    x, _ = synth.bernoulli_pa_matrix(n, d, 11)
    nei = synth.contiguity_graph(n, 11)
    prop, center, disp = synth.default_init(d)
    from pangenomenem_amd.engine import solve
    got = solve(x, nei, 3, prop, center, disp, algo=algo, beta=beta, disper=disper, it_max=it_max, tie="hash", seed=7)
    want = oracle.run(x, nei, 3, prop, center, disp, algo=algo, beta=beta, disper=disper, it_max=it_max,
                      tie="hash", seed=7)
    assert got["status"] == want["status"]
    assert got["iters"] == want["iters"]
    assert got["converged"] == want["converged"]
    assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1))           # labels bit-exact
    if algo == "ncem":
        assert np.array_equal(got["c"], want["c"])
    assert maxdiff(got["c"], want["c"]) <= TOL
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, key
    assert np.array_equal(got["center"], want["center"])
    # criteria: float sums in reference order; the device exp/log may move the last digits
    assert_crit_close(got["crit"], want["crit"], 1e-6)


@pytest.mark.parametrize("k", [2, 4, 5, 7, 10])
def test_k_sweep_free_dispersion(gpu_lib, oracle, k):
    """BASELINE configs[4] (down-scaled): K in 2..10, eps_kj per class and organism."""
    n, d = 4000, 100
    x, _ = synth.grouped_pa_matrix(n, d, 5, groups=10)
    nei = synth.contiguity_graph(n, 5)
    prop, center, disp = synth.kclass_init(x, k)
    from pangenomenem_amd.engine import solve
    for algo, it_max in (("ncem", 100), ("nem", 15)):
        got = solve(x, nei, k, prop, center, disp, algo=algo, beta=0.5, disper="skd", it_max=it_max, seed=3)
        want = oracle.run(x, nei, k, prop, center, disp, algo=algo, beta=0.5, disper="skd", it_max=it_max,
                          tie="hash", seed=3)
        assert got["status"] == want["status"] and got["iters"] == want["iters"], algo
        assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1)), algo
        assert maxdiff(got["c"], want["c"]) <= TOL, algo
        assert maxdiff(got["disp"], want["disp"]) <= TOL, algo
        assert maxdiff(got["prop"], want["prop"]) <= TOL, algo


@pytest.mark.parametrize("n,d,k", [(2048, 15, 3), (1500, 33, 3), (3000, 70, 4), (2500, 300, 3), (1200, 1000, 3), (20000, 500, 3),
                                   (4000, 257, 6)])
def test_free_dispersion_fast_forward_equals_stepping(gpu_lib, oracle, n, d, k):
    """Free dispersion (skd, PPanGGOLiN's free_dispersion): every organism of a class has its own epsilon, E1's chain
    takes the general path of k_density_fused whatever the fast-forward switch says (a per-organism fast-forward was
    built, measured and dropped in round 3) -- both settings of the switch give the same run, bit
    for bit, and it equals the oracle's."""
    from pangenomenem_amd.engine import solve
    x, _ = synth.ushaped_pa_matrix(n, d, n + d)
    nei = synth.contiguity_graph(n, n + d)
    if k == 3:
        prop, center, disp = synth.default_init(d)
    else:
        prop, center, disp = synth.kclass_init(x, k)
    cfg = dict(algo="ncem", beta=0.5, disper="skd", propor="pk", it_max=8, tie="hash", seed=5)
    runs = [solve(x, nei, k, prop, center, disp, fast_forward=mode, **cfg) for mode in (1, 0)]
    a, b = runs
    assert a["iters"] == b["iters"] and a["status"] == b["status"]
    assert np.array_equal(a["c"], b["c"])
    for key in ("disp", "prop", "center", "nbobs_k", "crit"):
        assert bits_equal(np.asarray(a[key]), np.asarray(b[key])), key
    want = oracle.run(x, nei, k, prop, center, disp, **cfg)
    assert a["iters"] == want["iters"] and a["status"] == want["status"]
    if a["status"] == 0:
        assert np.array_equal(a["c"], want["c"]) and bits_equal(a["center"], want["center"])
        assert maxdiff(a["disp"], want["disp"]) <= TOL and maxdiff(a["prop"], want["prop"]) <= TOL


def test_free_dispersion_fast_forward_density_is_bit_exact(gpu_lib, oracle):
    """the densities after two free-dispersion iterations (dispersions of every size): LogPkFki equals the oracle's
    bit for bit with the fast-forward switch on and off"""
    from pangenomenem_amd.engine import NemEngine
    n, d = 3000, 420
    x, _ = synth.ushaped_pa_matrix(n, d, 77)
    nei = synth.contiguity_graph(n, 77)
    prop, center, disp = synth.default_init(d)
    for mode in (1, 0):
        eng = NemEngine(n, d, 3)
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        eng.configure(algo="ncem", beta=0.5, disper="skd", it_max=2, cvtest="none", tie="hash", seed=5)
        eng.set_fast_forward(mode)
        eng.run()
        pk = np.zeros((n, 3), np.float64); lp = np.zeros((n, 3), np.float32)
        eng._chk(eng.lib.nemgpu_get_density(eng._h, pk.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p)))
        par = eng.params()
        eng.close()
        opk, olp, _ = oracle.density(x, par["prop"], par["center"], par["disp"])
        assert bits_equal(lp, olp)
        assert ulp_diff64(pk, opk) <= 2

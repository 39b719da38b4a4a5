"""Parity at BASELINE.json's sizes: configs[1] (20 000 x 500) and configs[2]'s matrix (50 000 x 1 000)
against the oracle (it still finishes in seconds there), plus size-independent properties."""
import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import assert_crit_close, maxdiff

pytestmark = pytest.mark.gpu
TOL = 1e-6
CRIT_TOL = 1e-6      # criteria: i-ordered float sums whose terms went through the device's exp / log


def _solve_both(oracle, cfgname, algo, it_max, n=None, d=None, **kw):
    from pangenomenem_amd.engine import solve
    p = synth.make_config(cfgname, n=n, d=d)
    got = solve(p["x"], p["nei"], p["k"], p["prop"], p["center"], p["disp"], algo=algo, beta=p["beta"], it_max=it_max,
                seed=4, **kw)
    want = oracle.run(p["x"], p["nei"], p["k"], p["prop"], p["center"], p["disp"], algo=algo, beta=p["beta"],
                      it_max=it_max, tie="hash", seed=4, **kw)
    return p, got, want


def _check(got, want, algo):
    assert got["status"] == want["status"] and got["iters"] == want["iters"]
    assert got["converged"] == want["converged"]
    assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1))
    if algo == "ncem":
        assert np.array_equal(got["c"], want["c"])
    assert maxdiff(got["c"], want["c"]) <= TOL
    assert np.array_equal(got["center"], want["center"])
    assert maxdiff(got["disp"], want["disp"]) <= TOL
    assert maxdiff(got["prop"], want["prop"]) <= TOL


def test_config2_ncem_full_size(gpu_lib, oracle):
    p, got, want = _solve_both(oracle, "C2", "ncem", 100)
    _check(got, want, "ncem")
    # size-independent properties
    n = p["x"].shape[0]
    assert float(got["nbobs_k"].sum()) == n                      # class sizes partition the families
    assert abs(float(got["prop"].sum()) - 1.0) <= 1e-6
    assert np.all(got["c"].sum(1) == 1.0)


def test_config2_fuzzy_full_size(gpu_lib, oracle):
    _, got, want = _solve_both(oracle, "C2", "nem", 6)
    _check(got, want, "nem")


def test_config3_matrix_ncem_full_size(gpu_lib, oracle):
    """50 000 x 1 000 on ONE GPU (the 8-GPU sharding of configs[2] is covered by tests/test_distributed.py)."""
    p, got, want = _solve_both(oracle, "C3", "ncem", 100)
    _check(got, want, "ncem")


@pytest.mark.parametrize("k", list(range(2, 11)))
def test_config5_k_sweep_free_dispersion(gpu_lib, oracle, k):
    """BASELINE configs[4] at full size: 20 000 x 500, free dispersion (skd), every K in 2..10 -- NCEM to convergence
    (labels, iteration count, centres bit-exact; epsilon, pi within 1e-6) and fuzzy NEM for 10 iterations (posteriors
    within 1e-6), both against the oracle."""
    from pangenomenem_amd.engine import solve
    n, d = 20000, 500
    x, _ = synth.grouped_pa_matrix(n, d, 5, groups=10)
    nei = synth.contiguity_graph(n, 5)
    prop, center, disp = synth.kclass_init(x, k)
    for algo, it_max, cvtest in (("ncem", 100, "clas"), ("nem", 10, "none")):
        got = solve(x, nei, k, prop, center, disp, algo=algo, beta=0.5, disper="skd", it_max=it_max, cvtest=cvtest, seed=2)
        want = oracle.run(x, nei, k, prop, center, disp, algo=algo, beta=0.5, disper="skd", it_max=it_max, cvtest=cvtest,
                          tie="hash", seed=2)
        _check(got, want, algo)
        if want["status"] == 2:                                  # (K = 9: a class empties at the second iteration,
            assert got["emptyk"] == want["emptyk"]               #  as in the reference: no result)
            continue
        assert got["iters"] == 10 if algo == "nem" else got["converged"]
        assert_crit_close(got["crit"], want["crit"], CRIT_TOL)


def test_idempotence_and_restart(gpu_lib):
    """A converged NCEM solution is a fixed point: restarting the loop from its parameters (flag 1) converges
    at the first iteration with the same labels; with fixed parameters (flag 2) too."""
    from pangenomenem_amd.engine import solve
    p = synth.make_config("C2", n=6000, d=200)
    a = solve(p["x"], p["nei"], 3, p["prop"], p["center"], p["disp"], algo="ncem", beta=0.5, seed=1)
    assert a["converged"]
    b = solve(p["x"], p["nei"], 3, a["prop"], a["center"], a["disp"], algo="ncem", beta=0.5, seed=1)
    assert b["converged"] and b["iters"] == 1 and np.array_equal(a["c"], b["c"])
    c = solve(p["x"], p["nei"], 3, a["prop"], a["center"], a["disp"], algo="ncem", beta=0.5, seed=1, param_fix=True)
    assert np.array_equal(a["c"], c["c"]) and np.array_equal(c["disp"], a["disp"])


def test_engine_reuse_reset_and_steps(gpu_lib, oracle):
    """One engine, many runs (the reference is called repeatedly per process, ppanggolin.py:1045-1086):
    run -> reset -> step-wise run gives identical results; reconfiguring between runs works."""
    from pangenomenem_amd.engine import NemEngine
    p = synth.make_config("C1")
    nei = synth.contiguity_graph(p["x"].shape[0], 1)
    eng = NemEngine(p["x"].shape[0], p["x"].shape[1], 3)
    eng.set_matrix(p["x"]); eng.set_graph(nei); eng.set_params(p["prop"], p["center"], p["disp"])
    eng.configure(algo="ncem", beta=0.5, seed=3)
    a = eng.run()
    eng.reset(); eng.init_partition()
    r = eng.iterate(100)
    assert r["iters"] == a["iters"] and r["converged"]
    assert np.array_equal(eng.partition(), a["c"])
    eng.configure(algo="nem", beta=0.5, it_max=5, seed=3)
    f = eng.run()
    want = oracle.run(p["x"], nei, 3, p["prop"], p["center"], p["disp"], algo="nem", beta=0.5, it_max=5, tie="hash", seed=3)
    assert maxdiff(f["c"], want["c"]) <= TOL and f["iters"] == want["iters"]
    eng.close()


def test_config4_matrix_stress_properties(gpu_lib, oracle):
    """BASELINE configs[3]'s matrix (200 000 x 5 000, bit-packed 125 MB) on one GPU.  The reference has no
    meaningful answer here (densities underflow), so this is checked through properties
    that do not depend on size: E1's float log-density chain against the oracle on sampled families
    (bit-exact), the M-step against the oracle's on the GPU's own partition (bit-exact), and the
    partition invariants."""
    from pangenomenem_amd.engine import NemEngine
    n, d, k = 200000, 5000, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 4)
    nei = synth.contiguity_graph(n, 4)
    prop, center, disp = synth.default_init(d)
    eng = NemEngine(n, d, k)
    eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp)
    eng.configure(algo="ncem", beta=0.5, it_max=2, cvtest="none", seed=9)
    res = eng.run()
    assert res["iters"] == 2 or res["status"] == 2
    c = res["c"]
    assert np.all(c.sum(1) == 1.0) and float(res["nbobs_k"].sum()) == n or res["status"] == 2
    # E1 on the final parameters, sampled rows vs the oracle
    pk, lp = eng.density()
    rows = np.random.Generator(np.random.PCG64(1)).choice(n, size=3000, replace=False)
    rows.sort()
    opk, olp, _ = oracle.density(x[rows], res["prop"], res["center"], res["disp"])
    assert np.array_equal(lp[rows].view(np.uint32), olp.view(np.uint32))
    assert np.array_equal(pk[rows] == 0.0, opk == 0.0)
    # M-step on the GPU's partition vs the oracle's
    eng.set_partition(c)
    rc, ek = eng.mstep()
    got = eng.params()
    want = oracle.mstep(x, c, "sk_", "pk", res["prop"], res["center"], res["disp"])
    assert rc == want["status"] and ek == want["emptyk"]
    for key in ("center", "disp", "prop", "nbobs_k"):
        assert np.array_equal(got[key].view(np.uint32), want[key].view(np.uint32)), key
    eng.close()

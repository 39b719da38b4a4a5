"""The CPU oracle against the committed golden vectors (generated from the unmodified reference by
tests/golden/make_golden.py).  Bit-for-bit: posteriors, parameters, criteria, iteration counts."""
import numpy as np
import pytest

from tests.golden_util import case_names, load_case
from tests.util import bits_equal


@pytest.mark.parametrize("name", case_names())
def test_oracle_reproduces_reference_golden(oracle, name):
    case = load_case(name)
    cfg, exp = case["cfg"], case["expected"]
    got = oracle.run(case["x"], case["nei"], case["k"], case["prop"], case["center"], case["disp"],
                     algo=cfg["algo"], beta=cfg["beta"], disper=cfg["disper"], propor=cfg["propor"],
                     cvtest=cfg["cvtest"], cvthres=cfg["cvthres"], it_max=cfg["it_max"], param_fix=cfg["param_fix"],
                     tie="libc", seed=case["meta"]["libc_seed"])
    assert got["status"] == int(exp["status"])
    assert got["iters"] == int(exp["iters"])
    assert got["converged"] == bool(exp["converged"])
    for key in ("c", "prop", "center", "disp", "nbobs_k"):
        assert bits_equal(got[key], exp[key]), key
    if got["status"] == 0:
        assert bits_equal(got["crit"], exp["crit"]), (got["crit"], exp["crit"])
    assert (got["n_zero_density"] > 0) == bool(exp["zero_density"])


def test_underflow_case_goes_uniform(oracle):
    """D = 1150: for shell-like rows exp(-dk) underflows double in EVERY class, so their posteriors take
    the cumnum == 0 branch and become exactly 1/K (nem_alg.c:2603-2607); the rows
    that go uniform are exactly the rows whose K densities are all zero."""
    case = load_case("underflow_d1150_nem")
    c = case["expected"]["c"]
    uniform = np.all(c == np.float32(1.0 / 3), axis=1)
    assert bool(case["expected"]["zero_density"]) and 0 < uniform.sum() < len(c)
    got = oracle.run(case["x"], case["nei"], 3, case["prop"], case["center"], case["disp"], algo="nem", beta=0.0,
                     it_max=case["cfg"]["it_max"], tie="libc")
    assert np.array_equal(np.all(got["pkfki"] == 0.0, axis=1), uniform)


def test_empty_class_case(oracle):
    case = load_case("empty_class")
    assert int(case["expected"]["status"]) == 2          # STS_W_EMPTYCLASS: nem() returns 1, writes no files


def test_tie_case_has_exact_ties(oracle):
    """Two identical classes: the densities tie exactly; with the hash rule the oracle is reproducible
    and the tied sites are the same set the reference broke with random()."""
    case = load_case("ties_two_equal_classes")
    cfg = case["cfg"]
    pk, _, _ = oracle.density(case["x"], case["prop"], case["center"], case["disp"])
    assert np.all(pk[:, 0] == pk[:, 1])
    a = oracle.run(case["x"], case["nei"], 2, case["prop"], case["center"], case["disp"], algo="ncem",
                   beta=cfg["beta"], it_max=1, tie="hash", seed=5)
    b = oracle.run(case["x"], case["nei"], 2, case["prop"], case["center"], case["disp"], algo="ncem",
                   beta=cfg["beta"], it_max=1, tie="hash", seed=5)
    assert np.array_equal(a["c"], b["c"])
    assert 0.3 < a["c"][:, 0].mean() < 0.7               # labels are spread, not all "first"

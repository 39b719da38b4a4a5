"""GPU engine against the committed golden vectors of the UNMODIFIED reference, and the drop-in
nem() entry point against the reference's own output files."""
import os

import numpy as np
import pytest

from tests.golden_util import case_names, load_case
from tests.util import assert_crit_close, assert_printed_g_close, maxdiff

pytestmark = pytest.mark.gpu
TOL = 1e-6

# Two cases tie (ties_two_equal_classes by construction; k10_ncem_skd: the K-class init picks data rows as centres,
# two of the ten coincide): their labels depend on the reference's random() stream, which the fixtures were
# generated with after srandom(libc_seed).  The engine's TIE_LIBC rule draws from the same stream in the same order.
TIE_CASES = {"ties_two_equal_classes", "k10_ncem_skd"}


@pytest.mark.parametrize("name", case_names())
def test_engine_matches_reference_golden(gpu_lib, name):
    from pangenomenem_amd.engine import solve
    case = load_case(name)
    cfg, exp = case["cfg"], case["expected"]
    got = solve(case["x"], case["nei"], case["k"], case["prop"], case["center"], case["disp"], algo=cfg["algo"],
                beta=cfg["beta"], disper=cfg["disper"], propor=cfg["propor"], cvtest=cfg["cvtest"],
                cvthres=cfg["cvthres"], it_max=cfg["it_max"], param_fix=cfg["param_fix"], tie="libc",
                seed=case["meta"]["libc_seed"])
    if name in TIE_CASES:
        assert got["tie_draws"] > 0                          # (the case does exercise the stream)
    assert got["status"] == int(exp["status"])
    assert got["iters"] == int(exp["iters"])
    if got["status"] == 2:                                   # empty class: the reference writes nothing
        assert got["emptyk"] > 0
        return
    assert got["converged"] == bool(exp["converged"])
    assert np.array_equal(got["c"].argmax(1), exp["c"].argmax(1))        # integer assignments: bit-exact
    if cfg["algo"] == "ncem":
        assert np.array_equal(got["c"], exp["c"])
    assert maxdiff(got["c"], exp["c"]) <= TOL                            # posteriors
    assert np.array_equal(got["center"], exp["center"])                  # mu in {0, 1/2, 1}: exact
    assert maxdiff(got["disp"], exp["disp"]) <= TOL                      # epsilon
    assert maxdiff(got["prop"], exp["prop"]) <= TOL                      # pi
    assert maxdiff(got["nbobs_k"], exp["nbobs_k"]) <= 1e-6 * max(1.0, float(np.max(exp["nbobs_k"])))
    assert_crit_close(got["crit"], exp["crit"], 1e-6)                    # non-finite entries: the same non-finite value
    assert (got["n_zero_density"] > 0) == bool(exp["zero_density"])


@pytest.mark.parametrize("name", sorted(TIE_CASES))
@pytest.mark.parametrize("tie", ["hash", "first", "libc"])
def test_tie_cases_match_oracle_with_same_rule(gpu_lib, oracle, name, tie):
    """Where the reference draws random() the engine uses a reproducible rule; with the same rule the
    oracle and the engine agree bit for bit (labels, parameters, iteration count)."""
    from pangenomenem_amd.engine import solve
    case = load_case(name)
    cfg = case["cfg"]
    kw = dict(algo=cfg["algo"], beta=cfg["beta"], disper=cfg["disper"], propor=cfg["propor"], it_max=cfg["it_max"])
    got = solve(case["x"], case["nei"], case["k"], case["prop"], case["center"], case["disp"], tie=tie, seed=99, **kw)
    want = oracle.run(case["x"], case["nei"], case["k"], case["prop"], case["center"], case["disp"], tie=tie, seed=99,
                      **kw)
    assert got["iters"] == want["iters"] and got["status"] == want["status"]
    assert np.array_equal(got["c"], want["c"])
    assert np.array_equal(got["center"], want["center"])
    assert maxdiff(got["disp"], want["disp"]) <= TOL and maxdiff(got["prop"], want["prop"]) <= TOL


@pytest.mark.parametrize("name", case_names(files_only=True))
def test_dropin_nem_writes_reference_files(gpu_lib, tmp_path, name):
    """nem(Fname, ...) called the way ppanggolin.py:1814-1826 calls it, on the five ASCII files:
    .uf must be the reference's text byte for byte; .mf parameter lines byte for byte and the
    criteria line to print precision."""
    from pangenomenem_amd import nemfiles
    import nem as nem_module
    case = load_case(name)
    cfg = case["cfg"]
    base = nemfiles.write_nem_inputs(str(tmp_path), case["x"], case["nei"], case["prop"], case["center"], case["disp"])
    rc = nem_module.nem(Fname=base.encode(), nk=case["k"], algo=cfg["algo"].encode(), beta=cfg["beta"],
                        convergence=cfg["cvtest"].encode(), convergence_th=cfg["cvthres"], format=b"fuzzy",
                        it_max=cfg["it_max"], dolog=True, model_family=b"bern", proportion=cfg["propor"].encode(),
                        dispersion=cfg["disper"].encode(), init_mode=2)
    assert rc == case["meta"]["nem_rc"]
    if rc != 0:                                              # an emptied class (here: NaN rows from the exp overflow of
        assert rc == 1 and case["meta"]["nem_wrote"] == []   # heavy edge weights): EXIT_W_RESULT and NO result files,
        assert not os.path.exists(base + ".uf") and not os.path.exists(base + ".mf")   # nem_exe.c:624-631, 660-663
        assert "empty class" in open(base + ".stderr").read()
        return
    assert open(base + ".uf", "rb").read() == case["ref_uf"]
    got_mf = open(base + ".mf", "rb").read().split(b"\n")
    ref_mf = case["ref_mf"].split(b"\n")
    assert len(got_mf) == len(ref_mf)
    for i, (a, b) in enumerate(zip(got_mf, ref_mf)):
        if i == 2:                                           # "  U    D    L    M   error"
            ta, tb = a.split(), b.split()
            assert len(ta) == len(tb) == 5 and ta[4] == tb[4] == b"nan"
            for u, v in zip(ta[:4], tb[:4]):
                if v in (b"-inf", b"inf", b"nan", b"-nan"):  # (M = -inf with PPanGGOLiN's own edge weights)
                    assert u == v, (a, b)
                else:
                    assert_printed_g_close(u.decode(), v.decode(), (a, b))
        else:
            assert a == b, (i, a, b)
    # the reference-side parser (run_partitioning's contract) accepts our files
    labels, params, m_crit, _ = nemfiles.read_nem_outputs(str(tmp_path), case["x"].shape[1], q=case["k"])
    assert len(labels) == case["x"].shape[0]
    assert os.path.isfile(base + ".stderr") and os.path.isfile(base + ".log")
    text = open(base + ".stderr").read()
    assert ("NEM converged after %d iterations" % case["meta"]["iters"] in text) == case["meta"]["converged"]


def test_dropin_empty_class_returns_1_and_writes_nothing(gpu_lib, tmp_path):
    from pangenomenem_amd import nemfiles
    import nem as nem_module
    case = load_case("empty_class")
    base = nemfiles.write_nem_inputs(str(tmp_path), case["x"], case["nei"], case["prop"], case["center"], case["disp"])
    rc = nem_module.nem(base.encode(), 3, b"ncem", 0.5, b"clas", 1e-8, b"fuzzy", 100, True, b"bern", b"pk", b"sk_", 2)
    assert rc == 1                                            # EXIT_W_RESULT, lib_io.h:26
    assert not os.path.exists(base + ".uf") and not os.path.exists(base + ".mf")
    assert "empty class" in open(base + ".stderr").read()


def test_dropin_argument_errors(gpu_lib, tmp_path):
    from pangenomenem_amd import nemfiles
    import nem as nem_module
    case = load_case("c1_beta0_ncem_sk")
    base = nemfiles.write_nem_inputs(str(tmp_path), case["x"], None, case["prop"], case["center"], case["disp"])
    args = [base.encode(), 3, b"ncem", 0.0, b"clas", 1e-8, b"fuzzy", 100, True, b"bern", b"pk", b"sk_", 2]

    def call(**kw):
        a = list(args)
        names = ["Fname", "nk", "algo", "beta", "convergence", "convergence_th", "format", "it_max", "dolog",
                 "model_family", "proportion", "dispersion", "init_mode"]
        for k, v in kw.items():
            a[names.index(k)] = v
        return nem_module.nem(*a)

    assert call(nk=0) == 3                                    # STS_E_ARG returned raw (nem_exe.c:301)
    assert call(Fname=(base + "_missing").encode()) == 5      # STS_E_FILEIN returned raw (nem_exe.c:309)
    assert call(model_family=b"norm") == 2                    # unsupported family -> EXIT_E_ARGS
    assert call(dispersion=b"xyz") == 6                       # unknown dispersion -> EXIT_E_BUG like the reference
    assert call(init_mode=0) == 2 and call(init_mode=3) == 2  # INIT_SORT / INIT_FILE: not supported
    assert call(format=b"hard") == 0 and os.path.isfile(base + ".cf")
    labels = open(base + ".cf").read().split()
    assert len(labels) == case["x"].shape[0] and set(labels) <= {"1", "2", "3"}
    assert call(algo=b"typo") == 0                            # quirk: unknown algo runs as "nem" (nem_exe.c:371-376, 472)


@pytest.mark.parametrize("algo,disper,n,d", [("ncem", "sk_", 2048, 15), ("nem", "skd", 900, 21), ("ncem", "s__", 1500, 9)])
def test_dropin_log_file_equals_the_reference(gpu_lib, tmp_path, algo, disper, n, d):
    """<Fname>.log (dolog=1): header, criteria before and after every E-step sweep, all parameters per iteration --
    the text of the reference's own log except its date line."""
    import shutil
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("compiled reference (oracle/_ref) not present")
    from pangenomenem_amd import nemfiles, synth
    import nem as nem_module
    x, _ = synth.bernoulli_pa_matrix(n, d, n + d)
    nei = synth.contiguity_graph(n, n + d)
    prop, center, disp = synth.default_init(d)
    ours, ref = str(tmp_path / "ours"), str(tmp_path / "ref")
    base = nemfiles.write_nem_inputs(ours, x, nei, prop, center, disp)
    shutil.copytree(ours, ref)
    ref_base = os.path.join(ref, "nem_file")
    args = (3, algo.encode(), 0.5, b"clas", 1e-8, b"fuzzy", 20, True, b"bern", b"pk", disper.encode(), 2)
    assert nem_module.nem(base.encode(), *args) == 0
    assert pyoracle.Reference().nem(ref_base, *args) == 0
    from tests.util import assert_same_nem_log
    assert_same_nem_log(open(base + ".log").read(), open(ref_base + ".log").read())


@pytest.mark.parametrize("algo,disper,thres,dolog", [("ncem", "sk_", 1e-4, True), ("nem", "skd", 1e-4, True),
                                                     ("ncem", "sk_", 0.05, True), ("nem", "sk_", 1e-3, True)])
def test_dropin_crit_convergence_equals_the_reference(gpu_lib, tmp_path, algo, disper, thres, dolog):
    """nem(..., convergence="crit", ...) as a logging run (dolog, what PPanGGOLiN passes): the reference compares the
    first iteration's criterion with the initial partition's (WriteLogCrit, nem_alg.c:1980, 2398).  Same iteration
    count, same .uf byte for byte, same parameter lines of .mf as the reference's own nem() on the same files."""
    import shutil
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("compiled reference (oracle/_ref) not present")
    from pangenomenem_amd import nemfiles, synth
    import nem as nem_module
    n, d = 2048, 15
    x, _ = synth.ushaped_pa_matrix(n, d, 4)
    nei = synth.contiguity_graph(n, 4)
    prop, center, disp = synth.default_init(d)
    ours, ref = str(tmp_path / "ours"), str(tmp_path / "ref")
    base = nemfiles.write_nem_inputs(ours, x, nei, prop, center, disp)
    shutil.copytree(ours, ref)
    ref_base = os.path.join(ref, "nem_file")
    args = (3, algo.encode(), 0.5, b"crit", thres, b"fuzzy", 60, dolog, b"bern", b"pk", disper.encode(), 2)
    assert nem_module.nem(base.encode(), *args) == 0
    assert pyoracle.Reference().nem(ref_base, *args) == 0
    import re
    it_ours = re.search(r"NEM (converged|did not converge) after (\d+) iterations", open(base + ".stderr").read())
    it_ref = re.search(r"NEM (converged|did not converge) after (\d+) iterations", open(ref_base + ".stderr", errors="replace").read())
    assert it_ours and it_ref and it_ours.groups() == it_ref.groups()
    assert open(base + ".uf", "rb").read() == open(ref_base + ".uf", "rb").read()
    ours_mf, ref_mf = open(base + ".mf", "rb").read().split(b"\n"), open(ref_base + ".mf", "rb").read().split(b"\n")
    assert ours_mf[-4:] == ref_mf[-4:]


RANDOM_LOGS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "random_logs")


def _random_log_cases():
    return sorted(x for x in os.listdir(RANDOM_LOGS) if os.path.isdir(os.path.join(RANDOM_LOGS, x))) if os.path.isdir(RANDOM_LOGS) else []


@pytest.mark.parametrize("case", _random_log_cases())
def test_dropin_random_start_log_equals_the_reference_fixture(gpu_lib, tmp_path, monkeypatch, case):
    """<Fname>.log of an INIT_RANDOM run (init_mode=1, dolog=1): RandNemAlgo's "Random initialization %d :" blocks --
    line 0 with the start's drawn centres and whatever NbObs_KD the run so far left (NaN, then the last EstimPara's
    sizes, an emptied class's included), the header, a line per iteration, "Class %d empty ...", "Best start was" --
    against the text the unmodified reference wrote for the same inputs and srandom() seed
    (tests/golden/make_random_logs.py; nem_alg.c:1632-1636, 1662-1669, 1730-1732)."""
    import gzip
    import json
    from pangenomenem_amd import nemfiles, synth
    import nem as nem_module
    from tests.util import assert_same_nem_log
    here = os.path.join(RANDOM_LOGS, case)
    meta = json.load(open(os.path.join(here, "meta.json")))
    z = np.load(os.path.join(here, "inputs.npz"))
    n, d, k = int(z["n"]), int(z["d"]), int(z["k"])
    x = np.unpackbits(z["xbits"], axis=1, bitorder="little")[:, :d]
    nei = (z["nei_ptr"], z["nei_idx"], z["nei_w"])
    prop, center, disp = synth.default_init(d)
    base = nemfiles.write_nem_inputs(str(tmp_path), x, nei, prop, center, disp)
    os.remove(base + ".m")                                     # not read in this mode (nem_exe.c:513-522)
    monkeypatch.setenv("NEM_MI355X_SEED", str(meta["seed"]))
    rc = nem_module.nem(base.encode(), k, meta["algo"].encode(), meta["beta"], b"clas", 1e-8, b"fuzzy", 100, True, b"bern",
                        b"pk", meta["disper"].encode(), 1)
    assert rc == 0
    want = gzip.open(os.path.join(here, "ref_log.txt.gz"), "rt").read()
    got = open(base + ".log").read()
    assert got.count("Random initialization") == 50 and got.count(" empty at iteration ") == meta["starts_with_empty_class"]
    assert_same_nem_log(got, want)
    assert "Best start was %d " % (meta["best_start"] + 1) in open(base + ".stderr").read()
    uf = open(base + ".uf").read()
    # (that was the logged run with the starts in lock step; one start after the other: the same text, the same partition)
    monkeypatch.setenv("NEM_MI355X_BATCH_STARTS_LOGGED", "0")
    assert nem_module.nem(base.encode(), k, meta["algo"].encode(), meta["beta"], b"clas", 1e-8, b"fuzzy", 100, True, b"bern",
                          b"pk", meta["disper"].encode(), 1) == 0
    assert_same_nem_log(open(base + ".log").read(), want)
    assert open(base + ".log").read().split("\n", 1)[1] == got.split("\n", 1)[1]      # (but the date line)
    assert open(base + ".uf").read() == uf
    monkeypatch.delenv("NEM_MI355X_BATCH_STARTS_LOGGED")
    # NEM_MI355X_LOG=0: the lock-step run and a header-only log; the same partition
    monkeypatch.setenv("NEM_MI355X_LOG", "0")
    assert nem_module.nem(base.encode(), k, meta["algo"].encode(), meta["beta"], b"clas", 1e-8, b"fuzzy", 100, True, b"bern",
                          b"pk", meta["disper"].encode(), 1) == 0
    assert open(base + ".uf").read() == uf
    assert "Random initialization" not in open(base + ".log").read()


@pytest.mark.gpu
@pytest.mark.parametrize("d,seed,algo,tie", [(1069, 2, "ncem", None), (1100, 1, "ncem", None), (12, 5, "ncem", None),
                                            (12, 5, "ncem", "hash"), (40, 6, "nem", None), (40, 6, "ncem", "first")])
def test_dropin_random_start_log_in_lock_step_and_one_after_the_other(gpu_lib, tmp_path, monkeypatch, d, seed, algo, tie):
    """The per-iteration log of an INIT_RANDOM run with the 50 starts in lock step (one EM iteration per step for all of
    them, their criteria side by side, the lines handed to the writer start by start afterwards) against the same call
    with the starts one after the other: the same text but the date.  d = 1 069 / 1 100: starts draw tie-breaks behind
    their initial sweeps -- the lock-step attempt notices, hands nothing over and the call runs sequentially (the log
    must not carry a line twice); d = 12: hundreds of ties per start, all in the initial sweeps -- lock step throughout."""
    from pangenomenem_amd import nemfiles, synth
    import nem as nem_module
    n, k = (300, 3) if d > 100 else (2000, 3)
    x, _ = synth.bernoulli_pa_matrix(n, d, seed, p=(0.5, 0.5, 0.5) if d > 100 else (0.9, 0.5, 0.1))
    nei = synth.contiguity_graph(n, seed)
    prop, center, disp = synth.default_init(d)
    base = nemfiles.write_nem_inputs(str(tmp_path), x, nei, prop, center, disp)
    os.remove(base + ".m")
    monkeypatch.setenv("NEM_MI355X_SEED", str(seed))
    if tie is not None:                                        # (the stateless rules and the fuzzy algorithm: no stream to bet on)
        monkeypatch.setenv("NEM_MI355X_TIE", tie)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NEM_MI355X_BATCH_STARTS_LOGGED", mode)
        rc = nem_module.nem(base.encode(), k, algo.encode(), 0.5, b"clas", 1e-8, b"fuzzy", 12, True, b"bern", b"pk", b"sk_", 1)
        out[mode] = (rc, open(base + ".log").read().split("\n", 1)[1], open(base + ".uf").read() if rc == 0 else "",
                     open(base + ".mf").read() if rc == 0 else "")
    assert out["1"][0] == out["0"][0]
    assert out["1"][1].count("Random initialization") == 50
    assert out["1"][1] == out["0"][1]
    assert out["1"][2:] == out["0"][2:]

"""Edge cases of the hot path on the GPU against the oracle: ragged and tiny sizes (tile / wave / word boundaries),
many classes (the run-time-K kernels), wide matrices on every dispersion model, graph-less smoothing, fixed
parameters, isolated families."""
import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import assert_crit_close, maxdiff

pytestmark = pytest.mark.gpu
TOL = 1e-6


def same_run(got, want, algo):
    assert got["status"] == want["status"], (got["status"], want["status"])
    assert got["iters"] == want["iters"]
    assert got["converged"] == want["converged"]
    if want["status"] == 2:
        assert got["emptyk"] == want["emptyk"]
    assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1))
    if algo == "ncem":
        assert np.array_equal(got["c"], want["c"])
    assert maxdiff(got["c"], want["c"]) <= TOL
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, key
    assert np.array_equal(got["center"], want["center"])


def run_both(oracle, x, nei, k, prop, center, disp, **cfg):
    from pangenomenem_amd.engine import solve
    got = solve(x, nei, k, prop, center, disp, tie="hash", seed=3, **cfg)
    want = oracle.run(x, nei, k, prop, center, disp, tie="hash", seed=3, **cfg)
    same_run(got, want, cfg.get("algo", "ncem"))
    return got


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1025])
@pytest.mark.parametrize("algo", ["ncem", "nem"])
def test_ragged_family_counts(gpu_lib, oracle, n, algo):
    d = 40
    x, _ = synth.bernoulli_pa_matrix(n, d, 100 + n)
    nei = synth.contiguity_graph(n, 7) if n > 1 else None
    prop, center, disp = synth.default_init(d)
    run_both(oracle, x, nei, 3, prop, center, disp, algo=algo, beta=0.5, disper="sk_", it_max=8)


@pytest.mark.parametrize("d", [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257])
def test_ragged_organism_counts(gpu_lib, oracle, d):
    n = 700
    x, _ = synth.bernoulli_pa_matrix(n, d, 200 + d)
    nei = synth.contiguity_graph(n, 9)
    prop, center, disp = synth.default_init(d)
    for disper in ("sk_", "skd"):
        run_both(oracle, x, nei, 3, prop, center, disp, algo="ncem", beta=0.5, disper=disper, it_max=6)


@pytest.mark.parametrize("k", [11, 12, 17])
@pytest.mark.parametrize("algo", ["ncem", "nem"])
def test_many_classes_take_the_runtime_k_kernels(gpu_lib, oracle, k, algo):
    n, d = 3000, 60
    x, _ = synth.grouped_pa_matrix(n, d, 31, groups=20)
    nei = synth.contiguity_graph(n, 31)
    prop, center, disp = synth.kclass_init(x, k)
    run_both(oracle, x, nei, k, prop, center, disp, algo=algo, beta=0.3, disper="skd", it_max=5)


@pytest.mark.parametrize("disper", ["sk_", "skd", "s__", "s_d"])
@pytest.mark.parametrize("algo", ["ncem", "nem"])
def test_wide_matrix_every_dispersion_model(gpu_lib, oracle, disper, algo):
    # D > 1024: parameter update in k_finish, density in k_density (uniform and per-organism chains)
    n, d = 600, 1300
    x, _ = synth.bernoulli_pa_matrix(n, d, 77)
    nei = synth.contiguity_graph(n, 77)
    prop, center, disp = synth.default_init(d, low_disp=0.3)          # high dispersions: densities stay above underflow
    run_both(oracle, x, nei, 3, prop, center, disp, algo=algo, beta=0.5, disper=disper, it_max=4)


def test_no_graph_with_positive_beta_and_isolated_families(gpu_lib, oracle):
    n, d = 900, 25
    x, _ = synth.bernoulli_pa_matrix(n, d, 5)
    prop, center, disp = synth.default_init(d)
    run_both(oracle, x, None, 3, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", it_max=10)
    # a graph in which most families have no neighbour at all
    ptr = np.zeros(n + 1, np.int32)
    idx, w = [], []
    for i in range(0, n - 1, 50):
        idx += [i + 1]; w += [2.0]
        ptr[i + 1:] += 1
    nei = (ptr, np.array(idx, np.int32), np.array(w, np.float32))
    run_both(oracle, x, nei, 3, prop, center, disp, algo="ncem", beta=1.5, disper="sk_", it_max=10)
    run_both(oracle, x, nei, 3, prop, center, disp, algo="nem", beta=1.5, disper="skd", it_max=6)


@pytest.mark.parametrize("algo", ["ncem", "nem"])
def test_fixed_parameters(gpu_lib, oracle, algo):
    # .m flag 2: parameters are never re-estimated (nem_alg.c:1806); only the E-step iterates
    n, d = 1500, 33
    x, _ = synth.bernoulli_pa_matrix(n, d, 8)
    nei = synth.contiguity_graph(n, 8)
    prop, center, disp = synth.default_init(d)
    run_both(oracle, x, nei, 3, prop, center, disp, algo=algo, beta=0.8, disper="sk_", it_max=7, param_fix=True)


def test_all_identical_families_and_constant_columns(gpu_lib, oracle):
    n, d = 500, 48
    x = np.zeros((n, d), np.uint8)
    x[:, ::3] = 1                                                        # every family the same row
    nei = synth.contiguity_graph(n, 2)
    prop, center, disp = synth.default_init(d)
    for tie in ("hash", "first"):
        from pangenomenem_amd.engine import solve
        got = solve(x, nei, 3, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", it_max=6, tie=tie, seed=9)
        want = oracle.run(x, nei, 3, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", it_max=6, tie=tie, seed=9)
        same_run(got, want, "ncem")


@pytest.mark.parametrize("n,d,k,algo,disper,starts,seed", [
    (1200, 40, 3, "ncem", "sk_", 10, 1), (900, 25, 3, "nem", "skd", 6, 7), (2000, 64, 4, "ncem", "skd", 8, 3),
    (700, 300, 3, "ncem", "s__", 5, 5), (500, 16, 2, "nem", "s_d", 5, 11)])
def test_random_starts_match_oracle(gpu_lib, oracle, n, d, k, algo, disper, starts, seed):
    """nemgpu_run_random = the reference's init_mode INIT_RANDOM: same draws (the restated glibc generator), same
    ranking of the starts, same final M-step."""
    from pangenomenem_amd.engine import NemEngine
    x, _ = synth.bernoulli_pa_matrix(n, d, seed)
    nei = synth.contiguity_graph(n, seed)
    eng = NemEngine(n, d, k)
    eng.set_matrix(x)
    eng.set_graph(nei)
    eng.configure(algo=algo, beta=0.5, disper=disper, propor="pk", it_max=30, tie="hash", seed=seed)
    got = eng.run_random(n_starts=starts, rng_seed=seed)
    want = oracle.run_random(x, nei, k, n_starts=starts, rng_seed=seed, algo=algo, disper=disper, beta=0.5, it_max=30,
                             tie="hash", seed=seed)
    assert got["status"] == want["status"]
    assert got["best_start"] == want["best_start"]
    assert got["iters"] == want["iters"] and got["converged"] == want["converged"]
    assert np.array_equal(got["c"].argmax(1), want["c"].argmax(1))
    if algo == "ncem":
        assert np.array_equal(got["c"], want["c"])
    assert maxdiff(got["c"], want["c"]) <= TOL
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, key
    assert np.array_equal(got["center"], want["center"])
    assert_crit_close(got["crit"], want["crit"], 1e-6)
    eng.close()


@pytest.mark.parametrize("n,d,k,starts,seed", [(600, 6, 3, 6, 21), (1500, 4, 4, 5, 8)])
def test_random_starts_with_the_reference_tie_stream(gpu_lib, oracle, reference, n, d, k, starts, seed):
    """INIT_RANDOM with TieRule = TIE_RANDOM as in the reference: the starts' centre draws and the C-steps' tie draws
    are ONE random() stream (few organisms => many identical families => many ties), so every start's draws depend
    on how many ties the starts before it met.  Engine == oracle == the compiled reference, draw for draw."""
    from pangenomenem_amd.engine import NemEngine
    x, _ = synth.bernoulli_pa_matrix(n, d, seed, p=(0.9, 0.5, 0.1))
    nei = synth.contiguity_graph(n, seed)
    eng = NemEngine(n, d, k)
    eng.set_matrix(x)
    eng.set_graph(nei)
    eng.configure(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=20, tie="libc", seed=seed)
    got = eng.run_random(n_starts=starts, rng_seed=seed)
    want = oracle.run_random(x, nei, k, n_starts=starts, rng_seed=seed, algo="ncem", disper="sk_", beta=0.5, it_max=20,
                             tie="libc")
    ref = reference.classify_random(x, nei, k, n_starts=starts, rng_seed=seed, algo="ncem", disper="sk_", beta=0.5,
                                    it_max=20)
    assert got["tie_draws"] > starts * k                     # ties did draw
    for other in (want, ref):
        assert got["status"] == other["status"] and got["best_start"] == other["best_start"]
        assert np.array_equal(got["c"], other["c"])
        assert np.array_equal(got["center"], other["center"])
        for key in ("disp", "prop"):
            assert maxdiff(got[key], other[key]) <= TOL, key
    eng.close()


@pytest.mark.parametrize("d,seed,first_alone", [(1066, 3, "1"), (1068, 1, "1"), (1068, 2, "1"), (1069, 2, "1"), (1069, 2, "0"),
                                               (1100, 1, "1"), (1100, 1, "0")])
def test_random_starts_whose_iterations_draw(gpu_lib, oracle, monkeypatch, d, seed, first_alone):
    """TIE_LIBC random starts in lock step bet that a start draws only in its two initial sweeps (all classes share the
    whole sample's dispersion there; later their parameters differ).  Around d = 1068 half-and-half columns the densities
    of SOME families underflow to an all-zero row in SOME starts' iterations -- a K-way tie, a draw: the bet is lost in
    the middle of a round and the starts behind the one that drew are redone from where it left the stream (d = 1100:
    every start draws at every family, every iteration).  Engine == oracle, start for start."""
    from pangenomenem_amd.engine import NemEngine
    monkeypatch.setenv("NEM_MI355X_STARTS_FIRST_ALONE", first_alone)
    n, k, starts = 300, 3, 8
    x, _ = synth.bernoulli_pa_matrix(n, d, seed, p=(0.5, 0.5, 0.5))
    nei = synth.contiguity_graph(n, seed)
    eng = NemEngine(n, d, k)
    eng.set_matrix(x)
    eng.set_graph(nei)
    eng.configure(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=12, tie="libc", seed=seed)
    got = eng.run_random(n_starts=starts, rng_seed=seed)
    how = eng.random_start_counters()
    want = oracle.run_random(x, nei, k, n_starts=starts, rng_seed=seed, algo="ncem", disper="sk_", beta=0.5, it_max=12, tie="libc")
    assert how["in_lockstep"] + how["alone"] == starts
    if d < 1100 or first_alone == "0":
        assert how["redone"] > 0, how                             # a round did lose the bet
    else:
        assert how["alone"] == starts and how["redone"] == 0, how  # start 0 told: nobody was run on the bet
    assert got["status"] == want["status"] and got["best_start"] == want["best_start"]
    assert got["iters"] == want["iters"]
    assert np.array_equal(got["c"], want["c"])
    assert np.array_equal(got["center"], want["center"])
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, key
    assert_crit_close(got["crit"], want["crit"], 1e-6)
    eng.close()


@pytest.mark.parametrize("n,d,k,beta,disper,starts,it_max,graph", [
    (800, 12, 3, 0.0, "sk_", 7, 15, True),       # beta = 0: the sweeps read no neighbour, the draw counter still couples the sites
    (800, 12, 3, 0.5, "sk_", 7, 15, False),      # no graph at all
    (700, 10, 5, 0.7, "skd", 9, 12, True),       # K = 5, free dispersion
    (600, 8, 2, 0.5, "s__", 6, 10, True),        # K = 2, one dispersion for everything
    (500, 9, 3, 0.5, "sk_", 70, 6, True),        # more starts than a lock-step round holds (64): two rounds of phase A / B
    (900, 14, 3, 0.5, "sk_", 1, 20, True),       # one start
    (900, 14, 3, 0.5, "sk_", 5, 0, True),        # it_max = 0: initial partitions only
    (3000, 7, 4, 1.5, "s_d", 8, 10, True)])      # a strong field: long dominoes in the initial beta sweep
def test_random_starts_on_the_tie_stream_across_configurations(gpu_lib, oracle, n, d, k, beta, disper, starts, it_max, graph):
    """RandNemAlgo under TIE_LIBC in its two-phase lock-step form (initial sweeps in stream order, iterations in lock
    step) against the oracle's plain loop, start for start: few organisms => many identical families => every start
    ties at hundreds of families."""
    from pangenomenem_amd.engine import NemEngine
    seed = 1000 + n + d
    x, _ = synth.bernoulli_pa_matrix(n, d, seed, p=(0.9, 0.5, 0.1))
    nei = synth.contiguity_graph(n, seed) if graph else (np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    eng = NemEngine(n, d, k)
    eng.set_matrix(x)
    if graph:
        eng.set_graph(nei)
    eng.configure(algo="ncem", beta=beta, disper=disper, propor="pk", it_max=it_max, tie="libc", seed=seed)
    got = eng.run_random(n_starts=starts, rng_seed=seed)
    how = eng.random_start_counters()
    want = oracle.run_random(x, nei, k, n_starts=starts, rng_seed=seed, algo="ncem", disper=disper, beta=beta, it_max=it_max, tie="libc")
    assert how["in_lockstep"] + how["alone"] == starts
    assert got["tie_draws"] >= starts * k                     # (the centres' draws; all cases but the last tie on top of them)
    assert got["status"] == want["status"] and got["best_start"] == want["best_start"] and got["iters"] == want["iters"]
    assert np.array_equal(got["c"], want["c"])
    assert np.array_equal(got["center"], want["center"])
    for key in ("disp", "prop"):
        assert maxdiff(got[key], want[key]) <= TOL, key
    assert_crit_close(got["crit"], want["crit"], 1e-6)
    eng.close()


@pytest.mark.parametrize("n,d,beta,init_mode,stops", [(3000, 5, 0.5, "pipelined", True), (3000, 5, 1.5, "pipelined", True),
                                                       (6000, 4, 1.0, "pipelined", False), (3000, 5, 1.5, "host", False)])
def test_a_tie_stream_start_that_needs_more_rounds_than_were_enqueued(gpu_lib, oracle, monkeypatch, n, d, beta, init_mode, stops):
    """Under TIE_LIBC a run's two initial sweeps ride in its first pipelined batch with a fixed number of relaxation rounds
    (verified by the loop control on the device).  With 4-5 organisms thousands of families are identical: equal
    dispersions at the start make whole blocks tie, a draw's number depends on every tie before it, and the beta sweep
    needs far more rounds than were enqueued -- the batch stops itself, books nothing, and the start is done again from
    the host (third case: 4 583 draws, but through in the rounds enqueued).  Result = oracle either way, and with
    NEM_MI355X_LIBC_INIT=host, the form without the device check."""
    from pangenomenem_amd.engine import NemEngine
    monkeypatch.setenv("NEM_MI355X_LIBC_INIT", init_mode)
    seed = 31
    x, _ = synth.bernoulli_pa_matrix(n, d, seed, p=(0.9, 0.5, 0.1))
    nei = synth.contiguity_graph(n, seed)
    prop = np.full(3, 1.0 / 3, np.float32)
    center = x[[0, n // 2, n - 1]].astype(np.float32)              # three families as centres, one dispersion: the random starts' shape
    if len({tuple(r) for r in center}) < 3:
        center = np.array([[0] * d, [1] * d, [1] + [0] * (d - 1)], np.float32)
    disp = np.full((3, d), 0.12, np.float32)
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp)
    eng.configure(algo="ncem", beta=beta, disper="sk_", propor="pk", it_max=12, tie="libc", seed=seed)
    got = eng.run()
    want = oracle.run(x, nei, 3, prop, center, disp, algo="ncem", beta=beta, disper="sk_", it_max=12, tie="libc", seed=seed)
    assert got["tie_draws"] > 100, got["tie_draws"]
    if stops:
        assert eng.graph_counters()["host_finished_sweeps"] >= 1      # the first batch did stop itself
    if init_mode == "host":
        assert eng.graph_counters()["host_finished_sweeps"] == 0
    same_run(got, want, "ncem")
    got2 = eng.run()                                                   # (the rounds enqueued have adapted or not: same result)
    same_run(got2, want, "ncem")
    eng.close()


def test_very_wide_matrix_and_maximum_class_count(gpu_lib, oracle):
    # D > 32768: the class masks of the uniform chain no longer fit its LDS staging, every class takes the general
    # chain; K = 32 is the engine's maximum
    n, d = 300, 33000
    x, _ = synth.bernoulli_pa_matrix(n, d, 99)
    nei = synth.contiguity_graph(n, 99)
    prop, center, disp = synth.default_init(d, low_disp=0.45)
    run_both(oracle, x, nei, 3, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", it_max=2)
    n, d, k = 4000, 40, 32
    x, _ = synth.grouped_pa_matrix(n, d, 98, groups=32)
    nei = synth.contiguity_graph(n, 98)
    prop, center, disp = synth.kclass_init(x, k)
    for algo in ("ncem", "nem"):
        run_both(oracle, x, nei, k, prop, center, disp, algo=algo, beta=0.4, disper="skd", it_max=3)
    from pangenomenem_amd.engine import NemEngine, NemGpuError
    with pytest.raises(NemGpuError):
        NemEngine(100, 10, 33)


@pytest.mark.parametrize("algo,disper,tie,n,d,k,starts", [
    ("ncem", "sk_", "hash", 3000, 40, 3, 12), ("nem", "skd", "hash", 1500, 25, 3, 7), ("ncem", "skd", "libc", 2500, 30, 4, 10),
    ("ncem", "sk_", "libc", 600, 6, 3, 9)])
def test_lockstep_random_starts_equal_sequential_ones(gpu_lib, monkeypatch, algo, disper, tie, n, d, k, starts):
    """RandNemAlgo's starts in lock step (one launch per EM step for all starts, state-only twins of the engine) against
    the same starts one after the other: best start, its labels, parameters, criteria, draws -- identical.  The last
    case ties heavily under TIE_LIBC: every start's position in the stream depends on the ties before it."""
    from pangenomenem_amd.engine import NemEngine
    x, _ = synth.bernoulli_pa_matrix(n, d, 77, p=(0.9, 0.5, 0.1))
    nei = synth.contiguity_graph(n, 77)
    out = []
    for mode in ("0", "1"):
        monkeypatch.setenv("NEM_MI355X_BATCH_STARTS", mode)
        eng = NemEngine(n, d, k)
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.configure(algo=algo, beta=0.5, disper=disper, propor="pk", it_max=25, tie=tie, seed=5)
        out.append(eng.run_random(n_starts=starts, rng_seed=5))
        out.append(eng.run_random(n_starts=starts, rng_seed=5))      # the twins are reused
        eng.close()
    ref = out[0]
    for other in out[1:]:
        assert other["best_start"] == ref["best_start"] and other["status"] == ref["status"]
        assert other["iters"] == ref["iters"] and other["tie_draws"] == ref["tie_draws"]
        for key in ("c", "center", "disp", "prop", "crit"):
            assert np.array_equal(other[key], ref[key], equal_nan=True), key


def test_the_starts_beta_sweep_gets_two_rounds_once_that_was_enough_and_three_again_when_it_is_not(gpu_lib, oracle):
    """NCEM under a stateless tie rule: a pipelined start enqueues three relaxation rounds for its beta sweep until three
    starts in a row were through in two, then two (the third launch only found out that it had nothing to do).  A start
    that does need the third round afterwards stops its batch, is finished from the host, and the count goes back up for
    good.  Easy starts first (the same problem four times), then starts with a strong field from poor parameters: every
    run equals the oracle's, and the hard ones did go through the host at least once."""
    from pangenomenem_amd.engine import NemEngine
    n, d = 6000, 40
    x, _ = synth.bernoulli_pa_matrix(n, d, 5)
    nei = synth.contiguity_graph(n, 5)
    prop, center, disp = synth.default_init(d)
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x); eng.set_graph(nei)
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", propor="pk", it_max=6, tie="hash", seed=3)
    want = oracle.run(x, nei, 3, prop, center, disp, **cfg)
    for rep in range(5):
        eng.set_params(prop, center, disp)
        eng.configure(**cfg)
        same_run(eng.run(), want, "ncem")
    before = eng.graph_counters()["host_finished_sweeps"]
    rng = np.random.default_rng(1)
    for rep in range(6):
        # poor parameters and a strong field: label changes run down the path, the beta sweep of the start needs more rounds
        center2 = (rng.random((3, d)) < 0.5).astype(np.float32)
        disp2 = np.full((3, d), 0.45, np.float32)
        cfg2 = dict(cfg, beta=float(1.0 + rep))
        eng.set_params(prop, center2, disp2)
        eng.configure(**cfg2)
        same_run(eng.run(), oracle.run(x, nei, 3, prop, center2, disp2, **cfg2), "ncem")
    assert eng.graph_counters()["host_finished_sweeps"] > before
    eng.set_params(prop, center, disp)
    eng.configure(**cfg)
    same_run(eng.run(), want, "ncem")
    eng.close()


def test_one_engine_through_many_different_starts(gpu_lib, oracle):
    """One engine, sixty runs in a row that differ in parameters, field strength, tie rule and iteration cap: what the engine
    learns from a run about the rounds to enqueue (a start's beta sweep, the tie stream's two initial sweeps, the first
    iterations' third round) is only ever a guess about the next one -- every run must equal the oracle's whatever the
    guess was."""
    from pangenomenem_amd.engine import NemEngine
    n, d = 4000, 24
    x, _ = synth.bernoulli_pa_matrix(n, d, 9, p=(0.9, 0.5, 0.1))
    nei = synth.contiguity_graph(n, 9)
    eng = NemEngine(n, d, 3)
    eng.set_matrix(x); eng.set_graph(nei)
    rng = np.random.default_rng(4)
    prop0, center0, disp0 = synth.default_init(d)
    for rep in range(60):
        easy = rep % 5 != 4
        if easy:
            prop, center, disp = prop0, center0, disp0
        else:                                                  # two classes alike, poor dispersions: ties and long sweeps
            prop = np.array([0.3, 0.3, 0.4], np.float32)
            center = (rng.random((3, d)) < 0.5).astype(np.float32); center[1] = center[0]
            disp = np.full((3, d), np.float32(rng.choice([0.2, 0.45])), np.float32)
        cfg = dict(algo="ncem", beta=float(rng.choice([0.0, 0.5, 1.0, 2.5])), disper=str(rng.choice(["sk_", "skd"])), propor="pk",
                   it_max=int(rng.choice([0, 1, 5, 30])), tie=str(rng.choice(["hash", "libc", "first"])), seed=int(rng.integers(0, 99)))
        eng.set_params(prop, center, disp)
        eng.configure(**cfg)
        got = eng.run()
        want = oracle.run(x, nei, 3, prop, center, disp, **cfg)
        same_run(got, want, "ncem")
    eng.close()

"""Several engines at once from worker threads of one process (pangenomenem_amd/batch.py):
same answers as one at a time, through the in-memory API and through the drop-in nem()."""
import numpy as np
import pytest

from pangenomenem_amd import nemfiles, synth

pytestmark = pytest.mark.gpu


def _problems(count, n=3000, d=120):
    out = []
    for p in range(count):
        x, _ = synth.bernoulli_pa_matrix(n + 17 * p, d + p, 50 + p)          # different shapes on purpose
        nei = synth.contiguity_graph(n + 17 * p, 50 + p)
        prop, center, disp = synth.default_init(d + p)
        out.append((x, nei, 3, prop, center, disp))
    return out


def test_concurrent_engines_give_the_single_engine_answers(gpu_lib, oracle):
    from pangenomenem_amd.batch import solve_many
    probs = _problems(12)
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=4)
    alone = solve_many(probs, workers=1, **cfg)
    crowd = solve_many(probs, workers=6, **cfg)
    for a, b, p in zip(alone, crowd, probs):
        assert a["iters"] == b["iters"] and a["status"] == b["status"]
        assert np.array_equal(a["c"], b["c"]) and np.array_equal(a["disp"], b["disp"])
    want = oracle.run(*probs[3], **cfg)
    assert np.array_equal(crowd[3]["c"], want["c"]) and crowd[3]["iters"] == want["iters"]


def test_solve_many_groups_bit_rows_and_problems_without_a_graph(gpu_lib):
    """nemgpu_solve_many: groups smaller than the batch (builders run ahead of the lock-step runs), bit rows instead
    of bytes, a problem without neighbours -- every answer is the problem's own solve()."""
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.engine import solve
    probs = _problems(11)
    probs[4] = (probs[4][0], None) + probs[4][2:]
    cfg = dict(algo="ncem", beta=0.5, disper="skd", tie="hash", seed=9)
    want = [solve(*p, **cfg) for p in probs]
    as_bits = []
    for x, nei, k, prop, center, disp in probs:
        b = np.packbits(x, axis=1, bitorder="little")
        b = np.pad(b, ((0, 0), (0, (-b.shape[1]) % 4)))
        as_bits.append((np.ascontiguousarray(b).view(np.uint32), nei, k, prop, center, disp))
    for batch in (solve_many(probs, workers=3, group=4, **cfg), solve_many(as_bits, workers=5, group=3, **cfg),
                  solve_many(probs[:1], workers=2, group=8, **cfg)):
        for got, w in zip(batch, want):
            assert got["iters"] == w["iters"] and got["status"] == w["status"] and got["converged"] == w["converged"]
            for f in ("c", "prop", "center", "disp", "nbobs_k", "crit"):
                assert np.array_equal(got[f], w[f]), f


def test_pooled_resources_and_captured_batches_are_reused_without_changing_results(gpu_lib):
    """Streams, device / pinned blocks and the lock-step contexts (slabs + captured graphs) of destroyed engines go to
    a per-device pool: later groups of the same shape replay the graphs earlier ones captured.  Same answers on every
    pass, also after the pool has been emptied."""
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.engine import load_library
    nei = synth.contiguity_graph(4000, 5)
    prop, center, disp = synth.default_init(96)
    probs = [(synth.ushaped_pa_matrix(4000, 96, 5 + p)[0], nei, 3, prop, center, disp) for p in range(24)]
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=2)
    first = solve_many(probs, workers=4, group=6, **cfg)           # 4 groups of one shape: plain, captured, replayed
    again = solve_many(probs, workers=4, group=6, **cfg)
    load_library().nemgpu_release_cached()
    fresh = solve_many(probs, workers=2, group=6, **cfg)
    assert len({r["iters"] for r in first}) > 1                      # (members of a group stop at different iterations)
    for a, b, c in zip(first, again, fresh):
        assert a["iters"] == b["iters"] == c["iters"]
        for f in ("c", "prop", "center", "disp", "crit"):
            assert np.array_equal(a[f], b[f]) and np.array_equal(a[f], c[f]), f


def test_without_the_resource_pool(gpu_lib, tmp_path):
    """NEM_MI355X_POOL_MB=0: nothing is kept, every engine allocates and frees for itself -- same answers."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from pangenomenem_amd import synth\n"
        "from pangenomenem_amd.batch import solve_many\n"
        "from pangenomenem_amd.engine import solve\n"
        "nei = synth.contiguity_graph(1500, 3)\n"
        "prop, center, disp = synth.default_init(70)\n"
        "probs = [(synth.ushaped_pa_matrix(1500, 70, 40 + p)[0], nei, 3, prop, center, disp) for p in range(6)]\n"
        "cfg = dict(algo='ncem', beta=0.5, disper='sk_', tie='hash', seed=1)\n"
        "a = solve_many(probs, workers=3, group=2, **cfg)\n"
        "b = [solve(*p, **cfg) for p in probs]\n"
        "assert all(np.array_equal(x['c'], y['c']) and x['iters'] == y['iters'] for x, y in zip(a, b))\n"
        "print('ok', sum(x['iters'] for x in a))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mb in ("0", "8192"):
        env = dict(os.environ, NEM_MI355X_POOL_MB=mb, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and outs[0].startswith("ok")


def test_solve_many_reports_the_problem_that_is_wrong(gpu_lib):
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.engine import NemGpuError
    probs = _problems(5, n=800, d=40)
    bad = probs[2][0].copy()
    bad[5, 7] = 3                                               # not a presence/absence value
    probs[2] = (bad,) + probs[2][1:]
    with pytest.raises(NemGpuError, match="0/1"):
        solve_many(probs, workers=2, group=2, algo="ncem", beta=0.5)
    assert len(solve_many(probs[:2], workers=2, group=2, algo="ncem", beta=0.5)) == 2    # (the library is fine afterwards)


def test_concurrent_dropin_calls(gpu_lib, tmp_path):
    from pangenomenem_amd.batch import nem_many
    probs = _problems(8, n=1500, d=60)
    calls = []
    for p, (x, nei, k, prop, center, disp) in enumerate(probs):
        base = nemfiles.write_nem_inputs(str(tmp_path / str(p)), x, nei, prop, center, disp)
        calls.append(dict(Fname=base.encode(), nk=3, algo=b"ncem", beta=0.5, convergence=b"clas", convergence_th=1e-8,
                          format=b"fuzzy", it_max=100, dolog=True, model_family=b"bern", proportion=b"pk",
                          dispersion=b"sk_", init_mode=2))
    assert nem_many(calls, workers=1) == [0] * len(calls)
    alone = [open(c["Fname"].decode() + ".uf", "rb").read() for c in calls]
    alone_mf = [open(c["Fname"].decode() + ".mf", "rb").read() for c in calls]
    assert nem_many(calls, workers=8) == [0] * len(calls)
    for c, uf, mf in zip(calls, alone, alone_mf):
        assert open(c["Fname"].decode() + ".uf", "rb").read() == uf
        assert open(c["Fname"].decode() + ".mf", "rb").read() == mf


@pytest.mark.parametrize("free_dispersion", [False, True])
def test_run_partitioning_arrays_equals_the_file_route(gpu_lib, tmp_path, free_dispersion):
    """The in-memory counterpart of PPanGGOLiN's run_partitioning (f1) returns what parsing nem()'s files returns."""
    import nem as nem_module
    from pangenomenem_amd.partitioning import run_partitioning_arrays
    n, d = 4000, 90
    x, _ = synth.bernoulli_pa_matrix(n, d, 71)
    nei = synth.contiguity_graph(n, 71)
    prop, center, disp = synth.default_init(d)
    base = nemfiles.write_nem_inputs(str(tmp_path), x, nei, prop, center, disp)
    rc = nem_module.nem(base.encode(), 3, b"ncem", 0.5, b"clas", 1e-8, b"fuzzy", 100, True, b"bern", b"pk",
                        b"skd" if free_dispersion else b"sk_", 2)
    assert rc == 0
    labels, params, _, _ = nemfiles.read_nem_outputs(str(tmp_path), d, q=3)
    got_labels, got_params = run_partitioning_arrays(x, nei, 0.5, free_dispersion)
    assert [got_labels["fam%d" % (i + 1)] for i in range(n)] == labels
    assert set(got_labels.values()) <= {"P", "S", "C"}
    for k in range(3):
        assert got_params[k][0] == params[k][0]                                  # mu as booleans
        assert np.allclose(got_params[k][1], params[k][1], rtol=0, atol=1e-5)    # epsilon through the .mf text
        assert abs(got_params[k][2] - params[k][2]) <= 5.1e-4                   # pi is printed with 3 decimals


def test_lockstep_batch_is_bit_identical_to_solo_runs(gpu_lib, oracle):
    """nemgpu_run_many: problems of different sizes, class counts, algorithms' worth of kernels in ONE launch per EM
    step -- every member's labels, parameters, criteria, iteration count equal its solo run bit for bit."""
    from pangenomenem_amd.engine import NemEngine, run_many
    probs = _problems(9)
    for algo, disper, tie in (("ncem", "sk_", "hash"), ("ncem", "skd", "libc"), ("nem", "sk_", "hash")):
        cfg = dict(algo=algo, beta=0.5, disper=disper, tie=tie, seed=4, it_max=12 if algo == "nem" else 100)
        solo = []
        engines = []
        for (x, nei, k, prop, center, disp) in probs:
            eng = NemEngine(x.shape[0], x.shape[1], k)
            eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp); eng.configure(**cfg)
            solo.append(eng.run())
            engines.append(eng)
        many = run_many(engines)
        again = run_many(engines[::-1])[::-1]                 # another order, another lead
        for a, b, c in zip(solo, many, again):
            for other in (b, c):
                assert a["iters"] == other["iters"] and a["status"] == other["status"] and a["converged"] == other["converged"]
                # (relaxation-round counts may differ: the buffers a run starts from are whatever the run before left,
                #  and a sweep's rounds depend on its first guess -- its fixed point does not)
                assert a["tie_draws"] == other["tie_draws"]
                for key in ("c", "center", "disp", "prop", "nbobs_k", "crit"):
                    assert np.array_equal(a[key], other[key], equal_nan=True), (algo, key)
        for e in engines:
            e.close()
    want = oracle.run(*probs[5], algo="nem", beta=0.5, disper="sk_", tie="hash", seed=4, it_max=12)
    assert want["iters"] == many[5]["iters"] and np.array_equal(many[5]["c"].argmax(1), want["c"].argmax(1))


def test_lockstep_batch_with_members_that_stop_early(gpu_lib):
    """Members converge at different iterations, one has an empty class (stops at its first M-step), one gets
    it_max = 0: the others go on, everybody's result is its solo result."""
    from pangenomenem_amd.engine import NemEngine, run_many
    probs = _problems(5, n=2500, d=40)
    engines, solo = [], []
    for p, (x, nei, k, prop, center, disp) in enumerate(probs):
        if p == 1:
            x = np.ones_like(x)                               # every family everywhere: classes 2 and 3 empty
        eng = NemEngine(x.shape[0], x.shape[1], k)
        eng.set_matrix(x); eng.set_graph(nei); eng.set_params(prop, center, disp)
        eng.configure(algo="ncem", beta=0.5, disper="sk_", it_max=0 if p == 3 else 100, tie="hash", seed=p)
        solo.append(eng.run())
        engines.append(eng)
    many = run_many(engines)
    assert solo[1]["status"] == 2 and solo[3]["iters"] == 0
    for a, b in zip(solo, many):
        assert a["iters"] == b["iters"] and a["status"] == b["status"] and a["emptyk"] == b["emptyk"]
        for key in ("c", "center", "disp", "prop"):
            assert np.array_equal(a[key], b[key]), key
    for e in engines:
        e.close()


def test_solve_many_over_a_device_list_equals_one_device(gpu_lib):
    """nemgpu_solve_many_devices: the lock-step groups dealt round-robin over a device list (here twice the one GPU of
    the box: two pipelines, each with its own worker threads, streams and lock-step contexts) -- every problem's
    result is bit-identical to the single-device call and to the problem solved alone."""
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.engine import solve
    probs = _problems(13)
    probs[6] = (probs[6][0], None) + probs[6][2:]
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=5)
    one = solve_many(probs, workers=4, group=3, **cfg)
    for devices in ([0, 0], [0, 0, 0], [0]):
        many = solve_many(probs, workers=6, group=3, devices=devices, **cfg)
        for a, b in zip(one, many):
            assert a["iters"] == b["iters"] and a["status"] == b["status"] and a["converged"] == b["converged"]
            for f in ("c", "prop", "center", "disp", "nbobs_k", "crit"):
                assert np.array_equal(a[f], b[f]), (devices, f)
    alone = solve(*probs[9], **cfg)
    assert np.array_equal(one[9]["c"], alone["c"]) and one[9]["iters"] == alone["iters"]
    from pangenomenem_amd.engine import NemGpuError
    with pytest.raises(NemGpuError, match="device"):
        solve_many(probs[:2], devices=[0, 99], **cfg)


def test_solve_many_fuzzy_memberships_come_back_with_the_group(gpu_lib):
    """NEM (fuzzy) problems of different shapes in one group: the memberships (n x k floats per member) and the parameter
    blocks reach the host in the group's one copy -- every member's are its own solve()'s."""
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.engine import solve
    probs = _problems(7, n=2500, d=90)
    probs[2] = (probs[2][0], None) + probs[2][2:]
    cfg = dict(algo="nem", beta=0.5, disper="sk_")
    want = [solve(*p, **cfg) for p in probs]
    for batch in (solve_many(probs, workers=4, group=4, **cfg), solve_many(probs, workers=2, group=7, **cfg)):
        for got, w in zip(batch, want):
            assert got["iters"] == w["iters"] and got["status"] == w["status"]
            for f in ("c", "prop", "center", "disp", "nbobs_k", "crit"):
                assert np.array_equal(got[f], w[f]), f


def test_solve_many_with_two_runners_on_the_device(gpu_lib):
    """four groups and more with six workers and more: nemgpu_solve_many deals the groups to two runners on the one
    device (the device list names it twice) -- every answer is still the problem's own solve()."""
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.engine import solve
    probs = _problems(14, n=1800, d=70)
    cfg = dict(algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=3)
    want = [solve(*p, **cfg) for p in probs]
    for batch in (solve_many(probs, workers=6, group=3, **cfg), solve_many(probs, workers=8, group=2, devices=[0, 0, 0], **cfg)):
        for got, w in zip(batch, want):
            assert got["iters"] == w["iters"] and got["status"] == w["status"]
            for f in ("c", "prop", "center", "disp", "nbobs_k", "crit"):
                assert np.array_equal(got[f], w[f]), f

"""The drop-in `nem()` at BASELINE configs[1] size, called the way ppanggolin.py:1814-1826 calls it, next to the
compiled reference's own `nem()` on the same five ASCII files (oracle/_ref travels to the GPU box).  Checks the
files byte for byte and records both whole-call times (parse + EM + write) under gpurun_out/ when it can."""
import json
import os
import shutil
import time

import numpy as np
import pytest

from pangenomenem_amd import nemfiles, synth
from tests.util import assert_printed_g_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _call(fn, base, k):
    t0 = time.perf_counter()
    rc = fn(base.encode(), k, b"ncem", 0.5, b"clas", 1e-8, b"fuzzy", 100, True, b"bern", b"pk", b"sk_", 2)
    return rc, time.perf_counter() - t0


def test_dropin_whole_call_at_configs1(gpu_lib, tmp_path):
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("compiled reference (oracle/_ref) not present")
    import nem as nem_module
    cfg = synth.make_config("C2")
    ours_dir, ref_dir = str(tmp_path / "ours"), str(tmp_path / "ref")
    t0 = time.perf_counter()
    base = nemfiles.write_nem_inputs(ours_dir, cfg["x"], cfg["nei"], cfg["prop"], cfg["center"], cfg["disp"])
    t_write = time.perf_counter() - t0
    shutil.copytree(ours_dir, ref_dir)
    ref_base = os.path.join(ref_dir, "nem_file")

    _call(nem_module.nem, base, 3)                            # first call: library load, context creation
    rc, t_ours = _call(nem_module.nem, base, 3)
    assert rc == 0
    ref = pyoracle.Reference()
    rc_ref, t_ref = _call(ref.nem, ref_base, 3)
    assert rc_ref == 0

    # tie-free data: the reference's time-seeded random() never fires, the files must agree
    assert open(base + ".uf", "rb").read() == open(ref_base + ".uf", "rb").read()
    got, want = open(base + ".mf", "rb").read().split(b"\n"), open(ref_base + ".mf", "rb").read().split(b"\n")
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        if i == 2:                                            # criteria line: print precision, device exp/log
            ta, tb = a.split(), b.split()
            assert len(ta) == len(tb)
            for u, v in zip(ta[:4], tb[:4]):
                assert_printed_g_close(u.decode(), v.decode(), (a, b))
        else:
            assert a == b, (i, a[:80], b[:80])
    labels, params, _, _ = nemfiles.read_nem_outputs(ours_dir, cfg["x"].shape[1], q=3)
    assert len(labels) == cfg["x"].shape[0]
    # <Fname>.log: the per-iteration log (criteria before / after each E-step, all parameters), line for line
    # except the date in the first one
    from tests.util import assert_same_nem_log
    assert_same_nem_log(open(base + ".log").read(), open(ref_base + ".log").read())

    rec = dict(workload="BASELINE configs[1] files: 20000 x 500, K=3, beta=0.5, ncem/sk_/pk, dolog=1",
               input_bytes=sum(os.path.getsize(base + e) for e in (".str", ".dat", ".nei", ".m")),
               reference_nem_whole_call_s=t_ref, this_library_nem_whole_call_s=t_ours,
               speedup_whole_call=t_ref / t_ours, python_input_write_s=t_write, host_cores_used=1,
               phases=[l.strip() for l in open(base + ".stderr").read().splitlines() if "[engine]" in l])
    print(json.dumps(rec))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "dropin_whole_call.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass
    assert t_ours < t_ref


def test_dropin_random_starts_equal_the_reference_given_its_seed(gpu_lib, tmp_path, oracle):
    """init_mode = 1 (INIT_RANDOM), what partition_shell uses (ppanggolin.py:1207): 50 random starts.  The reference
    seeds with time(NULL); with the seed fixed (NEM_MI355X_SEED here, srandom() in the reference harness) and
    tie-free data both sides draw the same centres, so the partition and the parameters must agree."""
    from oracle import pyoracle
    import nem as nem_module
    n, d, k, seed = 1500, 30, 3, 4242
    x, _ = synth.bernoulli_pa_matrix(n, d, 17)
    nei = synth.contiguity_graph(n, 17)
    prop, center, disp = synth.default_init(d)
    base = nemfiles.write_nem_inputs(str(tmp_path), x, nei, prop, center, disp)
    os.remove(base + ".m")                                     # not read in this mode (nem_exe.c:513-522)
    os.environ["NEM_MI355X_SEED"] = str(seed)
    try:
        rc = nem_module.nem(base.encode(), k, b"ncem", 0.5, b"clas", 1e-8, b"fuzzy", 100, True, b"bern", b"pk", b"sk_", 1)
    finally:
        del os.environ["NEM_MI355X_SEED"]
    assert rc == 0
    labels, params, _, _ = nemfiles.read_nem_outputs(str(tmp_path), d, q=k, init="random")
    want = oracle.run_random(x, nei, k, n_starts=50, rng_seed=seed, algo="ncem", disper="sk_", beta=0.5, it_max=100,
                             tie="hash", seed=seed)
    got_c = np.loadtxt(base + ".uf", dtype=np.float32).reshape(n, k)
    assert np.array_equal(got_c.argmax(1), want["c"].argmax(1))
    assert "Best start was %d" % (want["best_start"] + 1) in open(base + ".stderr").read()
    if pyoracle.have_reference():
        ref = pyoracle.Reference().classify_random(x, nei, k, n_starts=50, rng_seed=seed, algo="ncem", disper="sk_",
                                                   beta=0.5, it_max=100)
        assert ref["status"] == 0 and ref["best_start"] == want["best_start"]
        assert np.array_equal(got_c.argmax(1), ref["c"].argmax(1))

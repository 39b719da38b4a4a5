"""Robustness of the library's boundary on the GPU: inputs it must sanitise, faults it must report, device selection."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pangenomenem_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pack_bits(x):
    n, d = x.shape
    wf = (d + 31) // 32
    pad = np.zeros((n, wf * 32), np.uint8)
    pad[:, :d] = x
    return (pad.reshape(n, wf, 32).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(axis=2).astype(np.uint32)


@pytest.mark.parametrize("d", [33, 500])
@pytest.mark.parametrize("algo", ["ncem", "nem"])
def test_dirty_padding_bits_of_bit_rows_are_ignored(gpu_lib, oracle, d, algo):
    """Bits above organism d-1 in a row's last word are not data (include/nem_mi355x.h): rows whose padding is all ones
    give the same run as clean rows and as the byte matrix -- no phantom organisms in the popcount M-step, no popcount
    beyond d in the density kernels' lane ordering."""
    from pangenomenem_amd.engine import NemEngine
    n, k = 3000, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 11)
    nei = synth.contiguity_graph(n, 11)
    prop, center, disp = synth.default_init(d)
    clean = pack_bits(x)
    dirty = clean.copy()
    dirty[:, -1] |= np.uint32(0xFFFFFFFF) << np.uint32(d & 31)
    assert not np.array_equal(clean, dirty)
    outs = []
    for bits in (clean, dirty):
        eng = NemEngine(n, d, k)
        eng.set_matrix_bits(bits)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        eng.configure(algo=algo, beta=0.5, disper="sk_", it_max=6, tie="hash", seed=3)
        outs.append(eng.run())
        eng.close()
    assert np.array_equal(dirty[:, -1] >> np.uint32(d & 31), np.full(n, 0xFFFFFFFF >> (d & 31), np.uint32))   # caller's buffer untouched
    for key in ("c", "center", "disp", "prop", "nbobs_k", "crit"):
        assert np.array_equal(outs[0][key], outs[1][key], equal_nan=True), key
    assert outs[0]["iters"] == outs[1]["iters"]
    want = oracle.run(x, nei, k, prop, center, disp, algo=algo, beta=0.5, disper="sk_", it_max=6, tie="hash", seed=3)
    assert np.array_equal(outs[1]["c"].argmax(1), want["c"].argmax(1))
    assert np.array_equal(outs[1]["center"], want["center"])


def test_dirty_padding_through_solve_many(gpu_lib):
    from pangenomenem_amd import batch
    n, d, k = 1500, 70, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 5)
    nei = synth.contiguity_graph(n, 5)
    prop, center, disp = synth.default_init(d)
    clean = pack_bits(x)
    dirty = clean.copy()
    dirty[:, -1] |= np.uint32(0xFFFFFFFF) << np.uint32(d & 31)
    res = batch.solve_many([(clean, nei, k, prop, center, disp), (dirty, nei, k, prop, center, disp)],
                           algo="ncem", beta=0.5, disper="sk_", it_max=8, tie="hash", seed=1)
    for key in ("c", "center", "disp", "prop"):
        assert np.array_equal(res[0][key], res[1][key]), key


_FAULT_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from pangenomenem_amd import synth
from pangenomenem_amd.engine import NemEngine, NemGpuError
n, d, k = 2500, 96, 3
x, _ = synth.bernoulli_pa_matrix(n, d, 3)
prop, center, disp = synth.default_init(d)
eng = NemEngine(n, d, k)
eng.set_matrix(x); eng.set_graph(synth.contiguity_graph(n, 3)); eng.set_params(prop, center, disp)
eng.configure(algo="nem", beta=0.5, disper="sk_", it_max=4)
mode = sys.argv[1]
try:
    if mode == "mstep":
        eng.init_partition()
        eng.mstep()
    else:
        eng.run()
    print("NOFAULT")
except NemGpuError as exc:
    print("FAULT", str(exc))
"""


@pytest.mark.parametrize("mode", ["mstep", "run"])
def test_a_stalled_handover_of_the_fuzzy_mstep_is_an_error_not_a_result(gpu_lib, tmp_path, mode):
    """k_mstep_fuzzy_pc bounds its spins; a spin that runs out raises FLAG_FAULT and the call returns
    NEMGPU_E_INTERNAL (status 10) instead of sums that are silently wrong.  NEM_MI355X_FAULT_INJECT=fuzzy_pc makes one
    producer wave skip a hand-over (own process: the variable is read when an engine is created)."""
    script = tmp_path / "fault.py"
    script.write_text(_FAULT_SCRIPT % dict(root=ROOT))
    env = dict(os.environ, NEM_MI355X_FAULT_INJECT="fuzzy_pc")
    out = subprocess.run([sys.executable, str(script), mode], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "FAULT" in out.stdout and "NOFAULT" not in out.stdout, out.stdout
    assert "status 10" in out.stdout and "fault" in out.stdout
    env.pop("NEM_MI355X_FAULT_INJECT")
    out = subprocess.run([sys.executable, str(script), mode], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "NOFAULT" in out.stdout, (out.stdout, out.stderr[-2000:])


def test_dropin_reports_a_kernel_fault_as_an_internal_error(gpu_lib, tmp_path):
    """nem() maps NEMGPU_E_INTERNAL to 6 (EXIT_E_BUG, lib_io.h:22-34) and writes no result files."""
    from pangenomenem_amd import nemfiles
    n, d, k = 2500, 96, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 3)
    base = str(tmp_path / "nem_file")
    prop, center, disp = synth.default_init(d)
    nemfiles.write_nem_inputs(str(tmp_path), x, synth.contiguity_graph(n, 3), prop, center, disp)
    code = ("import sys; sys.path.insert(0, %r); import nem; "
            "print('RC', nem.nem(Fname=%r.encode(), nk=3, algo=b'nem', beta=0.5, convergence=b'clas', convergence_th=1e-8, "
            "format=b'fuzzy', it_max=3, dolog=True, model_family=b'bern', proportion=b'pk', dispersion=b'sk_', init_mode=2))"
            % (ROOT, base))
    env = dict(os.environ, NEM_MI355X_FAULT_INJECT="fuzzy_pc", NEM_MI355X_LOG="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "RC 6" in out.stdout, (out.stdout, out.stderr[-2000:])
    assert not os.path.exists(base + ".uf") and not os.path.exists(base + ".mf")


def test_default_device_selection(gpu_lib, monkeypatch):
    from pangenomenem_amd import engine
    monkeypatch.delenv("NEM_MI355X_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert engine.default_device() == 0                    # the current HIP device of a process that chose none
    monkeypatch.setenv("LOCAL_RANK", "3")                  # ... a launcher's LOCAL_RANK when there is one
    assert engine.default_device() == 3 % engine.device_count()
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setenv("NEM_MI355X_DEVICE", "0")
    assert engine.default_device() == 0
    monkeypatch.setenv("NEM_MI355X_DEVICE", "auto")
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert engine.default_device() == 5 % engine.device_count()
    for bad in ("banana", "-1", "9999", "1x", " "):
        monkeypatch.setenv("NEM_MI355X_DEVICE", bad)
        with pytest.raises(engine.NemGpuError, match="NEM_MI355X_DEVICE"):
            engine.default_device()


def test_sharded_path_rejects_the_criterion_convergence_test(gpu_lib):
    from pangenomenem_amd.engine import NemEngine, NemGpuError
    n, d, k = 512, 40, 3
    x, _ = synth.bernoulli_pa_matrix(n, d, 1)
    prop, center, disp = synth.default_init(d)
    from pangenomenem_amd.distributed import slot_layout
    blk, stride = slot_layout(n, 1, k + k * d)
    eng = NemEngine(stride, d, k, site_lo=0, site_hi=n)
    eng.set_matrix(x); eng.set_graph(None); eng.set_params(prop, center, disp)
    eng.shard_layout(1, 0, blk, stride, n)
    for cv in ("crit", "crit_logged"):
        eng.configure(algo="ncem", beta=0.0, cvtest=cv, tie="hash")
        with pytest.raises(NemGpuError, match="none and clas"):
            eng.shard_begin()
    eng.close()

"""Python-side pins: dicts captured from the REAL run_partitioning (ppanggolin.py:1761-1980, run in
the build container by tests/golden/make_pypins.py on top of the compiled reference nem()).  Here, without a GPU:
the repo's restatement of its parsing contract (pangenomenem_amd/nemfiles.read_nem_outputs) applied to the
reference's own .uf / .mf text, and partition_dicts applied to the CPU oracle's full-precision run, must both
reproduce them.  The GPU counterparts are in tests/test_gpu_pypins.py."""
import gzip
import json
import os
import shutil

import numpy as np
import pytest

from pangenomenem_amd import nemfiles, synth
from pangenomenem_amd.partitioning import partition_dicts

PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pypins")


def pin_names():
    with open(os.path.join(PINS, "manifest.json")) as f:
        return json.load(f)


def load_pin(name):
    d = os.path.join(PINS, name)
    with open(os.path.join(d, "pin.json")) as f:
        pin = json.load(f)
    inp = np.load(os.path.join(d, "inputs.npz"))
    n, dd = int(inp["n"]), int(inp["d"])
    pin["x"] = np.unpackbits(inp["xbits"], axis=1, bitorder="little")[:, :dd].astype(np.uint8)
    pin["nei"] = (inp["nei_ptr"], inp["nei_idx"], inp["nei_w"]) if bool(inp["has_graph"]) else None
    for ext in ("uf", "mf"):
        p = os.path.join(d, "ref_%s.txt.gz" % ext)
        pin["ref_" + ext] = gzip.open(p, "rb").read() if os.path.isfile(p) else None
    assert pin["x"].shape == (n, dd)
    return pin


def assert_matches_pin(pin, labels, params):
    """labels: list in family order; params: {k: (mu bools, eps floats, pi)} -- against the pinned dicts.  mu exact;
    epsilon through the .mf's %10g (6 significant digits); pi through its %5.3g (3 significant digits)."""
    assert "".join(str(v) for v in labels) == pin["labels"]
    assert sorted(int(k) for k in pin["params"]) == sorted(params)
    for k, want in pin["params"].items():
        mu, eps, pi = params[int(k)]
        assert [bool(v) for v in mu] == want["mu"]
        assert np.allclose(eps, want["epsilon"], rtol=6e-6, atol=1e-12)
        assert abs(pi - want["proportion"]) <= 5.1e-4


@pytest.mark.parametrize("name", pin_names())
def test_output_parser_reproduces_run_partitioning(name, tmp_path):
    pin = load_pin(name)
    if not pin["has_outputs"]:
        # nem() wrote nothing (empty class): run_partitioning's IOError branch, every family 'U', no parameters
        with pytest.raises(IOError):
            nemfiles.read_nem_outputs(str(tmp_path), pin["d"])
        assert pin["labels"] == "U" * pin["n"] and pin["params"] == {}
        return
    for ext in ("uf", "mf"):
        with open(os.path.join(str(tmp_path), "nem_file." + ext), "wb") as f:
            f.write(pin["ref_" + ext])
    labels, params, m_crit, bic = nemfiles.read_nem_outputs(str(tmp_path), pin["d"])
    assert_matches_pin(pin, labels, params)
    assert np.isfinite(m_crit) and np.isfinite(bic)


@pytest.mark.parametrize("name", pin_names())
def test_oracle_run_gives_run_partitioning_dicts(name, oracle):
    pin = load_pin(name)
    d = pin["d"]
    prop, center, disp = synth.default_init(d)
    res = oracle.run(pin["x"], pin["nei"], 3, prop, center, disp, algo="ncem", beta=pin["beta"],
                     disper="skd" if pin["free_dispersion"] else "sk_", tie="hash", seed=1)
    names = ["fam%d" % (i + 1) for i in range(pin["n"])]
    labels, params = partition_dicts(res, names)
    if not pin["has_outputs"]:
        assert res["status"] != 0 and set(labels.values()) == {"U"} and params == {}
        return
    assert_matches_pin(pin, [labels[nm] for nm in names], params)


def test_pins_describe_their_inputs():
    import hashlib
    for name in pin_names():
        pin = load_pin(name)
        h = hashlib.sha256(np.packbits(pin["x"], axis=1, bitorder="little").tobytes()).hexdigest()
        assert h == pin["x_sha256"]

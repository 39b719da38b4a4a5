"""GPU side of the Python-side pins (tests/test_pypins.py): the drop-in nem() on the five ASCII files followed by
the run_partitioning parsing contract, and the in-memory run_partitioning_arrays, both reproduce the dicts the REAL
run_partitioning returned on the same inputs (tests/golden/pypins, made by tests/golden/make_pypins.py)."""
import numpy as np
import pytest

from pangenomenem_amd import nemfiles, synth
from tests.test_pypins import assert_matches_pin, load_pin, pin_names

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", pin_names())
def test_dropin_nem_then_parsing_contract(gpu_lib, tmp_path, name):
    import nem as nem_module
    pin = load_pin(name)
    d = pin["d"]
    prop, center, disp = synth.default_init(d)
    base = nemfiles.write_nem_inputs(str(tmp_path), pin["x"], pin["nei"], prop, center, disp)
    rc = nem_module.nem(Fname=base.encode(), nk=3, algo=b"ncem", beta=pin["beta"], convergence=b"clas",
                        convergence_th=1e-8, format=b"fuzzy", it_max=100, dolog=True, model_family=b"bern",
                        proportion=b"pk", dispersion=b"skd" if pin["free_dispersion"] else b"sk_", init_mode=2)
    if not pin["has_outputs"]:
        assert rc == 1                                         # EXIT_W_RESULT: empty class, no files
        with pytest.raises(IOError):
            nemfiles.read_nem_outputs(str(tmp_path), d)
        return
    assert rc == 0
    labels, params, _, _ = nemfiles.read_nem_outputs(str(tmp_path), d)
    assert_matches_pin(pin, labels, params)
    # the .uf text itself is the reference's, byte for byte
    assert open(base + ".uf", "rb").read() == pin["ref_uf"]


@pytest.mark.parametrize("name", pin_names())
def test_run_partitioning_arrays_reproduces_run_partitioning(gpu_lib, name):
    from pangenomenem_amd.partitioning import run_partitioning_arrays
    pin = load_pin(name)
    labels, params = run_partitioning_arrays(pin["x"], pin["nei"], pin["beta"], pin["free_dispersion"])
    names = ["fam%d" % (i + 1) for i in range(pin["n"])]
    if not pin["has_outputs"]:
        assert set(labels.values()) == {"U"} and params == {}
        return
    assert_matches_pin(pin, [labels[nm] for nm in names], params)

"""nemgpu_solve_chunks -- the samples of partition()'s voting loop (ppanggolin.py:1045-1086) formed ON THE DEVICE from one
resident master pangenome -- against the same samples formed on the host (chunks.form_chunk_host, the reference's
__write_nem_input_files recipe, CPU-tested in tests/test_chunks_host.py) and solved by nemgpu_solve_many: the kept
families, every label, every parameter, every iteration count identical; and against the oracle on one of them."""
import numpy as np
import pytest

from pangenomenem_amd import synth
from tests.util import maxdiff

pytestmark = pytest.mark.gpu


def samples(d, dc, count, seed):
    rng = np.random.default_rng(seed)
    return [rng.permutation(d)[:dc] for _ in range(count)]


@pytest.mark.parametrize("n,d,dc,count,disper,group", [(3000, 200, 50, 12, "sk_", 5), (5000, 96, 96, 3, "sk_", 32), (2500, 320, 33, 9, "skd", 4),
                                                       (20000, 1000, 500, 6, "sk_", 3)])
def test_device_formed_chunks_equal_host_formed_ones(gpu_lib, n, d, dc, count, disper, group):
    from pangenomenem_amd.batch import solve_many
    from pangenomenem_amd.chunks import Master, form_chunk_host
    x, (ptr, idx), eb = synth.master_pangenome(n, d, 7)
    subs = samples(d, dc, count, 3)
    cfg = dict(algo="ncem", beta=0.5, disper=disper, it_max=30, tie="hash", seed=2)
    m = Master(x, ptr, idx, eb)
    got = m.solve_chunks(subs, workers=4, group=group, **cfg)
    m.close()
    prop, center, disp = synth.default_init(dc)
    host = [form_chunk_host(x, ptr, idx, eb, s) for s in subs]
    want = solve_many([(xc, nei, 3, prop, center, disp) for xc, nei, _ in host], workers=4, group=group, **cfg)
    dropped = 0
    for g, w, (xc, nei, fam) in zip(got, want, host):
        assert np.array_equal(g["families"], fam)
        assert g["n"] == len(fam) and g["nnz"] == int(nei[0][-1])
        assert g["status"] == w["status"] and g["iters"] == w["iters"] and g["converged"] == w["converged"]
        assert np.array_equal(g["labels"], w["c"].argmax(1))
        for key in ("prop", "center", "disp", "nbobs_k"):
            assert np.array_equal(g[key], w[key]), key
        assert np.array_equal(g["crit"], w["crit"], equal_nan=True)
        dropped += n - len(fam)
    assert dropped > 0 or dc == d


def test_a_device_formed_chunk_against_the_oracle(gpu_lib, oracle):
    from pangenomenem_amd.chunks import Master, form_chunk_host
    n, d, dc = 4000, 150, 60
    x, (ptr, idx), eb = synth.master_pangenome(n, d, 11)
    sub = samples(d, dc, 1, 5)[0]
    m = Master(x, ptr, idx, eb)
    got = m.solve_chunks([sub], workers=1, group=1, algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=4)[0]
    m.close()
    xc, nei, fam = form_chunk_host(x, ptr, idx, eb, sub)
    prop, center, disp = synth.default_init(dc)
    want = oracle.run(xc, nei, 3, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", tie="hash", seed=4)
    assert got["iters"] == want["iters"] and got["status"] == want["status"]
    assert np.array_equal(got["labels"], want["c"].argmax(1))
    assert np.array_equal(got["center"], want["center"]) and maxdiff(got["disp"], want["disp"]) <= 1e-6


@pytest.mark.parametrize("dc,beta", [(6, 0.5), (9, 1.0), (40, 0.5)])
def test_device_formed_chunks_under_the_reference_tie_rule(gpu_lib, oracle, dc, beta):
    """The drop-in's own tie rule (TIE_LIBC, every problem its own srandom(seed) stream) through nemgpu_solve_chunks.  Two
    classes start with the same centre, dispersion and proportion: they tie wherever they win -- about 1 800 draws per
    sample, most in the two initial sweeps (which ride in the first lock-step batch, verified on the device, and are
    done again from the host when they need more rounds than were enqueued).  Each sample against the oracle on the
    host-formed problem, and the lock-step groups against one sample at a time."""
    from pangenomenem_amd.chunks import Master, form_chunk_host
    n, d = 3000, 120
    x, (ptr, idx), eb = synth.master_pangenome(n, d, 21)
    subs = samples(d, dc, 7, 8)
    cfg = dict(algo="ncem", beta=beta, disper="sk_", it_max=25, tie="libc", seed=6)
    start = dict(center_k=(1.0, 1.0, 0.0), disp_k=(0.1, 0.1, 0.1))
    m = Master(x, ptr, idx, eb)
    got = m.solve_chunks(subs, workers=2, group=4, **start, **cfg)
    alone = [m.solve_chunks([s], workers=1, group=1, **start, **cfg)[0] for s in subs]
    m.close()
    p0 = np.float32(0.33333)
    prop = np.array([p0, p0, np.float32(np.float32(np.float32(1.0) - p0) - p0)], np.float32)
    center = np.stack([np.ones(dc), np.ones(dc), np.zeros(dc)]).astype(np.float32)
    disp = np.full((3, dc), 0.1, np.float32)
    for g, a, sub in zip(got, alone, subs):
        xc, nei, fam = form_chunk_host(x, ptr, idx, eb, sub)
        want = oracle.run(xc, nei, 3, prop, center, disp, **cfg)
        assert np.array_equal(g["families"], fam)
        assert g["status"] == want["status"] and g["iters"] == want["iters"] and g["converged"] == want["converged"]
        assert np.array_equal(g["labels"], want["c"].argmax(1))
        assert np.array_equal(g["center"], want["center"]) and maxdiff(g["disp"], want["disp"]) <= 1e-6 and maxdiff(g["prop"], want["prop"]) <= 1e-6
        for key in ("labels", "prop", "center", "disp", "nbobs_k", "crit"):
            assert np.array_equal(g[key], a[key], equal_nan=True), key


def test_bad_samples_are_refused(gpu_lib):
    from pangenomenem_amd.chunks import Master
    from pangenomenem_amd.engine import NemGpuError
    x, (ptr, idx), eb = synth.master_pangenome(500, 40, 1)
    m = Master(x, ptr, idx, eb)
    with pytest.raises(NemGpuError, match="out of range"):
        m.solve_chunks([[0, 1, 40]])
    with pytest.raises(NemGpuError):
        m.solve_chunks([[]])
    x0 = x.copy(); x0[:, 5] = 0                               # an organism nobody lives in: a sample of it holds no family
    m2 = Master(x0, ptr, idx, eb)
    with pytest.raises(NemGpuError, match="no family"):
        m2.solve_chunks([[5]])
    m.close(); m2.close()

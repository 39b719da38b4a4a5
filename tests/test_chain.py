"""The exact segmented float chains (pangenomenem_amd/csrc/nem_chain.hpp): acc = (float)((double)acc + x_i) evaluated
as integer prefix sums inside float binades must equal the plain sequential loop bit for bit, on inputs that hit
every exit of the integer form (binade changes, exact and near ties, shrinking sums, zeros of both signs, huge and
tiny values, non-finite values).  CPU: the host emulation of the device procedure; GPU: the device procedure."""
import ctypes as C

import numpy as np
import pytest

from pangenomenem_amd import build as nem_build


def _lib():
    lib = C.CDLL(nem_build.build())
    lib.nemgpu_chain_host.restype = C.c_float
    lib.nemgpu_chain_host.argtypes = [C.c_void_p, C.c_longlong, C.c_float, C.c_int]
    lib.nemgpu_chain_device.restype = C.c_int
    lib.nemgpu_chain_device.argtypes = [C.c_void_p, C.c_longlong, C.c_float, C.c_int, C.POINTER(C.c_float)]
    return lib


def cases():
    rng = np.random.Generator(np.random.PCG64(7))
    out = []
    # what the criteria sum: negative logs of a few hundred, all of one sign
    out.append(("log-like", -rng.uniform(150, 400, 30000), 0.0))
    out.append(("log-like float", (-rng.uniform(150, 400, 30000)).astype(np.float32).astype(np.float64), 0.0))
    # -0.0 entries (the additive identity k_crit_terms stores for skipped terms) between real ones
    x = -rng.uniform(1, 9, 20000); x[rng.random(20000) < 0.6] = -0.0
    out.append(("with -0.0", x, 0.0))
    # exact ties on every grid: multiples of 1/2^k added to a growing sum, odd and even partners
    out.append(("halves", rng.integers(1, 2000, 50000) * 0.5, 0.0))
    out.append(("quarter steps", rng.integers(1, 64, 50000) * 0.25, 3.0))
    out.append(("ones from 2^24-5", np.ones(5000), float(2 ** 24 - 5)))
    out.append(("threes", np.full(70000, 3.0), 1.0))
    # near ties: a tie plus / minus something far below the float grid (decided by the double rounding)
    x = rng.integers(1, 99, 20000) * 0.5 + rng.choice([0.0, 2.0 ** -40, -2.0 ** -40, 2.0 ** -27, -2.0 ** -27], 20000)
    out.append(("near ties", x, 2.0 ** 20))
    # mixed signs (the sum shrinks and changes sign), wide dynamic range
    out.append(("mixed", rng.normal(0, 1, 20000) * 10.0 ** rng.integers(-8, 8, 20000), 0.0))
    out.append(("mixed small", rng.normal(0, 1e-3, 9000), 1.0))
    # tiny values against a large accumulator, values above the accumulator, subnormals
    out.append(("tiny", rng.uniform(0, 1e-12, 10000), 1000.0))
    out.append(("huge first", np.concatenate([[1e30], rng.uniform(0, 1e24, 5000)]), 0.0))
    out.append(("subnormal", rng.uniform(0, 1e-40, 5000), 0.0))
    out.append(("negative start", -rng.uniform(0, 5, 8000), -7.25))
    # non-finite values and overflow
    x = rng.uniform(0, 1, 3000); x[1500] = np.inf
    out.append(("inf", x, 0.0))
    x = rng.uniform(0, 1, 3000); x[700] = np.nan
    out.append(("nan", x, 0.0))
    out.append(("overflow", np.full(4000, 3e38), 0.0))
    # an accumulator that is infinite (or NaN) for most of the chain: the steps that change nothing are skipped
    out.append(("all -inf", np.full(20000, -np.inf), 0.0))                               # (every criterion at 200 000 x 5 000)
    x = rng.uniform(-1, 1, 30000); x[3] = -np.inf
    out.append(("-inf then finite", x, 0.0))
    x = rng.uniform(-1, 1, 30000); x[3] = -np.inf; x[17000] = np.inf
    out.append(("-inf then +inf", x, 0.0))
    x = rng.uniform(-1, 1, 30000); x[3] = np.inf; x[29999] = np.nan
    out.append(("+inf then nan", x, 0.0))
    x = rng.uniform(-1, 1, 30000); x[5000] = np.inf; x[5001] = np.inf; x[9000] = -np.inf
    out.append(("+inf +inf -inf", x, 2.5))
    out.append(("inf accumulator", rng.uniform(-1, 1, 9000), np.inf))
    out.append(("nan accumulator", rng.uniform(-1, 1, 9000), np.nan))
    x = rng.uniform(-1, 1, 9000); x[4096] = -np.inf
    out.append(("-inf accumulator meets -inf", x, -np.inf))
    # sizes around the window / chunk boundaries
    for n in (0, 1, 2, 3, 4, 5, 63, 64, 65, 4095, 4096, 4097, 8192, 12289):
        out.append(("n=%d" % n, rng.uniform(0.5, 1.5, n), 0.0))
    return out


CASES = cases()


def same(a, b):
    return np.float32(a).tobytes() == np.float32(b).tobytes() or (np.isnan(a) and np.isnan(b))


def numpy_chain(x, init):
    acc = np.float32(init)
    for v in x:
        acc = np.float32(np.float64(acc) + v)
    return acc


@pytest.mark.parametrize("name,x,init", CASES, ids=[c[0] for c in CASES])
def test_host_emulation_equals_the_sequential_loop(name, x, init):
    lib = _lib()
    x = np.ascontiguousarray(x, np.float64)
    with np.errstate(all="ignore"):
        want = lib.nemgpu_chain_host(x.ctypes.data, len(x), init, 0)
        got = lib.nemgpu_chain_host(x.ctypes.data, len(x), init, 1)
        if len(x) <= 8192:                                  # the C loop itself against numpy's float32 / float64
            assert same(want, numpy_chain(x, init)), name
    assert same(got, want), (name, got, want)


def test_host_emulation_random_soak():
    lib = _lib()
    rng = np.random.Generator(np.random.PCG64(11))
    for t in range(300):
        n = int(rng.integers(1, 6000))
        kind = t % 4
        if kind == 0:
            x = rng.integers(0, 1 << int(rng.integers(1, 20)), n) * 2.0 ** int(rng.integers(-6, 3))
        elif kind == 1:
            x = -rng.uniform(0, 10.0 ** int(rng.integers(-3, 6)), n)
        elif kind == 2:
            x = rng.normal(0, 1, n) * 2.0 ** rng.integers(-30, 30, n)
        else:
            x = (rng.uniform(0, 700, n)).astype(np.float32).astype(np.float64)
        init = float(np.float32(rng.choice([0.0, 1.0, -3.5, 2.0 ** 23, 1e-30])))
        x = np.ascontiguousarray(x, np.float64)
        want = lib.nemgpu_chain_host(x.ctypes.data, n, init, 0)
        got = lib.nemgpu_chain_host(x.ctypes.data, n, init, 1)
        assert same(got, want), (t, n, init)


@pytest.mark.gpu
@pytest.mark.parametrize("name,x,init", CASES, ids=[c[0] for c in CASES])
def test_device_procedure_equals_the_sequential_loop(gpu_lib, name, x, init):
    lib = _lib()
    x = np.ascontiguousarray(x, np.float64)
    want = lib.nemgpu_chain_host(x.ctypes.data, len(x), init, 0)
    got = C.c_float(0)
    assert lib.nemgpu_chain_device(x.ctypes.data, len(x), init, 0, C.byref(got)) == 0
    assert same(got.value, want), (name, got.value, want)


@pytest.mark.gpu
def test_device_procedure_random_soak(gpu_lib):
    lib = _lib()
    rng = np.random.Generator(np.random.PCG64(12))
    for t in range(120):
        n = int(rng.integers(1, 40000))
        kind = t % 3
        if kind == 0:
            x = rng.integers(0, 1 << int(rng.integers(1, 20)), n) * 2.0 ** int(rng.integers(-6, 3))
        elif kind == 1:
            x = -rng.uniform(0, 10.0 ** int(rng.integers(-3, 6)), n)
        else:
            x = rng.normal(0, 1, n) * 2.0 ** rng.integers(-30, 30, n)
        init = float(np.float32(rng.choice([0.0, 1.0, -3.5, 2.0 ** 23, 1e-30])))
        x = np.ascontiguousarray(x, np.float64)
        want = lib.nemgpu_chain_host(x.ctypes.data, n, init, 0)
        got = C.c_float(0)
        assert lib.nemgpu_chain_device(x.ctypes.data, n, init, 0, C.byref(got)) == 0
        assert same(got.value, want), (t, n, init)

"""InerToDispK_'s d-ordered float sum of a class's inertia values (reference nem_mod.c:1054-1058) evaluated in pieces
(pangenomenem_amd/csrc/nem_halfsum.hpp): the host emulation of the device procedure -- same walk, same piece
application, lane by lane -- against the plain float loop on non-negative multiples of 1/2, on inputs chosen to hit
every branch (ties on every grid and both parities, values that vanish against the sum, runs of zeros and of tiny
addends that keep the sum inside an uncertain zone, sums that stay exact, sums that pass many powers of two); and,
with a GPU, the device procedure itself."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from pangenomenem_amd import build, engine
    build.build()
    lib = engine.load_library()
    lib.nemgpu_halfsum_host.restype = C.c_float
    lib.nemgpu_halfsum_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.nemgpu_halfsum_device.restype = C.c_int
    lib.nemgpu_halfsum_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    return lib


def numpy_chain(x):
    s = np.float32(0)
    for v in x:
        s = np.float32(s + v)
    return s


def cases():
    rng = np.random.Generator(np.random.PCG64(20261004))
    out = []

    def add(name, x):
        out.append((name, np.ascontiguousarray(x, np.float32)))

    # class counts of a 200 000 x 5 000 NCEM iteration: min(s0, s1) and N_k / 2 entries, totals near 2^29
    for t in range(6):
        n = int(rng.integers(3000, 8192))
        nk = int(rng.integers(20000, 200000))
        s1 = rng.integers(0, nk + 1, size=n)
        x = np.minimum(s1, nk - s1).astype(np.float64)
        half = rng.random(n) < 0.02
        x[half] = nk / 2.0
        add("counts_%d" % t, x)
    # every value odd (a tie at the first rounding level for every element), then every value = 1 mod 4, 2 mod 4, ...
    for mod, rem in ((2, 1), (4, 2), (4, 1), (8, 4), (16, 8), (32, 16), (64, 32)):
        n = 6000
        x = (rng.integers(1 << 12, 1 << 18, size=n) // mod) * mod + rem
        add("ties_%d_%d" % (mod, rem), x.astype(np.float64))
        add("ties_half_%d_%d" % (mod, rem), x.astype(np.float64) / 2.0)
    # small sums: never round (level 0 only), and sums that only just pass 2^23
    add("exact_small", rng.integers(0, 2000, size=4000) / 2.0)
    add("just_past", np.concatenate([np.full(1, 2.0 ** 23 - 3), rng.integers(0, 8, size=500) / 2.0]))
    # addends that vanish against the sum or sit exactly on half a spacing
    top = 2.0 ** 24 - 2.0
    add("vanishing", np.concatenate([[top, top, top], np.full(3000, 1.0), np.full(3000, 2.5), np.full(2000, 4.0)]))
    add("half_spacing", np.concatenate([[top, top], np.full(5000, 2.0)]))            # spacing 4 from 2^25 on: every add an exact tie
    add("half_spacing_odd", np.concatenate([[top, top, 8.0], np.full(5000, 2.0)]))   # ... from an odd significand
    # long runs of zeros and of tiny addends around a power of two (the walk cannot tell the level there)
    add("zeros_around", np.concatenate([rng.integers(60000, 70000, size=256), np.zeros(700), np.full(300, 0.5),
                                        rng.integers(0, 3, size=2000) / 2.0, rng.integers(50000, 90000, size=3000)]))
    add("all_zero", np.zeros(777))
    add("one_value", np.array([12345.5]))
    add("empty_tail", np.concatenate([rng.integers(0, 1 << 23, size=70), np.zeros(0)]))
    # values up to just below 2^24, few and many
    add("huge_few", rng.integers(1 << 22, 1 << 24, size=65) - 0.5)
    add("huge_many", rng.integers(1 << 22, 1 << 24, size=8000) - 0.5)
    # random mixtures of magnitudes, lengths around the lane boundaries
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 1000, 4999, 5000, 5001, 8191, 8192):
        scale = int(rng.integers(4, 24))
        x = rng.integers(0, 1 << scale, size=n) / 2.0
        x[rng.random(n) < 0.3] = 0.0
        add("mix_%d" % n, x)
    return out


CASES = cases()


@pytest.mark.parametrize("waves", [16, 1, 5])
@pytest.mark.parametrize("name,x", CASES, ids=[c[0] for c in CASES])
def test_piecewise_chain_equals_the_plain_loop(lib, name, x, waves):
    assert np.all(x >= 0) and np.all(x * 2 == np.floor(x * 2)) and np.all(x < 2.0 ** 24)
    stepped = C.c_int(0)
    want = lib.nemgpu_halfsum_host(x.ctypes.data, len(x), 0, None)
    got = lib.nemgpu_halfsum_host(x.ctypes.data, len(x), waves, C.byref(stepped))
    assert np.float32(want).view(np.uint32) == np.float32(got).view(np.uint32), (name, want, got)
    if len(x) <= 6000 and waves == 16:                   # (the C loop itself against numpy's float32 arithmetic)
        assert np.float32(want) == numpy_chain(x)
    assert 0 <= stepped.value <= len(x)


def test_counts_like_chains_are_mostly_maps(lib):
    """the case the kernel meets (200 000 x 5 000: class counts, totals near 2^29): only the ranges that hold a power of
    two the sum passes -- a thread's five elements each -- are added for real, the rest are integer maps"""
    rng = np.random.Generator(np.random.PCG64(5))
    nk = 70000
    s1 = rng.integers(0, nk + 1, size=5000)
    x = np.ascontiguousarray(np.minimum(s1, nk - s1), np.float32)
    stepped = C.c_int(0)
    got = lib.nemgpu_halfsum_host(x.ctypes.data, len(x), 16, C.byref(stepped))
    assert got == lib.nemgpu_halfsum_host(x.ctypes.data, len(x), 0, None)
    assert stepped.value <= 8 * 5, stepped.value


def test_random_chains_fuzz(lib):
    rng = np.random.Generator(np.random.PCG64(99))
    for t in range(400):
        n = int(rng.integers(1, 8193))
        scale = int(rng.integers(1, 25))
        x = rng.integers(0, 1 << scale, size=n).astype(np.float64) / 2.0
        if t % 3 == 0:
            x[rng.random(n) < rng.random()] = 0.0
        if t % 5 == 0:
            x = np.floor(x)                              # integers only
        x = np.ascontiguousarray(x, np.float32)
        want = lib.nemgpu_halfsum_host(x.ctypes.data, n, 0, None)
        got = lib.nemgpu_halfsum_host(x.ctypes.data, n, (16, 1, 3, 8)[t % 4], None)
        assert np.float32(want).view(np.uint32) == np.float32(got).view(np.uint32), (t, n, scale, want, got)


@pytest.mark.gpu
@pytest.mark.parametrize("waves", [16, 1])
@pytest.mark.parametrize("name,x", CASES, ids=[c[0] for c in CASES])
def test_device_procedure_equals_the_plain_loop(gpu_lib, lib, name, x, waves):
    out = C.c_float(0)
    assert lib.nemgpu_halfsum_device(x.ctypes.data, len(x), waves, 0, C.byref(out)) == 0
    want = lib.nemgpu_halfsum_host(x.ctypes.data, len(x), 0, None)
    assert np.float32(want).view(np.uint32) == np.float32(out.value).view(np.uint32), (name, want, out.value)


@pytest.mark.gpu
def test_device_procedure_fuzz(gpu_lib, lib):
    rng = np.random.Generator(np.random.PCG64(7))
    for t in range(150):
        n = int(rng.integers(1, 8193))
        scale = int(rng.integers(1, 25))
        x = rng.integers(0, 1 << scale, size=n).astype(np.float64) / 2.0
        if t % 3 == 0:
            x[rng.random(n) < rng.random()] = 0.0
        x = np.ascontiguousarray(x, np.float32)
        out = C.c_float(0)
        assert lib.nemgpu_halfsum_device(x.ctypes.data, n, (16, 1)[t % 2], 0, C.byref(out)) == 0
        want = lib.nemgpu_halfsum_host(x.ctypes.data, n, 0, None)
        assert np.float32(want).view(np.uint32) == np.float32(out.value).view(np.uint32), (t, n, scale)

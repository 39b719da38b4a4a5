import numpy as np


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if a.dtype == np.float64:
        return np.array_equal(a.view(np.uint64), b.view(np.uint64))
    return np.array_equal(a, b)


def maxdiff(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b)
    d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
    d = np.where(np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b)), 0.0, d)
    return float(np.nanmax(d)) if d.size else 0.0


def assert_crit_close(got, want, tol=1e-6, what=None):
    """The six criteria (D, G, U, M, L, Z as floats).  Where the reference's value is finite the engine's is within
    `tol` relative; where it is NOT -- M and Z are -inf as soon as beta*sum(w) passes 88 on one site (the float exp of
    nem_alg.c:2740-2751), everything is NaN once a NaN row entered the partition -- the engine's value must be the SAME
    non-finite value: NaN for NaN, an infinity of the same sign for an infinity.  (Round 3 masked those entries.)"""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got, want, what)
    for j, (g, w) in enumerate(zip(got.ravel(), want.ravel())):
        if np.isnan(w):
            assert np.isnan(g), (j, got, want, what)
        elif np.isinf(w):
            assert np.isinf(g) and (g > 0) == (w > 0), (j, got, want, what)
        else:
            assert np.isfinite(g) and abs(g - w) <= tol * max(1.0, abs(w)), (j, got, want, what)


def ulp_diff64(a, b):
    """max distance in units in the last place between two float64 arrays (finite entries)."""
    a = np.ascontiguousarray(a, np.float64).view(np.int64)
    b = np.ascontiguousarray(b, np.float64).view(np.int64)
    return int(np.max(np.abs(a - b))) if a.size else 0


def random_fuzzy_partition(n, k, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.random((n, k)).astype(np.float32) ** 3
    c /= c.sum(axis=1, keepdims=True)
    return c.astype(np.float32)


def random_hard_partition(n, k, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    lab = rng.integers(0, k, size=n)
    c = np.zeros((n, k), np.float32)
    c[np.arange(n), lab] = 1.0
    return c


def assert_same_nem_log(ours_text, ref_text):
    """<Fname>.log against the reference's: every line identical except the date in the first one; the four criteria
    columns of an iteration line (%5.0f of i-ordered float sums that pass through the device's exp/log) may differ by
    one unit in the last printed digit."""
    a, b = ours_text.split("\n"), ref_text.split("\n")
    assert a[0].startswith("NEM log file  -  ") and b[0].startswith("NEM log file  -  ")
    assert len(a) == len(b) and len(a) > 6, (len(a), len(b))
    for i, (u, v) in enumerate(zip(a[1:], b[1:])):
        if u == v:
            continue
        if u.startswith("Best start was ") and v.startswith("Best start was "):      # RandNemAlgo's last line: "%d (U = %g)"
            assert u.split("(")[0] == v.split("(")[0], (u, v)
            fu, fv = float(u.split("=")[1].strip(" )")), float(v.split("=")[1].strip(" )"))
            assert abs(fu - fv) <= 2e-5 * abs(fv), (u, v)
            continue
        tu, tv = u.split(), v.split()
        assert len(tu) == len(tv), (i + 1, u[:100], v[:100])
        for p, (s, t) in enumerate(zip(tu, tv)):
            if s != t:
                assert p in (1, 2, 4, 5) and abs(float(s) - float(t)) <= 1.0, (i + 1, p, s, t)


def assert_printed_g_close(u, v, what=None):
    """Two numbers printed with %g (six significant digits): the same text, or one unit apart in the last digit %g
    can print for that value.  The engine-level criteria are held to 1e-6 relative (tests/test_gpu_parity.py); a
    difference there can move the sixth printed digit by one and no more."""
    import math
    if u == v:
        return
    fu, fv = float(u), float(v)
    if math.isnan(fu) and math.isnan(fv):
        return
    assert math.isfinite(fu) and math.isfinite(fv) and fv != 0.0, (u, v, what)
    unit = 10.0 ** (math.floor(math.log10(abs(fv))) - 5)
    assert abs(fu - fv) <= 1.0000001 * unit, (u, v, what)

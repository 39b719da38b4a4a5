"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports every symbol that
include/*.h declares, fails loudly without a GPU, and its host-only file layer reproduces the
reference's text formats (no kernel is launched in this file)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.golden_util import case_names, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pangenomenem_amd import build, engine
    build.build()
    return engine.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nem_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(nemgpu_[a-z0-9_]+|nemio_[a-z0-9_]+|nem)\s*\(", text))
    return sorted(names)


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert "nem" in names and "nemgpu_run" in names and len(names) > 25
    for name in names:
        assert hasattr(lib, name), "symbol %s declared in include/nem_mi355x.h is not exported" % name


def test_reference_signature_of_nem(lib):
    """13 arguments, same order and C types as nem_exe.h:23-35."""
    text = open(os.path.join(ROOT, "include", "nem_mi355x.h")).read()
    m = re.search(r"int nem\((.*?)\);", text, flags=re.S)
    args = [a.strip() for a in m.group(1).split(",")]
    assert [a.split()[-1] for a in args] == ["Fname", "nk", "algo", "beta", "convergence", "convergence_th", "format",
                                            "it_max", "dolog", "model_family", "proportion", "dispersion", "init_mode"]
    assert [" ".join(a.split()[:-1]) for a in args] == ["const char*", "const int", "const char*", "const float",
                                                        "const char*", "const float", "const char*", "const int",
                                                        "const int", "const char*", "const char*", "const char*",
                                                        "const int"]


def test_python_module_keeps_reference_keywords():
    """nem.pyx:2-14 keyword names, as ppanggolin.py:1814-1826 passes them."""
    import inspect
    import nem as nem_module
    from pangenomenem_amd.nem import nem
    assert nem_module.nem is nem
    assert list(inspect.signature(nem).parameters) == ["Fname", "nk", "algo", "beta", "convergence", "convergence_th",
                                                       "format", "it_max", "dolog", "model_family", "proportion",
                                                       "dispersion", "init_mode"]


def test_no_gpu_means_loud_failure(lib):
    from pangenomenem_amd import engine
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(engine.NemGpuError) as ei:
        engine.NemEngine(100, 10, 3)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/: the
    package must not import, include, link or dlopen anything from it."""
    pkg = os.path.join(ROOT, "pangenomenem_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|pyoracle|libnem_oracle|libnem_ref|#\s*include\s+\"[^\"]*oracle",
                     re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(text), os.path.join(dirpath, f)


@pytest.mark.parametrize("name", [n for n in case_names(files_only=True) if load_case(n)["meta"].get("nem_rc", 0) == 0])
def test_writers_reproduce_reference_text(lib, tmp_path, name):
    """SaveResults' formats: given the reference's full-precision arrays, our writers emit the
    reference's .uf and .mf byte for byte (the criteria line with M = -inf included: the cases with PPanGGOLiN's own
    edge weights; the cases whose class empties have no files, nem_exe.c:624-631)."""
    from pangenomenem_amd import engine
    case = load_case(name)
    exp = case["expected"]
    uf = str(tmp_path / "out.uf")
    mf = str(tmp_path / "out.mf")
    assert engine.write_uf(uf, exp["c"]) == 0
    assert engine.write_mf(mf, exp["crit"], case["cfg"]["beta"], exp["center"], exp["prop"], exp["disp"]) == 0
    assert open(uf, "rb").read() == case["ref_uf"]
    assert open(mf, "rb").read() == case["ref_mf"]


@pytest.mark.parametrize("name", ["c1_path_ncem_sk", "k5_ncem_skd", "w70_ncem_skd"])
def test_readers_parse_ppanggolin_style_files(lib, tmp_path, name):
    from pangenomenem_amd import engine, nemfiles
    case = load_case(name)
    base = nemfiles.write_nem_inputs(str(tmp_path), case["x"], case["nei"], case["prop"], case["center"], case["disp"])
    got = engine.read_inputs(base, case["k"])
    assert got["n"], got["d"] == case["x"].shape
    assert np.array_equal(got["x"], case["x"])
    ptr, idx, w = case["nei"]
    assert np.array_equal(got["nei"][0], ptr) and np.array_equal(got["nei"][1], idx)
    assert np.array_equal(got["nei"][2], w)
    assert got["param_mode"] == 1 and got["type"] == "S"
    assert np.array_equal(got["center"], case["center"])
    assert np.array_equal(got["disp"], case["disp"])
    np.testing.assert_array_equal(got["prop"], case["prop"])


def test_reader_quirks_and_errors(lib, tmp_path):
    from pangenomenem_amd import engine
    base = str(tmp_path / "q")
    open(base + ".str", "w").write("# a comment line\n# another\ns 4 3\n")
    open(base + ".dat", "w").write("1 0 1\n0\t0\t1\n1.0 1 0\n0 0 0\n")
    # point 2 lists an out-of-range neighbour (dropped) and a zero weight (dropped): the two lists are
    # compacted independently (ReadPtsNeighs, nem_exe.c:1408-1462)
    open(base + ".nei", "w").write("# hdr\n1\n1 2 2 3 1.5 2\n2 3 1 9 3 4 0 2.5\n3 0\n4 1 1 0.25\n")
    open(base + ".m", "w").write("2 0.2 0.3 1 1 1 0.5 0.5 0.5 0 0 0 0.1 0.1 0.1 0.5 0.5 0.5 0.1 0.1 0.1")
    got = engine.read_inputs(base, 3)
    assert (got["n"], got["d"], got["param_mode"]) == (4, 3, 2)
    assert got["x"].tolist() == [[1, 0, 1], [0, 0, 1], [1, 1, 0], [0, 0, 0]]
    ptr, idx, w = got["nei"]
    assert ptr.tolist() == [0, 2, 4, 4, 5]
    assert idx.tolist() == [1, 2, 0, 2, 0] and w.tolist() == [1.5, 2.0, 4.0, 2.5, 0.25]
    assert got["prop"].tolist() == [np.float32(0.2), np.float32(0.3),
                                    np.float32(np.float32(np.float32(1) - np.float32(0.2)) - np.float32(0.3))]
    # short .dat -> STS_E_FILE (7); missing .m values -> STS_E_FILEIN (5); bad flag -> 5
    open(base + ".dat", "w").write("1 0 1\n0 0\n")
    with pytest.raises(engine.NemGpuError, match="status 7"):
        engine.read_inputs(base, 3)
    open(base + ".dat", "w").write("1 0 1\n0 0 1\n1 1 0\n0 0 0\n")
    open(base + ".m", "w").write("1 0.2 0.3 1 1 1")
    with pytest.raises(engine.NemGpuError, match="status 5"):
        engine.read_inputs(base, 3)
    open(base + ".dat", "w").write("1 0 2\n0 0 1\n1 1 0\n0 0 0\n")
    with pytest.raises(engine.NemGpuError, match="0/1"):
        engine.read_inputs(base, 3)


def test_dat_reader_fast_path_agrees_with_the_tokeniser(lib, tmp_path):
    """The .dat reader packs 4 values per 8-byte load when a row is the text ppanggolin.py:850 writes
    ("0\\t1\\t...\\n"); every other spelling of the same numbers must fall back and give the same bits."""
    from pangenomenem_amd import engine
    rng = np.random.default_rng(5)
    base = str(tmp_path / "f")
    m = "1 0.3 0.3 " + " ".join(["1"] * 9 + ["0.5"] * 9 + ["0"] * 9) + " " + " ".join(["0.1"] * 27)
    for d in (1, 2, 3, 4, 5, 7, 8, 9, 33, 64, 67):
        n = 23
        x = (rng.random((n, d)) < 0.5).astype(np.uint8)
        open(base + ".str", "w").write("S\t%d\t%d\n" % (n, d))
        open(base + ".nei", "w").write("1\n" + "".join("%d\t0\n" % (i + 1) for i in range(n)))
        k3 = "1 0.3 0.3 " + " ".join(["1"] * d + ["0.5"] * d + ["0"] * d) + " " + " ".join(["0.1"] * (3 * d))
        open(base + ".m", "w").write(k3)
        rows = ["\t".join(str(v) for v in r) for r in x]
        variants = {
            "tabs": "\n".join(rows) + "\n",
            "no final newline": "\n".join(rows),
            "crlf": "\r\n".join(rows) + "\r\n",
            "spaces": "\n".join(r.replace("\t", " ") for r in rows) + "\n",
            "mixed": "\n".join((r if i % 2 else r.replace("\t", "  ")) for i, r in enumerate(rows)) + "\n",
            "floats in one row": "\n".join((r if i != 3 else r.replace("1", "1.0")) for i, r in enumerate(rows)) + "\n",
            "one long line": "\t".join(rows) + "\n",
            "blank lines": "\n\n".join(rows) + "\n\n",
        }
        for name, text in variants.items():
            open(base + ".dat", "w", newline="").write(text)
            got = engine.read_inputs(base, 3)
            assert np.array_equal(got["x"], x), (d, name)
    # a 2 hidden behind the fast path's mask must still be rejected
    open(base + ".dat", "w").write("\n".join("\t".join("2" if (i, j) == (5, 1) else "0" for j in range(67)) for i in range(23)) + "\n")
    with pytest.raises(engine.NemGpuError, match="0/1"):
        engine.read_inputs(base, 3)
    del m


def test_dat_reader_regular_file_in_parallel(lib, tmp_path):
    """A .dat of exactly n rows of 2*d bytes is read by several threads, each pread()ing its rows (6 MB here:
    3 threads); one irregular row anywhere sends the whole file to the general reader; both give the same bits,
    and a bad value is still reported."""
    from pangenomenem_amd import engine
    rng = np.random.default_rng(9)
    n, d = 6000, 500
    base = str(tmp_path / "big")
    x = (rng.random((n, d)) < 0.4).astype(np.uint8)
    open(base + ".str", "w").write("S\t%d\t%d\n" % (n, d))
    open(base + ".nei", "w").write("1\n" + "".join("%d\t0\n" % (i + 1) for i in range(n)))
    open(base + ".m", "w").write("1 0.3 0.3 " + " ".join(["1"] * d + ["0.5"] * d + ["0"] * d) + " " + " ".join(["0.1"] * (3 * d)))
    text = bytearray(b"".join(b"\t".join(b"1" if v else b"0" for v in r) + b"\n" for r in x))
    assert len(text) == 2 * d * n
    open(base + ".dat", "wb").write(text)
    assert np.array_equal(engine.read_inputs(base, 3)["x"], x)
    # same size, one row spelt with a space instead of a tab: general reader, same result
    t2 = bytearray(text); t2[2 * d * 4321 + 1] = ord(" ")
    open(base + ".dat", "wb").write(t2)
    assert np.array_equal(engine.read_inputs(base, 3)["x"], x)
    # a value that is neither 0 nor 1, in the last thread's share
    t3 = bytearray(text); t3[2 * d * 5990 + 10] = ord("2")
    open(base + ".dat", "wb").write(t3)
    with pytest.raises(engine.NemGpuError, match="0/1"):
        engine.read_inputs(base, 3)


@pytest.mark.parametrize("d", [4, 8, 12, 16, 20, 28, 32, 36, 48, 52, 64, 68, 100, 128, 132, 256, 260])
def test_dat_reader_regular_rows_of_every_length(lib, tmp_path, d):
    """The regular-file reader takes 16 values per 32-byte load where the CPU allows and finishes each row 4 at a
    time: every split of a row between the two must give the bits numpy gives."""
    from pangenomenem_amd import engine
    rng = np.random.default_rng(100 + d)
    n = 37
    base = str(tmp_path / "r")
    x = (rng.random((n, d)) < 0.5).astype(np.uint8)
    x[0] = 1; x[1] = 0
    open(base + ".str", "w").write("S\t%d\t%d\n" % (n, d))
    open(base + ".nei", "w").write("1\n" + "".join("%d\t0\n" % (i + 1) for i in range(n)))
    open(base + ".m", "w").write("1 0.3 0.3 " + " ".join(["1"] * d + ["0.5"] * d + ["0"] * d) + " " + " ".join(["0.1"] * (3 * d)))
    open(base + ".dat", "wb").write(b"".join(b"\t".join(b"1" if v else b"0" for v in r) + b"\n" for r in x))
    assert np.array_equal(engine.read_inputs(base, 3)["x"], x)


def test_restated_generator_is_glibc_random(lib):
    """csrc/nem_rng.hpp restates the generator behind the reference's random starts: libc random() after srandom(seed)
    (nem_exe.c:621, nem_rnd.c:40-63).  Checked against this host's libc draw for draw."""
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    libc.random.restype = C.c_long
    libc.srandom.argtypes = [C.c_uint]
    lib.nemgpu_glibc_random.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
    for seed in (0, 1, 2, 12345, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 1759500000):
        n = 2000
        got = np.zeros(n, np.int32)
        assert lib.nemgpu_glibc_random(seed, n, got.ctypes.data_as(C.c_void_p)) == 0
        libc.srandom(seed)
        want = np.array([libc.random() for _ in range(n)], np.int64)
        assert np.array_equal(got.astype(np.int64), want), seed


def test_uf_writer_fast_path_prints_like_printf(lib, tmp_path):
    """The .uf writer formats memberships itself (" %5.3f "); it must print what printf prints: exact binary value,
    three decimals, ties to even -- dyadic ties, values a hair beside them, the ends of the fast path's range,
    and the values it hands to snprintf (negative, -0.0, >= 10, NaN, inf)."""
    from pangenomenem_amd import engine
    rng = np.random.Generator(np.random.PCG64(5))
    ties = np.array([0.0625, 0.1875, 0.3125, 0.4375, 0.5625, 0.0005, 0.0015, 0.0025, 0.9995, 0.99949998, 0.99950004,
                     0.5, 0.25, 0.125, 0.0, 1.0, 1.0 / 3, 2.0 / 3, 1e-30, 1e-4, 4.99e-4, 5.01e-4, 9.9993, 9.9994, 9.99949,
                     9.9995, 10.0, 123.456, -0.0, -0.25, -1e-9, np.nan, np.inf, -np.inf], np.float32)
    near = np.nextafter(ties[:12], np.float32(2)), np.nextafter(ties[:12], np.float32(-2))
    vals = np.concatenate([ties, near[0], near[1], rng.random(3000).astype(np.float32),
                           (rng.integers(0, 8193, 3000) / 8192.0).astype(np.float32),      # many exact ties
                           (rng.random(900) * 10).astype(np.float32)])
    vals = vals[: (len(vals) // 3) * 3].reshape(-1, 3)
    path = str(tmp_path / "fmt.uf")
    assert engine.write_uf(path, vals) == 0
    want = "".join("".join(" %5.3f " % float(v) for v in row) + "\n" for row in vals)
    assert open(path).read() == want


def test_log_formatter_prints_like_printf(lib):
    """The per-iteration log lines are formatted without printf (" %5.3f", " %7.3f", " %7.1f"): same text as printf
    for the exact binary value -- dyadic ties, neighbours of ties, class sizes up to 2^24 and beyond the field width,
    and what goes to snprintf (negative, -0.0, huge, NaN, inf)."""
    import ctypes as C
    from pangenomenem_amd import build as nem_build
    L = C.CDLL(nem_build.build())
    L.nemio_format_fixed.restype = C.c_int
    L.nemio_format_fixed.argtypes = [C.c_float, C.c_int, C.c_int, C.c_char_p]
    rng = np.random.Generator(np.random.PCG64(3))
    base = np.array([0.0, 0.5, 1.0, 0.0625, 0.1875, 0.0005, 0.0015, 0.9995, 0.99949998, 0.05, 0.25, 0.125, 1e-30, 7.5, 12.25,
                     123.45, 99999.95, 999999.5, 16777216.0, 3.0e8, 9.9e8, 1.0e9, 5e12, -0.0, -1.5, np.nan, np.inf, -np.inf],
                    np.float32)
    vals = np.concatenate([base, np.nextafter(base[:12], np.float32(9)), np.nextafter(base[:12], np.float32(-9)),
                           rng.random(2000).astype(np.float32), (rng.integers(0, 8193, 2000) / 8192.0).astype(np.float32),
                           (rng.random(1000) * 20000).astype(np.float32), (rng.integers(0, 400001, 1000) / 20.0).astype(np.float32)])
    out = C.create_string_buffer(64)
    for width, dec in ((5, 3), (7, 3), (7, 1), (5, 0), (10, 2)):
        for v in vals:
            n = L.nemio_format_fixed(float(v), width, dec, out)
            want = " %*.*f" % (width, dec, float(v))
            assert n == len(want) and out.value.decode() == want, (width, dec, float(v), out.value, want)


def test_tokeniser_numbers_equal_strtof_and_strtol(lib, tmp_path):
    """The readers convert short plain decimals and integers by hand; every weight of a .nei written in all sorts of
    spellings must come out as libc's strtof gives it (the reference reads them with fscanf("%f"))."""
    import ctypes as C
    from pangenomenem_amd import engine
    libc = C.CDLL("libc.so.6")
    libc.strtof.restype = C.c_float
    libc.strtof.argtypes = [C.c_char_p, C.c_void_p]
    rng = np.random.Generator(np.random.PCG64(21))
    words = ["1", "8", "0.5", "1.5", "2.25", ".75", "3.", "+4", "0.1", "0.3333333", "16777217", "33554433", "0.00000001",
             "123456.78901234", "99999999.9999999", "1e0", "2.5e-1", "1E2", "0x10", "7.00000000000000001", "000012.5",
             "4294967297", "0.1234567890123456789", "5e-46", "1.17549435e-38", "3.4028235e38", "16777216.5", "8388608.5",
             "0.50000003", "1.00000006", "2.00000012", "9007199254740993"]
    for _ in range(3000):
        nd = int(rng.integers(1, 16)); nf = int(rng.integers(0, min(nd, 9) + 1))
        digs = "".join(rng.choice(list("0123456789"), nd))
        w = digs[: nd - nf] + ("." + digs[nd - nf:] if nf else "")
        if w.startswith("."): w = rng.choice(["", "0"]) + w
        words.append(str(w))
    for _ in range(1500):                                       # floats near ties: a float midpoint written in decimal
        f = np.float32(rng.uniform(0.01, 2000.0))
        mid = (float(f) + float(np.nextafter(f, np.float32(1e9)))) / 2
        words.append(("%.8f" % mid).rstrip("0"))
    words = [w for w in words if libc.strtof(w.encode(), None) != 0.0]   # (zero weights are dropped by the reader)
    n = len(words)
    base = str(tmp_path / "w")
    open(base + ".str", "w").write("S\t%d\t1\n" % n)
    open(base + ".dat", "w").write("\n".join("1" for _ in range(n)) + "\n")
    open(base + ".m", "w").write("1 0.3 0.3 1 0.5 0 0.1 0.1 0.1")
    open(base + ".nei", "w").write("1\n" + "".join("%d 1 %d %s\n" % (i + 1, (i + 1) % n + 1, w) for i, w in enumerate(words)))
    got = engine.read_inputs(base, 3)
    ptr, idx, wts = got["nei"]
    assert list(ptr) == list(range(n + 1)) and list(idx) == [(i + 1) % n for i in range(n)]
    want = np.array([libc.strtof(w.encode(), None) for w in words], np.float32)
    bad = [(w, a, b) for w, a, b in zip(words, wts, want) if np.float32(a).tobytes() != np.float32(b).tobytes()]
    assert not bad, bad[:5]


def test_groups_are_dealt_round_robin_over_the_device_list(lib):
    """nemgpu_solve_many_devices' scheduling (host arithmetic, no GPU): lock-step group g -- problems g*group ..
    g*group + group - 1 -- goes to slot g % n_devices of the device list; every slot's share is whole groups but the
    tail; one device takes everything."""
    from pangenomenem_amd.batch import deal_groups
    assert deal_groups(10, 3, 2) == [0, 0, 0, 1, 1, 1, 0, 0, 0, 1]
    assert deal_groups(5, 1, 3) == [0, 1, 2, 0, 1]
    assert deal_groups(7, 32, 4) == [0] * 7
    assert deal_groups(6, 2, 1) == [0] * 6
    assert deal_groups(0, 4, 2) == []
    for count, group, ndev in ((1000, 32, 8), (333, 7, 5)):
        slots = deal_groups(count, group, ndev)
        assert all(slots[i] == (i // group) % ndev for i in range(count))
        sizes = [slots.count(s) for s in range(ndev)]
        assert max(sizes) - min(sizes) <= group

"""ctypes bindings for the CHECKER libraries -- TEST INFRASTRUCTURE ONLY.

  * ``Oracle``    -> oracle/_build/libnem_oracle.so  (our plain-C restatement, nem_oracle.c)
  * ``Reference`` -> oracle/_ref/libnem_ref.so       (the unmodified reference C sources + ref_harness.c)

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (pangenomenem_amd/) never does.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libnem_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libnem_ref.so")
REF_SRC = "/root/reference/ppanggolin/NEM"

ALGO = {"nem": 0, "ncem": 1}
DISP = {"s__": 0, "sk_": 1, "s_d": 2, "skd": 3}
PROP = {"p_": 0, "pk": 1}
CVT = {"none": 0, "clas": 1, "crit": 2, "crit_logged": 3}
TIE = {"libc": 0, "first": 1, "hash": 2}


def build(ref=True):
    """Compile the checker libraries (gcc).  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref and os.path.isdir(REF_SRC):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def have_reference():
    return os.path.isfile(REF_SO)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _csr(n, nei):
    if nei is None:
        return (np.zeros(n + 1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32))
    ptr, idx, w = nei
    return (np.ascontiguousarray(ptr, np.int32), np.ascontiguousarray(idx if len(idx) else [0], np.int32),
            np.ascontiguousarray(w if len(w) else [0], np.float32))


class _Problem(C.Structure):
    _fields_ = [("n", C.c_int), ("d", C.c_int), ("k", C.c_int),
                ("x", C.POINTER(C.c_ubyte)),
                ("nei_ptr", C.POINTER(C.c_int)), ("nei_idx", C.POINTER(C.c_int)), ("nei_w", C.POINTER(C.c_float)),
                ("algo", C.c_int), ("disper", C.c_int), ("propor", C.c_int), ("cvtest", C.c_int),
                ("beta", C.c_float), ("cvthres", C.c_float),
                ("it_max", C.c_int), ("param_fix", C.c_int), ("tie_rule", C.c_int), ("tie_seed", C.c_uint)]


class _State(C.Structure):
    _fields_ = [("c_nk", C.POINTER(C.c_float)), ("prop_k", C.POINTER(C.c_float)),
                ("center_kd", C.POINTER(C.c_float)), ("disp_kd", C.POINTER(C.c_float)),
                ("nbobs_k", C.POINTER(C.c_float)), ("nbobs_kd", C.POINTER(C.c_float)),
                ("iner_kd", C.POINTER(C.c_float)),
                ("pkfki_nk", C.POINTER(C.c_double)), ("logpkfki_nk", C.POINTER(C.c_float)),
                ("crit", C.c_float * 6), ("iters", C.c_int), ("converged", C.c_int),
                ("emptyk", C.c_int), ("n_zero_density", C.c_int)]


class Oracle:
    """The plain-C restatement (oracle/nem_oracle.c)."""

    def __init__(self):
        if not os.path.isfile(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.orc_last_loop_seconds.restype = C.c_double
        self.lib.orc_mix32.restype = C.c_uint
        self.lib.orc_mix32.argtypes = [C.c_uint, C.c_uint, C.c_uint]

    def mix32(self, seed, sweep, site):
        return int(self.lib.orc_mix32(seed, sweep, site))

    def density(self, x, prop, center, disp):
        n, d = x.shape
        k = len(prop)
        x = np.ascontiguousarray(x, np.uint8)
        prop = np.ascontiguousarray(prop, np.float32)
        center = np.ascontiguousarray(center, np.float32)
        disp = np.ascontiguousarray(disp, np.float32)
        pk = np.zeros((n, k), np.float64)
        lp = np.zeros((n, k), np.float32)
        sts = self.lib.orc_density(n, d, k, _p(x, C.c_ubyte), _p(prop, C.c_float), _p(center, C.c_float),
                                   _p(disp, C.c_float), _p(pk, C.c_double), _p(lp, C.c_float))
        return pk, lp, sts

    def sweep(self, c, nei, beta, pkfki, ncem, tie="hash", seed=0, sweep_id=0):
        n, k = c.shape
        c = np.array(c, np.float32, order="C")
        ptr, idx, w = _csr(n, nei)
        pkfki = np.ascontiguousarray(pkfki, np.float64)
        nz = self.lib.orc_sweep(n, k, _p(ptr, C.c_int), _p(idx, C.c_int), _p(w, C.c_float), C.c_float(beta),
                                _p(pkfki, C.c_double), int(ncem), TIE[tie], C.c_uint(seed), C.c_uint(sweep_id),
                                _p(c, C.c_float))
        return c, nz

    def relax_round(self, lo, hi, nei_local, beta, pkfki_local, ncem, c_old, c_guess, c_out, tie="hash", seed=0,
                    sweep_id=0, key_bias=0):
        """One relaxation round for sites [lo, hi); c_* are GLOBAL [n_total, k] float32 arrays (c_out is
        written in place).  Returns the number of changed sites."""
        k = c_old.shape[1]
        ptr, idx, w = _csr(hi - lo, nei_local)
        pk = np.ascontiguousarray(pkfki_local, np.float64)
        assert c_old.flags.c_contiguous and c_guess.flags.c_contiguous and c_out.flags.c_contiguous
        return int(self.lib.orc_relax_round_keyed(lo, hi, k, _p(ptr, C.c_int), _p(idx, C.c_int), _p(w, C.c_float),
                                                  C.c_float(beta), _p(pk, C.c_double), int(ncem), TIE[tie], C.c_uint(seed),
                                                  C.c_uint(sweep_id), _p(c_old, C.c_float), _p(c_guess, C.c_float),
                                                  _p(c_out, C.c_float), int(key_bias)))

    def mstep(self, x, c, disper, propor, prop, center, disp):
        n, d = x.shape
        k = c.shape[1]
        x = np.ascontiguousarray(x, np.uint8)
        c = np.ascontiguousarray(c, np.float32)
        prop = np.array(prop, np.float32)
        center = np.array(center, np.float32).reshape(k, d)
        disp = np.array(disp, np.float32).reshape(k, d)
        nk = np.zeros(k, np.float32)
        nkd = np.zeros((k, d), np.float32)
        iner = np.zeros((k, d), np.float32)
        ek = C.c_int(0)
        sts = self.lib.orc_mstep(n, d, k, _p(x, C.c_ubyte), _p(c, C.c_float), DISP[disper], PROP[propor],
                                 _p(prop, C.c_float), _p(center, C.c_float), _p(disp, C.c_float),
                                 _p(nk, C.c_float), _p(nkd, C.c_float), _p(iner, C.c_float), C.byref(ek))
        return dict(status=sts, prop=prop, center=center, disp=disp, nbobs_k=nk, nbobs_kd=nkd, iner=iner,
                    emptyk=ek.value)

    def crit(self, c, nei, beta, pkfki, logpkfki):
        n, k = c.shape
        ptr, idx, w = _csr(n, nei)
        c = np.ascontiguousarray(c, np.float32)
        pkfki = np.ascontiguousarray(pkfki, np.float64)
        logpkfki = np.ascontiguousarray(logpkfki, np.float32)
        out = (C.c_float * 6)()
        self.lib.orc_crit(n, k, _p(ptr, C.c_int), _p(idx, C.c_int), _p(w, C.c_float), C.c_float(beta),
                          _p(c, C.c_float), _p(pkfki, C.c_double), _p(logpkfki, C.c_float), out)
        return np.array(list(out), np.float32)

    def run(self, x, nei, k, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", propor="pk",
            cvtest="clas", cvthres=1e-8, it_max=100, param_fix=False, tie="hash", seed=0):
        """Full loop (INIT_PARAM_FILE).  Returns a dict with full-precision results."""
        n, d = x.shape
        x = np.ascontiguousarray(x, np.uint8)
        ptr, idx, w = _csr(n, nei)
        prop = np.array(prop, np.float32)
        center = np.array(center, np.float32).reshape(k, d)
        disp = np.array(disp, np.float32).reshape(k, d)
        c = np.zeros((n, k), np.float32)
        nk = np.zeros(k, np.float32)
        nkd = np.zeros((k, d), np.float32)
        iner = np.zeros((k, d), np.float32)
        pk = np.zeros((n, k), np.float64)
        lp = np.zeros((n, k), np.float32)
        p = _Problem(n, d, k, _p(x, C.c_ubyte), _p(ptr, C.c_int), _p(idx, C.c_int), _p(w, C.c_float),
                     ALGO[algo], DISP[disper], PROP[propor], CVT[cvtest], beta, cvthres, it_max,
                     int(param_fix), TIE[tie], seed)
        s = _State(_p(c, C.c_float), _p(prop, C.c_float), _p(center, C.c_float), _p(disp, C.c_float),
                   _p(nk, C.c_float), _p(nkd, C.c_float), _p(iner, C.c_float), _p(pk, C.c_double),
                   _p(lp, C.c_float))
        sts = self.lib.orc_run(C.byref(p), C.byref(s))
        return dict(status=sts, c=c, prop=prop, center=center, disp=disp, nbobs_k=nk,
                    crit=np.array(list(s.crit), np.float32), iters=s.iters, converged=bool(s.converged),
                    emptyk=s.emptyk, n_zero_density=s.n_zero_density, pkfki=pk, logpkfki=lp,
                    loop_seconds=float(self.lib.orc_last_loop_seconds()))

    def run_random(self, x, nei, k, n_starts=50, rng_seed=1, algo="ncem", beta=0.5, disper="sk_", propor="pk",
                   cvtest="clas", cvthres=1e-8, it_max=100, tie="libc", seed=None):
        """INIT_RANDOM (RandNemAlgo): n_starts random starts drawn with libc random() after srandom(rng_seed).
        tie="libc" shares that stream with the tie-breaks (the reference); "hash"/"first" use `seed` as tie seed."""
        n, d = x.shape
        x = np.ascontiguousarray(x, np.uint8)
        ptr, idx, w = _csr(n, nei)
        prop = np.zeros(k, np.float32)
        center = np.zeros((k, d), np.float32)
        disp = np.zeros((k, d), np.float32)
        c = np.zeros((n, k), np.float32)
        nk = np.zeros(k, np.float32)
        nkd = np.zeros((k, d), np.float32)
        iner = np.zeros((k, d), np.float32)
        p = _Problem(n, d, k, _p(x, C.c_ubyte), _p(ptr, C.c_int), _p(idx, C.c_int), _p(w, C.c_float),
                     ALGO[algo], DISP[disper], PROP[propor], CVT[cvtest], beta, cvthres, it_max, 0, TIE[tie],
                     rng_seed if seed is None else seed)
        s = _State(_p(c, C.c_float), _p(prop, C.c_float), _p(center, C.c_float), _p(disp, C.c_float),
                   _p(nk, C.c_float), _p(nkd, C.c_float), _p(iner, C.c_float), None, None)
        best = C.c_int(-1)
        self.lib.orc_run_random.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.POINTER(C.c_int)]
        sts = self.lib.orc_run_random(C.byref(p), C.byref(s), int(n_starts), C.c_uint(rng_seed), C.byref(best))
        return dict(status=sts, c=c, prop=prop, center=center, disp=disp, nbobs_k=nk,
                    crit=np.array(list(s.crit), np.float32), iters=s.iters, converged=bool(s.converged),
                    best_start=best.value)


class Reference:
    """The unmodified reference C NEM, compiled from /root/reference by oracle/Makefile."""

    def __init__(self):
        if not os.path.isfile(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = C.CDLL(REF_SO)
        self.lib.nem.restype = C.c_int
        self.lib.nem.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_float, C.c_char_p, C.c_float, C.c_char_p,
                                 C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]

    def nem(self, fname, nk, algo=b"ncem", beta=0.5, convergence=b"clas", convergence_th=1e-8, format=b"fuzzy",
            it_max=100, dolog=1, model_family=b"bern", proportion=b"pk", dispersion=b"sk_", init_mode=2):
        """nem() exactly as ppanggolin.py:1814-1826 calls it.  dolog must be 1 (with 0 the
        reference fclose()s the process's stderr, nem_exe.c:274,657)."""
        if isinstance(fname, str):
            fname = fname.encode()
        return self.lib.nem(fname, nk, algo, beta, convergence, convergence_th, format, it_max, dolog,
                            model_family, proportion, dispersion, init_mode)

    def classify(self, x, nei, k, prop, center, disp, algo="ncem", beta=0.5, disper="sk_", propor="pk",
                 cvtest="clas", cvthres=1e-8, it_max=100, param_fix=False, seed=12345):
        n, d = x.shape
        xf = np.ascontiguousarray(x, np.float32)
        ptr, idx, w = _csr(n, nei)
        prop = np.array(prop, np.float32)
        center = np.array(center, np.float32).reshape(k, d)
        disp = np.array(disp, np.float32).reshape(k, d)
        c = np.zeros((n, k), np.float32)
        nk = np.zeros(k, np.float32)
        crit = np.zeros(6, np.float32)
        log = C.create_string_buffer(1 << 16)
        secs = C.c_double(0)
        sts = self.lib.ref_classify(n, d, k, _p(xf, C.c_float), _p(ptr, C.c_int), _p(idx, C.c_int),
                                    _p(w, C.c_float), ALGO[algo], C.c_float(beta), DISP[disper], PROP[propor],
                                    CVT[cvtest], C.c_float(cvthres), it_max, 0 if param_fix else 1,
                                    C.c_long(seed), _p(prop, C.c_float), _p(center, C.c_float),
                                    _p(disp, C.c_float), _p(c, C.c_float), _p(nk, C.c_float),
                                    _p(crit, C.c_float), log, len(log), C.byref(secs))
        text = log.value.decode("latin1")
        m = re.search(r"(converged|did not converge) after (\d+) iterations", text)
        iters = int(m.group(2)) if m else None
        if iters is None:
            m2 = re.findall(r"\x08\x08\x08\x08\x08\s*(\d+) ", text)
            iters = int(m2[-1]) if m2 else 0
        return dict(status=sts, c=c, prop=prop, center=center, disp=disp, nbobs_k=nk, crit=crit, iters=iters,
                    converged=bool(m and m.group(1) == "converged"), log=text, seconds=secs.value,
                    zero_density=("density = 0" in text))

    def classify_random(self, x, nei, k, n_starts=50, rng_seed=1, algo="ncem", beta=0.5, disper="sk_", propor="pk",
                        cvtest="clas", cvthres=1e-8, it_max=100, log_path=None):
        """ClassifyByNem with InitMode = INIT_RANDOM after srandom(rng_seed) (the reference seeds with time()).
        log_path: the reference writes its own per-iteration log there (DoLog, StartLogFile nem_alg.c:1478-1498)."""
        n, d = x.shape
        self.lib.ref_set_log.restype = None
        self.lib.ref_set_log(C.c_char_p(log_path.encode()) if log_path else None)
        xf = np.ascontiguousarray(x, np.float32)
        ptr, idx, w = _csr(n, nei)
        prop = np.zeros(k, np.float32)
        center = np.zeros((k, d), np.float32)
        disp = np.zeros((k, d), np.float32)
        c = np.zeros((n, k), np.float32)
        nk = np.zeros(k, np.float32)
        crit = np.zeros(6, np.float32)
        log = C.create_string_buffer(1 << 20)
        secs = C.c_double(0)
        sts = self.lib.ref_classify_ex(n, d, k, _p(xf, C.c_float), _p(ptr, C.c_int), _p(idx, C.c_int),
                                       _p(w, C.c_float), ALGO[algo], C.c_float(beta), DISP[disper], PROP[propor],
                                       CVT[cvtest], C.c_float(cvthres), it_max, 1, C.c_long(rng_seed), 1, int(n_starts),
                                       _p(prop, C.c_float), _p(center, C.c_float), _p(disp, C.c_float),
                                       _p(c, C.c_float), _p(nk, C.c_float), _p(crit, C.c_float), log, len(log),
                                       C.byref(secs))
        self.lib.ref_set_log(None)
        text = log.value.decode("latin1")
        m = re.search(r"Best start was (\d+)", text)
        return dict(status=sts, c=c, prop=prop, center=center, disp=disp, nbobs_k=nk, crit=crit,
                    best_start=int(m.group(1)) - 1 if m else -1, log=text, seconds=secs.value)

    def estim_para(self, x, c, disper, propor, prop, center, disp):
        n, d = x.shape
        k = c.shape[1]
        xf = np.ascontiguousarray(x, np.float32)
        c = np.ascontiguousarray(c, np.float32)
        prop = np.array(prop, np.float32)
        center = np.array(center, np.float32).reshape(k, d)
        disp = np.array(disp, np.float32).reshape(k, d)
        nk = np.zeros(k, np.float32)
        nkd = np.zeros((k, d), np.float32)
        iner = np.zeros((k, d), np.float32)
        ek = C.c_int(0)
        sts = self.lib.ref_estim_para(n, d, k, _p(xf, C.c_float), _p(c, C.c_float), DISP[disper], PROP[propor],
                                      _p(prop, C.c_float), _p(center, C.c_float), _p(disp, C.c_float),
                                      _p(nk, C.c_float), _p(nkd, C.c_float), _p(iner, C.c_float), C.byref(ek))
        return dict(status=sts, prop=prop, center=center, disp=disp, nbobs_k=nk, nbobs_kd=nkd, iner=iner,
                    emptyk=ek.value)

"""ctypes binding of the in-memory engine (include/nem_mi355x.h, ``nemgpu_*``).

This is plumbing: every number is produced by the HIP kernels in ``lib/libnem_mi355x.so``.
Loading fails loudly when the library has not been built; creating an engine fails loudly when
no HIP device is usable (there is no CPU fallback).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NEM_MI355X_LIB: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get("NEM_MI355X_LIB") or os.path.join(_HERE, "lib", "libnem_mi355x.so")

ALGO = {"nem": 0, "ncem": 1}
DISP = {"s__": 0, "sk_": 1, "s_d": 2, "skd": 3}
PROP = {"p_": 0, "pk": 1}
CVT = {"none": 0, "clas": 1, "crit": 2, "crit_logged": 3}
TIE = {"libc": 0, "first": 1, "hash": 2}
STATUS_OK, STATUS_W_EMPTYCLASS, STATUS_E_DEVICE, STATUS_E_INTERNAL = 0, 2, 9, 10


class NemGpuError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("algo", C.c_int), ("beta", C.c_float), ("disper", C.c_int), ("propor", C.c_int),
                ("cvtest", C.c_int), ("cvthres", C.c_float), ("it_max", C.c_int), ("param_fix", C.c_int),
                ("tie_rule", C.c_int), ("tie_seed", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("converged", C.c_int), ("emptyk", C.c_int),
                ("zero_density_sites", C.c_int), ("first_zero_density_site", C.c_int),
                ("sweep_rounds", C.c_int), ("crit", C.c_float * 6), ("loop_seconds", C.c_double),
                ("tie_draws", C.c_int)]


class Problem(C.Structure):
    """nemgpu_problem (include/nem_mi355x.h): one whole problem of nemgpu_solve_many, host arrays in and out."""
    _fields_ = [("n", C.c_int), ("d", C.c_int), ("k", C.c_int),
                ("x_bytes", C.c_void_p), ("x_bits", C.c_void_p),
                ("nei_ptr", C.c_void_p), ("nei_idx", C.c_void_p), ("nei_w", C.c_void_p),
                ("prop", C.c_void_p), ("center", C.c_void_p), ("disp", C.c_void_p),
                ("out_prop", C.c_void_p), ("out_center", C.c_void_p), ("out_disp", C.c_void_p),
                ("out_nbobs_k", C.c_void_p), ("out_c", C.c_void_p),
                ("result", Result), ("rc", C.c_int)]


_lib = None


def load_library():
    """dlopen the in-tree HIP library.  Raises if it has not been built (python -m pangenomenem_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NemGpuError("%s is missing: build it with `python pangenomenem_amd/build.py` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    lib.nemgpu_last_error.restype = C.c_char_p
    lib.nemgpu_default_device.argtypes = [C.POINTER(C.c_int)]
    lib.nemgpu_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.nemgpu_destroy.argtypes = [vp]
    lib.nemgpu_destroy.restype = None
    lib.nemgpu_set_matrix_bytes.argtypes = [vp, vp]
    lib.nemgpu_set_matrix_bits.argtypes = [vp, vp]
    lib.nemgpu_set_graph.argtypes = [vp, vp, vp, vp]
    lib.nemgpu_set_params.argtypes = [vp, vp, vp, vp]
    lib.nemgpu_configure.argtypes = [vp, C.POINTER(Config)]
    lib.nemgpu_run.argtypes = [vp, C.POINTER(Result)]
    lib.nemgpu_run_many.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Result)]
    lib.nemgpu_run_random.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(Result), ip]
    lib.nemgpu_glibc_random.argtypes = [C.c_uint32, C.c_int, vp]
    lib.nemgpu_init_partition.argtypes = [vp]
    lib.nemgpu_iterate.argtypes = [vp, C.c_int, C.POINTER(Result)]
    lib.nemgpu_reset.argtypes = [vp]
    lib.nemgpu_restart_iterate.argtypes = [vp, C.c_int, C.POINTER(Result)]
    lib.nemgpu_density.argtypes = [vp]
    lib.nemgpu_sweep.argtypes = [vp, C.c_float, ip]
    lib.nemgpu_mstep.argtypes = [vp, ip]
    lib.nemgpu_criteria.argtypes = [vp, fp]
    lib.nemgpu_set_partition.argtypes = [vp, vp]
    lib.nemgpu_get_partition.argtypes = [vp, vp]
    lib.nemgpu_get_labels.argtypes = [vp, vp]
    lib.nemgpu_get_params.argtypes = [vp, vp, vp, vp, vp]
    lib.nemgpu_get_results.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.nemgpu_solve_many.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    lib.nemgpu_solve_many_devices.argtypes = [vp, C.c_int, vp, ip, C.c_int, C.c_int, C.c_int]
    lib.nemgpu_deal_groups.argtypes = [C.c_int, C.c_int, C.c_int, ip]
    lib.nemgpu_get_density.argtypes = [vp, vp, vp]
    lib.nemgpu_profile_density.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), ip]
    lib.nemgpu_calibrate_fetch.argtypes = [C.c_size_t, C.c_int]
    lib.nemgpu_profile_kernels.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), ip]
    lib.nemgpu_rccl_time_allgather.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_double)]
    lib.nemgpu_set_stream.argtypes = [vp, vp]
    lib.nemgpu_set_fast_forward.argtypes = [vp, C.c_int]
    lib.nemgpu_set_graph_policy.argtypes = [vp, C.c_int]
    lib.nemgpu_graph_counters.argtypes = [vp, ip]
    lib.nemgpu_profile_density_many.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.nemgpu_sweep_counters.argtypes = [vp, ip]
    lib.nemgpu_random_start_counters.argtypes = [vp, ip]
    lib.nemgpu_rccl_ranks.argtypes = [vp]
    lib.nemgpu_rccl_selftest.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.nemgpu_ff_table.argtypes = [C.c_double, C.c_double, vp, vp]
    lib.nemgpu_shard_end_enqueue.argtypes = [vp]
    lib.nemgpu_stats_words.argtypes = [vp]
    lib.nemgpu_shard_layout.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.nemgpu_shard_begin.argtypes = [vp]
    lib.nemgpu_shard_begin_restart.argtypes = [vp]
    lib.nemgpu_shard_estep_round1_counts.argtypes = [vp, C.c_float, C.c_int, vp, vp, vp, vp]
    lib.nemgpu_shard_mstep_partial.argtypes = [vp, vp, vp]
    lib.nemgpu_shard_counts.argtypes = [vp, vp]
    lib.nemgpu_rccl_open.argtypes = [C.c_char_p]
    lib.nemgpu_rccl_unique_id.argtypes = [vp]
    lib.nemgpu_rccl_attach.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.nemgpu_shard_enqueue_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, C.c_int]
    lib.nemgpu_shard_estep_round0.argtypes = [vp, vp, C.c_float, C.c_int, vp, vp]
    lib.nemgpu_shard_estep_round1.argtypes = [vp, C.c_float, C.c_int, vp, vp, vp]
    lib.nemgpu_shard_finish_iteration.argtypes = [vp, C.c_float, C.c_int, vp, vp, vp]
    lib.nemgpu_shard_round_sync.argtypes = [vp, C.c_float, C.c_int, vp, vp, vp, ip]
    lib.nemgpu_shard_end.argtypes = [vp, C.POINTER(Result), ip, ip]
    lib.nemgpu_shard_set_labels.argtypes = [vp, vp, vp, vp]
    lib.nemgpu_shard_round_draws.argtypes = [vp, ip, ip]
    lib.nemgpu_shard_grow_draws.argtypes = [vp]
    lib.nemgpu_shard_book_draws.argtypes = [vp, C.c_int]
    lib.nemgpu_shard_set_sweep_number.argtypes = [vp, C.c_int]
    lib.nemgpu_shard_fuzzy_layout.argtypes = [vp, C.c_int]
    lib.nemgpu_shard_fuzzy_round.argtypes = [vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, ip, ip, ip]
    lib.nemgpu_shard_fuzzy_mstep_cols.argtypes = [vp, vp, vp, vp, vp]
    lib.nemgpu_shard_fuzzy_finish.argtypes = [vp, vp, vp, vp, ip]
    lib.nemgpu_shard_fuzzy_moved.argtypes = [vp, vp, vp, ip]
    lib.nemio_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.nemio_free.argtypes = [vp]
    lib.nemio_free.restype = None
    lib.nemio_sizes.argtypes = [vp, ip, ip, ip, ip, ip, ip]
    lib.nemio_copy.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nemio_write_uf.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    lib.nemio_write_cf.argtypes = [C.c_char_p, vp, C.c_int, C.c_int, C.c_int, C.c_uint32]
    lib.nemio_write_mf.argtypes = [C.c_char_p, fp, C.c_float, C.c_int, C.c_int, vp, vp, vp]
    lib.nem.restype = C.c_int
    lib.nem.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_float, C.c_char_p, C.c_float, C.c_char_p, C.c_int,
                        C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    _lib = lib
    return lib


def device_count():
    return int(load_library().nemgpu_device_count())


def default_device():
    """The device a call that was not given one runs on (NEM_MI355X_DEVICE = index | auto, else the current HIP
    device); raises in a forked child of a GPU-using process and on a malformed NEM_MI355X_DEVICE."""
    lib = load_library()
    dev = C.c_int(0)
    rc = lib.nemgpu_default_device(C.byref(dev))
    if rc != 0:
        raise NemGpuError("nemgpu_default_device failed (status %d): %s" % (rc, lib.nemgpu_last_error().decode()))
    return dev.value


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class NemEngine:
    """One NEM problem resident on one GPU (mirrors the reference's ClassifyByNem state:
    DataT + SpatialT + StatModelT + NemParaT, nem_typ.h:286-445)."""

    def __init__(self, n, d, k, device=0, stream=None, site_lo=0, site_hi=None):
        self.lib = load_library()
        self.n_total, self.d, self.k = int(n), int(d), int(k)
        self.lo = int(site_lo)
        self.hi = int(n if site_hi is None else site_hi)
        self.n = self.hi - self.lo
        self._h = C.c_void_p()
        self._chk(self.lib.nemgpu_create(C.byref(self._h), self.n_total, self.d, self.k, self.lo, self.hi,
                                         int(device), C.c_void_p(stream) if stream else None))

    def _chk(self, rc, allow=(STATUS_OK,)):
        if rc not in allow:
            raise NemGpuError("nemgpu call failed (status %d): %s" % (rc, self.lib.nemgpu_last_error().decode()))
        return rc

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nemgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs
    def set_matrix(self, x):
        x = np.ascontiguousarray(x, np.uint8)
        assert x.shape == (self.n, self.d)
        self._chk(self.lib.nemgpu_set_matrix_bytes(self._h, _vp(x)))

    def set_matrix_bits(self, bits):
        bits = np.ascontiguousarray(bits, np.uint32)
        assert bits.shape == (self.n, (self.d + 31) // 32)
        self._chk(self.lib.nemgpu_set_matrix_bits(self._h, _vp(bits)))

    def set_graph(self, nei):
        if nei is None:
            ptr = np.zeros(self.n + 1, np.int32)
            idx = np.zeros(1, np.int32)
            w = np.zeros(1, np.float32)
        else:
            ptr, idx, w = (np.ascontiguousarray(nei[0], np.int32), np.ascontiguousarray(nei[1], np.int32),
                           np.ascontiguousarray(nei[2], np.float32))
        self._chk(self.lib.nemgpu_set_graph(self._h, _vp(ptr), _vp(idx), _vp(w)))

    def set_params(self, prop, center, disp):
        prop = np.ascontiguousarray(prop, np.float32)
        center = np.ascontiguousarray(center, np.float32).reshape(self.k, self.d)
        disp = np.ascontiguousarray(disp, np.float32).reshape(self.k, self.d)
        self._chk(self.lib.nemgpu_set_params(self._h, _vp(prop), _vp(center), _vp(disp)))

    def configure(self, algo="ncem", beta=0.5, disper="sk_", propor="pk", cvtest="clas", cvthres=1e-8,
                  it_max=100, param_fix=False, tie="hash", seed=0):
        self.cfg = Config(ALGO[algo], beta, DISP[disper], PROP[propor], CVT[cvtest], cvthres, it_max,
                          int(param_fix), TIE[tie], seed)
        self._chk(self.lib.nemgpu_configure(self._h, C.byref(self.cfg)))

    # ---- whole run / steps
    def _result(self, r):
        return dict(status=r.status, iters=r.iters, converged=bool(r.converged), emptyk=r.emptyk,
                    n_zero_density=r.zero_density_sites, first_zero_density_site=r.first_zero_density_site,
                    sweep_rounds=r.sweep_rounds, crit=np.array(list(r.crit), np.float32),
                    loop_seconds=r.loop_seconds, tie_draws=r.tie_draws)

    def run(self):
        r = Result()
        self._chk(self.lib.nemgpu_run(self._h, C.byref(r)))
        out = self._result(r)
        out.update(self.results())
        return out

    def run_random(self, n_starts=50, rng_seed=1):
        """The reference's init_mode = INIT_RANDOM: n_starts random starts, best by criterion M (no set_params needed)."""
        r = Result()
        best = C.c_int(-1)
        self._chk(self.lib.nemgpu_run_random(self._h, int(n_starts), C.c_uint32(rng_seed), C.byref(r), C.byref(best)))
        out = self._result(r)
        out.update(self.results())
        out["best_start"] = best.value
        return out

    def init_partition(self):
        self._chk(self.lib.nemgpu_init_partition(self._h))

    def iterate(self, n_iters):
        r = Result()
        self._chk(self.lib.nemgpu_iterate(self._h, int(n_iters), C.byref(r)))
        return self._result(r)

    def reset(self):
        self._chk(self.lib.nemgpu_reset(self._h))

    def restart_iterate(self, n_iters):
        """reset + initial sweeps + up to n_iters EM iterations, enqueued as one pipelined sequence."""
        r = Result()
        self._chk(self.lib.nemgpu_restart_iterate(self._h, int(n_iters), C.byref(r)))
        return self._result(r)

    def density(self):
        self._chk(self.lib.nemgpu_density(self._h))
        pk = np.zeros((self.n, self.k), np.float64)
        lp = np.zeros((self.n, self.k), np.float32)
        self._chk(self.lib.nemgpu_get_density(self._h, _vp(pk), _vp(lp)))
        return pk, lp

    def sweep(self, beta):
        rounds = C.c_int(0)
        self._chk(self.lib.nemgpu_sweep(self._h, C.c_float(beta), C.byref(rounds)))
        return rounds.value

    def mstep(self):
        ek = C.c_int(0)
        rc = self._chk(self.lib.nemgpu_mstep(self._h, C.byref(ek)), allow=(STATUS_OK, STATUS_W_EMPTYCLASS))
        return rc, ek.value

    def criteria(self):
        out = (C.c_float * 6)()
        self._chk(self.lib.nemgpu_criteria(self._h, out))
        return np.array(list(out), np.float32)

    def set_partition(self, c):
        c = np.ascontiguousarray(c, np.float32)
        assert c.shape == (self.n_total, self.k)
        self._chk(self.lib.nemgpu_set_partition(self._h, _vp(c)))

    # ---- outputs
    def partition(self):
        c = np.zeros((self.n, self.k), np.float32)
        self._chk(self.lib.nemgpu_get_partition(self._h, _vp(c)))
        return c

    def labels(self):
        lab = np.zeros(self.n, np.uint8)
        self._chk(self.lib.nemgpu_get_labels(self._h, _vp(lab)))
        return lab

    def params(self):
        prop = np.zeros(self.k, np.float32)
        center = np.zeros((self.k, self.d), np.float32)
        disp = np.zeros((self.k, self.d), np.float32)
        nk = np.zeros(self.k, np.float32)
        self._chk(self.lib.nemgpu_get_params(self._h, _vp(prop), _vp(center), _vp(disp), _vp(nk)))
        return dict(prop=prop, center=center, disp=disp, nbobs_k=nk)

    def results(self):
        prop = np.zeros(self.k, np.float32)
        center = np.zeros((self.k, self.d), np.float32)
        disp = np.zeros((self.k, self.d), np.float32)
        nk = np.zeros(self.k, np.float32)
        c = np.zeros((self.n, self.k), np.float32)
        self._chk(self.lib.nemgpu_get_results(self._h, _vp(prop), _vp(center), _vp(disp), _vp(nk), _vp(c)))
        return dict(prop=prop, center=center, disp=disp, nbobs_k=nk, c=c)

    # ---- multi-GPU step pieces (raw device pointers; see pangenomenem_amd/distributed.py)
    def stats_words(self):
        return int(self.lib.nemgpu_stats_words(self._h))

    def shard_layout(self, world, rank, blk, stride, n_true):
        self._chk(self.lib.nemgpu_shard_layout(self._h, world, rank, blk, stride, n_true))

    def shard_begin(self, restart=False):
        self._chk((self.lib.nemgpu_shard_begin_restart if restart else self.lib.nemgpu_shard_begin)(self._h))

    def shard_estep_round1_counts(self, beta, sweep_id, old_ptr, guess_ptr, out_ptr, stats_ptr):
        self._chk(self.lib.nemgpu_shard_estep_round1_counts(self._h, C.c_float(beta), int(sweep_id), C.c_void_p(old_ptr),
                                                            C.c_void_p(guess_ptr), C.c_void_p(out_ptr), C.c_void_p(stats_ptr)))

    def shard_mstep_partial(self, labels_ptr, stats_ptr):
        self._chk(self.lib.nemgpu_shard_mstep_partial(self._h, C.c_void_p(labels_ptr), C.c_void_p(stats_ptr)))

    def shard_enqueue_batch(self, with_init, n_iters, base, beta, want_stats, lab_ptrs, stats_off):
        self._chk(self.lib.nemgpu_shard_enqueue_batch(self._h, int(with_init), int(n_iters), int(base), C.c_float(beta),
                                                      int(want_stats), C.c_void_p(lab_ptrs[0]), C.c_void_p(lab_ptrs[1]),
                                                      C.c_void_p(lab_ptrs[2]), int(stats_off)))

    def shard_counts(self, stats_ptr):
        self._chk(self.lib.nemgpu_shard_counts(self._h, C.c_void_p(stats_ptr)))

    def shard_estep_round0(self, stats_ptr, beta, sweep_id, old_ptr, out_ptr):
        self._chk(self.lib.nemgpu_shard_estep_round0(self._h, C.c_void_p(stats_ptr) if stats_ptr else None,
                                                     C.c_float(beta), int(sweep_id), C.c_void_p(old_ptr),
                                                     C.c_void_p(out_ptr)))

    def shard_estep_round1(self, beta, sweep_id, old_ptr, guess_ptr, out_ptr):
        self._chk(self.lib.nemgpu_shard_estep_round1(self._h, C.c_float(beta), int(sweep_id), C.c_void_p(old_ptr),
                                                     C.c_void_p(guess_ptr), C.c_void_p(out_ptr)))

    def shard_finish_iteration(self, beta, is_init, old_ptr, q_ptr, r_ptr):
        self._chk(self.lib.nemgpu_shard_finish_iteration(self._h, C.c_float(beta), int(is_init), C.c_void_p(old_ptr),
                                                         C.c_void_p(q_ptr), C.c_void_p(r_ptr)))

    def shard_round_sync(self, beta, sweep_id, old_ptr, guess_ptr, out_ptr):
        ch = C.c_int(0)
        self._chk(self.lib.nemgpu_shard_round_sync(self._h, C.c_float(beta), int(sweep_id), C.c_void_p(old_ptr),
                                                   C.c_void_p(guess_ptr), C.c_void_p(out_ptr), C.byref(ch)))
        return ch.value

    def shard_end_enqueue(self):
        self._chk(self.lib.nemgpu_shard_end_enqueue(self._h))

    def shard_end(self):
        r, commits, need = Result(), C.c_int(0), C.c_int(0)
        self._chk(self.lib.nemgpu_shard_end(self._h, C.byref(r), C.byref(commits), C.byref(need)))
        return dict(status=r.status, iters=r.iters, converged=bool(r.converged), emptyk=r.emptyk,
                    sweep_rounds=r.sweep_rounds, commits=commits.value, need_rounds=need.value)

    def shard_set_labels(self, ptrs):
        self._chk(self.lib.nemgpu_shard_set_labels(self._h, C.c_void_p(ptrs[0]), C.c_void_p(ptrs[1]), C.c_void_p(ptrs[2])))

    def shard_round_draws(self):
        d, short = C.c_int(0), C.c_int(0)
        self._chk(self.lib.nemgpu_shard_round_draws(self._h, C.byref(d), C.byref(short)))
        return d.value, short.value

    def shard_grow_draws(self):
        self._chk(self.lib.nemgpu_shard_grow_draws(self._h))

    def shard_book_draws(self, n):
        self._chk(self.lib.nemgpu_shard_book_draws(self._h, int(n)))

    def shard_set_sweep_number(self, n):
        self._chk(self.lib.nemgpu_shard_set_sweep_number(self._h, int(n)))

    def profile_density(self, reps=50):
        """Average duration of the E1 density kernel (HIP events on the engine's stream) and its algorithmic bytes."""
        ms, b, fused = C.c_double(0), C.c_double(0), C.c_int(0)
        self._chk(self.lib.nemgpu_profile_density(self._h, int(reps), C.byref(ms), C.byref(b), C.byref(fused)))
        return dict(density_ms_avg=ms.value, density_launches=int(reps), algorithmic_bytes_per_launch=b.value,
                    kernel="k_density_fused" if fused.value else "k_density")

    def profile_kernels(self, reps=50):
        """Average duration (HIP events around `reps` back-to-back launches) and algorithmic bytes per launch of the
        three kernels of a solo NCEM iteration: E1, one relaxation round, the M-step counts."""
        ms, by, which = (C.c_double * 3)(), (C.c_double * 3)(), C.c_int(0)
        self._chk(self.lib.nemgpu_profile_kernels(self._h, int(reps), ms, by, C.byref(which)))
        names = ("k_density_fused" if which.value else "k_density", "k_sweep", "k_mstep_counts")
        what = ("E1: Bernoulli log-density chains" + (" + NCEM parameter update" if which.value else ""),
                "E2: one relaxation round of the Gauss-Seidel site sweep", "M: integer popcount statistics")
        return [dict(kernel=names[i], what=what[i], avg_launch_ms=ms[i], algorithmic_bytes_per_launch=by[i], launches_timed=int(reps))
                for i in range(3)]

    def rccl_time_allgather(self, buf_ptr, reps=100):
        ms = C.c_double(0)
        self._chk(self.lib.nemgpu_rccl_time_allgather(self._h, C.c_void_p(buf_ptr), int(reps), C.byref(ms)))
        return ms.value

    def set_fast_forward(self, mode):
        """E1 binade fast-forward: 1 always, 0 never, -1 automatic (default); results are bit-identical."""
        self._chk(self.lib.nemgpu_set_fast_forward(self._h, int(mode)))

    def set_graph_policy(self, capture_on_first=True):
        """Capture a batch shape of the pipelined loop into a hipGraph the first time it is enqueued."""
        self._chk(self.lib.nemgpu_set_graph_policy(self._h, int(bool(capture_on_first))))

    def graph_counters(self):
        out = (C.c_int * 4)()
        self._chk(self.lib.nemgpu_graph_counters(self._h, out))
        return dict(plain=out[0], captured=out[1], replayed=out[2], host_finished_sweeps=out[3])

    def sweep_counters(self):
        out = (C.c_int * 4)()
        self._chk(self.lib.nemgpu_sweep_counters(self._h, out))
        return dict(fused_launches=out[0], fused_failed=out[1], fused_on=bool(out[2]))

    def random_start_counters(self):
        out = (C.c_int * 4)()
        self._chk(self.lib.nemgpu_random_start_counters(self._h, out))
        return dict(rounds=out[0], in_lockstep=out[1], alone=out[2], redone=out[3])

    def rccl_ranks(self):
        return int(self.lib.nemgpu_rccl_ranks(self._h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.nemgpu_set_stream(self._h, C.c_void_p(stream_ptr)))


def run_many(engines, fetch=True):
    """nemgpu_run of several engines in lock step (one launch per EM step for all of them).  Returns their result
    dicts, each identical to what engine.run() alone gives (fetch=False: without the arrays -- engine.results()
    fetches them, e.g. from worker threads)."""
    if not engines:
        return []
    lib = engines[0].lib
    handles = (C.c_void_p * len(engines))(*[e._h for e in engines])
    res = (Result * len(engines))()
    engines[0]._chk(lib.nemgpu_run_many(handles, len(engines), res))
    out = []
    for e, r in zip(engines, res):
        d = e._result(r)
        if fetch:
            d.update(e.results())
        out.append(d)
    return out


def calibrate_fetch(nbytes=1 << 30, reps=3):
    """One known-size 16-bytes-per-lane read, for scaling rocprofv3's FETCH_SIZE (profiles/README.md)."""
    rc = load_library().nemgpu_calibrate_fetch(int(nbytes), int(reps))
    if rc != 0:
        raise NemGpuError("nemgpu_calibrate_fetch failed (status %d)" % rc)


def profile_density_many(engines, reps=30):
    """E1 of a lock-step batch: the engines' density kernels in ONE launch, `reps` launches between one pair of HIP
    events.  Returns (ms per launch, algorithmic bytes per launch)."""
    lib = engines[0].lib
    handles = (C.c_void_p * len(engines))(*[e._h for e in engines])
    ms, by = C.c_double(0), C.c_double(0)
    rc = lib.nemgpu_profile_density_many(handles, len(engines), int(reps), C.byref(ms), C.byref(by))
    if rc != 0:
        raise NemGpuError("nemgpu_profile_density_many failed (status %d): %s" % (rc, lib.nemgpu_last_error().decode()))
    return ms.value, by.value


def solve(x, nei, k, prop, center, disp, device=0, fast_forward=None, **cfg):
    """Convenience: build an engine, run the whole EM, return full-precision results."""
    n, d = x.shape
    eng = NemEngine(n, d, k, device=device)
    try:
        if fast_forward is not None:
            eng.set_fast_forward(fast_forward)
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        eng.configure(**cfg)
        return eng.run()
    finally:
        eng.close()


# ---- file layer (host only) -------------------------------------------------------------------
def read_inputs(fname, nk):
    """Parse <fname>.str/.dat/.nei/.m with the library's readers.  Returns a dict of numpy arrays."""
    lib = load_library()
    h = C.c_void_p()
    rc = lib.nemio_read(fname.encode() if isinstance(fname, str) else fname, int(nk), C.byref(h))
    if rc != 0:
        raise NemGpuError("nemio_read failed (status %d): %s" % (rc, lib.nemgpu_last_error().decode()))
    try:
        n, d, nnz, mx, pm, ty = (C.c_int() for _ in range(6))
        lib.nemio_sizes(h, C.byref(n), C.byref(d), C.byref(nnz), C.byref(mx), C.byref(pm), C.byref(ty))
        n, d, nnz = n.value, d.value, nnz.value
        wf = (d + 31) // 32
        xbits = np.zeros((n, wf), np.uint32)
        ptr = np.zeros(n + 1, np.int32)
        idx = np.zeros(max(nnz, 1), np.int32)
        w = np.zeros(max(nnz, 1), np.float32)
        prop = np.zeros(nk, np.float32)
        center = np.zeros((nk, d), np.float32)
        disp = np.zeros((nk, d), np.float32)
        lib.nemio_copy(h, _vp(xbits), _vp(ptr), _vp(idx), _vp(w), _vp(prop), _vp(center), _vp(disp))
        x = ((xbits[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(n, wf * 32)[:, :d]
        return dict(n=n, d=d, x=x.astype(np.uint8), xbits=xbits, nei=(ptr, idx[:nnz], w[:nnz]), max_neighs=mx.value,
                    param_mode=pm.value, type=chr(ty.value), prop=prop, center=center, disp=disp)
    finally:
        lib.nemio_free(h)


def write_uf(path, c):
    c = np.ascontiguousarray(c, np.float32)
    return load_library().nemio_write_uf(path.encode(), _vp(c), c.shape[0], c.shape[1])


def write_mf(path, crit6, beta, center, prop, disp):
    center = np.ascontiguousarray(center, np.float32)
    disp = np.ascontiguousarray(disp, np.float32)
    prop = np.ascontiguousarray(prop, np.float32)
    k, d = center.shape
    crit = (C.c_float * 6)(*[float(v) for v in crit6])
    return load_library().nemio_write_mf(path.encode(), crit, C.c_float(beta), d, k, _vp(center), _vp(prop), _vp(disp))

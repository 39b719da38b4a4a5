"""Many independent NEM problems on ONE GPU at once.

PPanGGOLiN partitions pangenomes of more than 500 organisms by solving many 500-organism chunks and voting
(ppanggolin/ppanggolin.py:995-1097); the reference runs them in a multiprocessing.Pool because its nem() is
neither re-entrant nor thread-safe.  One 20 000 x 500 problem fills a fraction of an MI355X (one wave per SIMD in
the density kernel, launch-bound everywhere else), so the native way to run chunks is as ONE batch on the device:

  * ``solve_many``  -- the in-memory route, one library call (``nemgpu_solve_many``): every problem gets its engine
    (bit packing and uploads on ``workers`` threads of the library), then ``nemgpu_run_many`` runs the EM loops of a
    ``group`` of problems in LOCK STEP: each step of the loop -- M-step counts, density, relaxation rounds, criteria
    -- is one launch for all of them (problem = blockIdx.z), one host synchronisation per batch of iterations for
    everybody; results are fetched and engines recycled while the next groups are being built.
  * ``nem_many``    -- the drop-in route (five ASCII files per problem): nem() keeps no global state, so the calls
    run on worker threads, text parsing on different host cores, kernels interleaved on the GPU.
"""
from concurrent.futures import ThreadPoolExecutor

import ctypes as C

import numpy as np

from .engine import (ALGO, CVT, DISP, PROP, STATUS_OK, TIE, Config, NemGpuError, Problem, load_library)
from .nem import nem


def nem_many(calls, workers=8):
    """Run nem(**kw) for every kw of `calls` with `workers` threads; returns the list of return codes."""
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        return list(pool.map(lambda kw: nem(**kw), calls))


def deal_groups(count, group, n_devices):
    """index into the device list of the device that solves each of `count` problems (nemgpu_deal_groups): lock-step
    group g goes to slot g % n_devices"""
    out = (C.c_int * max(count, 1))()
    rc = load_library().nemgpu_deal_groups(int(count), int(group), int(n_devices), out)
    if rc != 0:
        raise NemGpuError("nemgpu_deal_groups failed (status %d)" % rc)
    return [int(out[i]) for i in range(count)]


def solve_many(problems, workers=8, group=32, device=0, algo="ncem", beta=0.5, disper="sk_", propor="pk", cvtest="clas",
               cvthres=1e-8, it_max=100, param_fix=False, tie="hash", seed=0, devices=None):
    """problems = [(x, nei, k, prop, center, disp), ...]; returns their solve() results, each bit-identical to the
    problem solved alone.  ONE library call (nemgpu_solve_many, include/nem_mi355x.h): `workers` threads of the library
    build the engines (bit packing, uploads) and fetch the results, every `group` of problems runs in lock step, and
    while one group runs the next ones are being built.  x: uint8 [n][d] of 0/1, or uint32 bit rows [n][ceil(d/32)]
    (then d is taken from the shape of `center`).  devices=[...]: the groups are dealt round-robin over several devices
    of this process (nemgpu_solve_many_devices; an entry may repeat), each running its share with workers / len(devices)
    library threads -- PPanGGOLiN's pool of chunk workers, with GPUs as the workers."""
    if not problems:
        return []
    lib = load_library()
    cfg = Config(ALGO[algo], beta, DISP[disper], PROP[propor], CVT[cvtest], cvthres, it_max, int(param_fix), TIE[tie], seed)
    arr = (Problem * len(problems))()
    keep, outs = [], []
    addr = lambda a: a.__array_interface__["data"][0]     # (a.ctypes.data builds a ctypes object per call: 1.6 us)
    for q, (x, nei, k, prop, center, disp) in zip(arr, problems):
        bits = x.dtype == np.uint32
        x = np.ascontiguousarray(x, np.uint32 if bits else np.uint8)
        n = x.shape[0]
        d = np.asarray(center).size // int(k) if bits else x.shape[1]
        prop = np.ascontiguousarray(prop, np.float32)
        center = np.ascontiguousarray(center, np.float32).reshape(k, d)
        disp = np.ascontiguousarray(disp, np.float32).reshape(k, d)
        q.n, q.d, q.k = n, d, int(k)
        if bits:
            assert x.shape == (n, (d + 31) // 32)
            q.x_bits = addr(x)
        else:
            q.x_bytes = addr(x)
        keep += [x, prop, center, disp]
        if nei is not None:
            ptr, idx, w = (np.ascontiguousarray(nei[0], np.int32), np.ascontiguousarray(nei[1], np.int32),
                           np.ascontiguousarray(nei[2], np.float32))
            assert ptr.shape == (n + 1,)
            q.nei_ptr, q.nei_idx, q.nei_w = addr(ptr), addr(idx), addr(w)
            keep += [ptr, idx, w]
        q.prop, q.center, q.disp = addr(prop), addr(center), addr(disp)
        # (every element is written by the library when the call succeeds; zero-filling the memberships -- n x k floats --
        # was 70 us per problem, 18 of the 25 ms this function spent outside the library on 256 problems of 20 000 x 500)
        o = dict(prop=np.empty(k, np.float32), center=np.empty((k, d), np.float32), disp=np.empty((k, d), np.float32),
                 nbobs_k=np.empty(k, np.float32), c=np.empty((n, k), np.float32))
        q.out_prop, q.out_center, q.out_disp, q.out_nbobs_k, q.out_c = (addr(o[f]) for f in ("prop", "center", "disp", "nbobs_k", "c"))
        outs.append(o)
    if devices is None:
        rc = lib.nemgpu_solve_many(arr, len(problems), C.byref(cfg), int(device), int(workers), int(group))
    else:
        dl = (C.c_int * len(devices))(*[int(v) for v in devices])
        rc = lib.nemgpu_solve_many_devices(arr, len(problems), C.byref(cfg), dl, len(devices), int(workers), int(group))
    if rc not in (STATUS_OK,):
        raise NemGpuError("nemgpu_solve_many failed (status %d): %s" % (rc, lib.nemgpu_last_error().decode()))
    res = []
    for q, o in zip(arr, outs):
        r = q.result
        meta = dict(status=r.status, iters=r.iters, converged=bool(r.converged), emptyk=r.emptyk,
                    n_zero_density=r.zero_density_sites, first_zero_density_site=r.first_zero_density_site,
                    sweep_rounds=r.sweep_rounds, crit=np.array(list(r.crit), np.float32),
                    loop_seconds=r.loop_seconds, tie_draws=r.tie_draws)
        meta.update(o)
        res.append(meta)
    return res

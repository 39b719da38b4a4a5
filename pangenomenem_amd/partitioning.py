"""`run_partitioning` without the files.

PPanGGOLiN's ``run_partitioning(nem_dir_path, nb_org, beta, free_dispersion, Q, init)``
(ppanggolin/ppanggolin.py:1761-1984) writes nothing itself: it calls ``nem()`` on the five ASCII files
``__write_nem_input_files`` produced (ppanggolin.py:821-930) and parses ``nem_file.uf`` / ``.mf`` back.
``run_partitioning_arrays`` takes the same problem as arrays, runs it through the in-memory C ABI and returns the
same two dicts.  What the text files did to the numbers is kept so that both routes give the same answer: the
posteriors are compared after rounding to the three decimals of ``.uf`` (nem_exe.c:1677) and the parameters after
the ``%g``-style precision of ``.mf`` is NOT needed (mu is only tested for truthiness, epsilon and pi are floats).
"""
import numpy as np

from . import synth
from .engine import NemEngine


def run_partitioning_arrays(x, nei, beta, free_dispersion=False, Q=3, init="param_file_default", names=None,
                            low_disp=0.1, params=None, rng_seed=1, device=0):
    """x uint8 [families, organisms]; nei = CSR (ptr, idx, w) or None; names = family identifiers (default
    fam1..famN).  init: "param_file_default" (PPanGGOLiN's default .m, ppanggolin.py:893-901), "param_file" (give
    params = (prop, center, disp)) or "random" (init_mode INIT_RANDOM, 50 starts, ppanggolin.py:1207).
    Returns ({family: 'P'|'S'|'C'|'U' or class index}, {k: (mu bool list, epsilon list, proportion)})."""
    x = np.ascontiguousarray(x, np.uint8)
    n, d = x.shape
    names = list(names) if names is not None else ["fam%d" % (i + 1) for i in range(n)]
    eng = NemEngine(n, d, Q, device=device)
    try:
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.configure(algo="ncem", beta=beta, disper="skd" if free_dispersion else "sk_", propor="pk", cvtest="clas",
                      cvthres=1e-8, it_max=100, seed=rng_seed)
        if init.startswith("param_file"):
            if init == "param_file_default":
                if Q != 3:
                    raise ValueError("the default parameter file describes 3 classes")
                prop, center, disp = synth.default_init(d, low_disp)
            else:
                prop, center, disp = params
            eng.set_params(prop, center, disp)
            res = eng.run()
        else:
            res = eng.run_random(50, rng_seed)
    finally:
        eng.close()
    return partition_dicts(res, names, Q, init)


def partition_dicts(res, names, Q=3, init="param_file_default"):
    """What run_partitioning makes of a finished NEM run (ppanggolin.py:1886-1980), from the run's full-precision
    results instead of the `.uf` / `.mf` text: res = dict(status, c [n, Q], center [Q, d], disp [Q, d], prop [Q])."""
    n = len(names)
    partitions = ["U"] * n
    all_parameters = {}
    if res["status"] != 0:                                    # empty class: nem() writes no files, everything stays 'U'
        return dict(zip(names, partitions)), all_parameters
    sum_mu, sum_eps = [], []
    for k in range(Q):                                        # ppanggolin.py:1907-1923
        mu_k = [bool(float(v)) for v in res["center"][k]]
        eps_k = [float(v) for v in res["disp"][k]]
        sum_mu.append(sum(mu_k))
        sum_eps.append(sum(eps_k))
        all_parameters[k] = (mu_k, eps_k, float(res["prop"][k]))
    partition = None
    if init == "param_file_default":                          # ppanggolin.py:1925-1957
        persistent_k = sum_mu.index(max(sum_mu))
        shell_k = sum_eps.index(max(sum_eps))
        cloud = list(set([0, 1, 2]) - set([persistent_k, shell_k]))
        partition = {persistent_k: "P", shell_k: "S"}
        if cloud:
            partition[cloud[0]] = "C"
        if partition.get(0) != "P" or partition.get(1) != "S" or partition.get(2) != "C":
            return dict(zip(names, partitions)), all_parameters       # the reference's ValueError branch: all 'U'
    c3 = np.round(res["c"].astype(np.float64), 3)             # what survives the " %5.3f" of .uf
    top = c3.max(axis=1, keepdims=True)
    for i in range(n):                                        # ppanggolin.py:1959-1972
        pos = np.flatnonzero(c3[i] == top[i])
        if init == "param_file_default":
            partitions[i] = "S" if len(pos) > 1 else partition[int(pos[-1])]
        else:
            partitions[i] = int(pos[-1])
    return dict(zip(names, partitions)), all_parameters

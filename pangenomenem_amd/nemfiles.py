"""The NEM ASCII files on the Python side of the boundary.

``write_nem_inputs`` emits ``nem_file.{str,dat,nei,m,index}`` in exactly the text form
PPanGGOLiN's ``__write_nem_input_files`` produces (ppanggolin/ppanggolin.py:821-930);
``read_nem_outputs`` re-states the parsing contract of ``run_partitioning``
(ppanggolin/ppanggolin.py:1886-1980) on ``nem_file.{uf,mf}``.  Host-side text handling only.
"""
import math
import os

import numpy as np


def _num(v):
    """Text of a parameter the way ppanggolin writes it: ``1``, ``0``, ``0.5``, ``0.1`` (str of a
    Python number, ppanggolin.py:895-901); 7 significant digits round-trip a float32."""
    v = float(v)
    return str(int(v)) if v.is_integer() else "%.7g" % v


def write_nem_inputs(nem_dir, x, nei, prop, center, disp, flag=1, names=None, basename="nem_file"):
    """Write <nem_dir>/<basename>.str/.dat/.nei/.m/.index.

    x [n, d] 0/1; nei = CSR (ptr, idx, w) with 0-based neighbours or None; prop[k] (only the first
    k-1 are written, like the reference's .m), center[k, d], disp[k, d]; flag 1 = initial values,
    2 = fixed parameters (nem_exe.c:1002-1014)."""
    os.makedirs(nem_dir, exist_ok=True)
    x = np.asarray(x)
    n, d = x.shape
    base = os.path.join(nem_dir, basename)
    with open(base + ".str", "w") as f:                       # ppanggolin.py:929-930
        f.write("S\t%d\t%d\n" % (n, d))
    with open(base + ".dat", "w") as f:                       # ppanggolin.py:850
        lut = np.array(["0", "1"])
        for row in x:
            f.write("\t".join(lut[row]) + "\n")
    with open(base + ".index", "w") as f:                     # ppanggolin.py:851-852
        for i in range(n):
            f.write("%d\t%s\n" % (i + 1, names[i] if names is not None else "fam%d" % (i + 1)))
    with open(base + ".nei", "w") as f:                       # ppanggolin.py:837, 883-890
        f.write("1\n")
        for i in range(n):
            if nei is None:
                f.write("%d\t0\n" % (i + 1))
                continue
            ptr, idx, w = nei
            b, e = int(ptr[i]), int(ptr[i + 1])
            if e == b:
                f.write("%d\t0\n" % (i + 1))
            else:
                items = [str(i + 1), str(e - b)] + [str(int(j) + 1) for j in idx[b:e]] + \
                        [str(round(float(v), 4)) if not float(v).is_integer() else str(int(v)) for v in w[b:e]]
                f.write("\t".join(items) + "\n")
    k = len(prop)
    center = np.asarray(center).reshape(k, d)
    disp = np.asarray(disp).reshape(k, d)

    with open(base + ".m", "w") as f:                         # ppanggolin.py:892-901
        f.write("%d " % flag)
        f.write(" ".join("%.5g" % float(p) for p in prop[:k - 1]) + " ")
        f.write(" ".join(" ".join(_num(v) for v in center[c]) for c in range(k)) + " ")
        f.write(" ".join(" ".join(_num(v) for v in disp[c]) for c in range(k)))
    return base


def read_nem_outputs(nem_dir, nb_org, q=3, init="param_file_default", basename="nem_file"):
    """Parse <basename>.uf/.mf the way run_partitioning does (ppanggolin.py:1886-1980).

    Returns (partitions list of 'P'/'S'/'C'/'U' or class indices, {k: (mu bool list, eps list, pi)},
    M criterion, BIC).  Raises IOError when the outputs are missing (the reference then labels
    every family 'U')."""
    base = os.path.join(nem_dir, basename)
    with open(base + ".uf") as fu, open(base + ".mf") as fm:
        parameter = fm.readlines()
        m_crit = float(parameter[2].split()[3])               # ppanggolin.py:1900
        n_fam = sum(1 for _ in open(base + ".uf"))
        bic = -2 * m_crit - (q * nb_org * 2 + q - 1) * math.log(n_fam) if n_fam > 0 else float("nan")
        sum_mu, sum_eps, params = [], [], {}
        for k, line in enumerate(parameter[-q:]):             # ppanggolin.py:1907-1923
            vector = line.split()
            mu_k = [bool(float(v)) for v in vector[0:nb_org]]
            eps_k = [float(v) for v in vector[nb_org + 1:]]
            prop = float(vector[nb_org])
            sum_mu.append(sum(mu_k))
            sum_eps.append(sum(eps_k))
            params[k] = (mu_k, eps_k, prop)
        partition = None
        if init == "param_file_default":                      # ppanggolin.py:1925-1957
            persistent_k = sum_mu.index(max(sum_mu))
            shell_k = sum_eps.index(max(sum_eps))
            cloud = list(set([0, 1, 2]) - set([persistent_k, shell_k]))
            partition = {persistent_k: "P", shell_k: "S"}
            if cloud:
                partition[cloud[0]] = "C"
            if partition.get(0) != "P" or partition.get(1) != "S" or partition.get(2) != "C":
                raise ValueError("vector mu_k and epsilon_k value in the mf file are not consistent with the "
                                 "initialisation value in the .m file")
        labels = []
        for line in fu:                                        # ppanggolin.py:1959-1972
            elements = [float(el) for el in line.split()]
            max_prob = max(elements)
            pos = [p for p, prob in enumerate(elements) if prob == max_prob]
            if init == "param_file_default":
                labels.append("S" if len(pos) > 1 else partition[pos.pop()])
            else:
                labels.append(pos.pop())
    return labels, params, m_crit, bic

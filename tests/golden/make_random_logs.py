#!/usr/bin/env python3
"""Golden vectors for the INIT_RANDOM log: tests/golden/random_logs/<case>/.

Runs only where /root/reference exists (this container).  The UNMODIFIED reference (oracle/_ref, ClassifyByNem through
oracle/ref_harness.c with DoLog = TRUE, InitMode = INIT_RANDOM, 50 starts, srandom(seed) instead of time()) writes its
own <Fname>.log -- RandNemAlgo, nem_alg.c:1632-1636, 1662-1669, 1730-1732 -- for a few small seeded problems; a case
stores the inputs, the call arguments and that text (gzip).  Data only: no reference source text is stored.

    python tests/golden/make_random_logs.py
"""
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from pangenomenem_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "random_logs")


def cases():
    x, _ = synth.ushaped_pa_matrix(300, 12, 5)
    yield dict(name="ncem_sk_k3", x=x, nei=synth.contiguity_graph(300, 5), k=3, seed=77, algo="ncem", disper="sk_")
    x, _ = synth.ushaped_pa_matrix(160, 9, 9)
    # (small samples, many classes: some starts end with an empty class, whose sizes the next start's line 0 prints.
    #  The reference itself segfaults on some such inputs -- e.g. 160 x 9, K = 6, nem/skd, seed 4242 -- these do not)
    yield dict(name="nem_skd_k6_empty", x=x, nei=synth.contiguity_graph(160, 9), k=6, seed=1, algo="nem", disper="skd")
    x, _ = synth.ushaped_pa_matrix(100, 8, 9)
    yield dict(name="ncem_sk_k8_empties", x=x, nei=synth.contiguity_graph(100, 9), k=8, seed=6, algo="ncem", disper="sk_")
    x, _ = synth.bernoulli_pa_matrix(256, 10, 3)
    # (three sharp latent classes, K = 5: exact ties between twin classes -- the starts' draws and the tie draws are
    #  one random() stream)
    yield dict(name="ncem_skd_k5_ties", x=x, nei=synth.contiguity_graph(256, 3), k=5, seed=9, algo="ncem", disper="skd")


def main():
    pyoracle.build(ref=True)
    ref = pyoracle.Reference()
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    for c in cases():
        x, nei, k = c["x"], c["nei"], c["k"]
        n, d = x.shape
        tmp = tempfile.mkdtemp(prefix="nemrlog_")
        path = os.path.join(tmp, "ref.log")
        r = ref.classify_random(x, nei, k, n_starts=50, rng_seed=c["seed"], algo=c["algo"], beta=0.5, disper=c["disper"],
                                propor="pk", cvtest="clas", cvthres=1e-8, it_max=100, log_path=path)
        text = open(path, "rb").read()
        shutil.rmtree(tmp)
        out = os.path.join(OUT, c["name"])
        os.makedirs(out)
        ptr, idx, w = nei
        np.savez_compressed(os.path.join(out, "inputs.npz"), xbits=np.packbits(x, axis=1, bitorder="little"), n=n, d=d,
                            k=k, nei_ptr=ptr, nei_idx=idx, nei_w=w)
        with open(os.path.join(out, "ref_log.txt.gz"), "wb") as raw:
            with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as g:
                g.write(text)
        empties = text.count(b" empty at iteration ")
        meta = dict(name=c["name"], n=n, d=d, k=k, seed=c["seed"], algo=c["algo"], disper=c["disper"], beta=0.5,
                    n_starts=50, status=int(r["status"]), best_start=int(r["best_start"]), starts_with_empty_class=empties,
                    lines=text.count(b"\n"))
        with open(os.path.join(out, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print("%-22s n=%4d d=%3d k=%d best=%2d empties=%2d lines=%4d bytes=%d" % (c["name"], n, d, k, r["best_start"], empties,
                                                                                 meta["lines"], len(text)))


if __name__ == "__main__":
    main()

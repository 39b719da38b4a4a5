#!/usr/bin/env python3
"""Generate the Python-side pins under tests/golden/pypins/ from the REAL `run_partitioning`
(/root/reference/ppanggolin/ppanggolin.py:1761-1980), SURVEY.md 8(c).

Runs only in the build container (where /root/reference exists); nothing of the reference travels: what is
stored is data -- the five NEM input files' content as arrays, the text of the `.uf` / `.mf` the reference's own
nem() wrote, and the two dicts run_partitioning returned ({family -> 'P'|'S'|'C'|'U'}, {k -> (mu, epsilon, pi)}).

How the reference is run here:
  * `import nem` is satisfied by a small module object whose `nem(**kwargs)` forwards, through ctypes and with the
    keyword names of nem.pyx:2-14, to the C symbol `nem` of oracle/_ref/libnem_ref.so (the unmodified reference C
    sources compiled by oracle/Makefile);
  * the four third-party modules ppanggolin.py imports at the top but run_partitioning never uses (bidict,
    ordered_set, fa2, highcharts -- not installed here, no network) are registered as empty stand-ins in
    sys.modules; sys.dont_write_bytecode keeps the read-only reference tree untouched.

    python tests/golden/make_pypins.py
"""
import ctypes as C
import gzip
import hashlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "pypins")

from oracle import pyoracle  # noqa: E402
from pangenomenem_amd import nemfiles, synth  # noqa: E402


def reference_run_partitioning():
    pyoracle.build(ref=True)
    lib = C.CDLL(pyoracle.REF_SO)
    lib.nem.restype = C.c_int
    lib.nem.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_float, C.c_char_p, C.c_float, C.c_char_p, C.c_int,
                        C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]

    def nem(Fname, nk, algo, beta, convergence, convergence_th, format, it_max, dolog, model_family, proportion,
            dispersion, init_mode):
        return lib.nem(Fname, nk, algo, beta, convergence, convergence_th, format, it_max, int(dolog), model_family,
                       proportion, dispersion, init_mode)

    mod = types.ModuleType("nem")
    mod.nem = nem
    mod.__all__ = ["nem"]
    sys.modules["nem"] = mod
    for name, attr in (("bidict", "bidict"), ("ordered_set", "OrderedSet"), ("fa2", "ForceAtlas2"),
                       ("highcharts", "Highchart")):
        stub = types.ModuleType(name)
        setattr(stub, attr, type(attr, (), {}))
        sys.modules[name] = stub
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from ppanggolin.ppanggolin import run_partitioning
    return run_partitioning


def cases():
    out = []
    x, _ = synth.bernoulli_pa_matrix(2048, 15, 1)
    nei = synth.contiguity_graph(2048, 1)
    out.append(dict(name="c1_default", x=x, nei=nei, beta=0.5, free_dispersion=False))
    out.append(dict(name="c1_free_dispersion", x=x, nei=nei, beta=0.5, free_dispersion=True))
    out.append(dict(name="c1_beta0", x=x, nei=None, beta=0.0, free_dispersion=False))
    xu, _ = synth.ushaped_pa_matrix(3000, 40, 11)
    out.append(dict(name="ushape_40", x=xu, nei=synth.contiguity_graph(3000, 11), beta=0.5, free_dispersion=False))
    out.append(dict(name="ushape_40_beta1_fd", x=xu, nei=synth.contiguity_graph(3000, 11, chord_frac=0.2), beta=1.0,
                    free_dispersion=True))
    # every family everywhere: the shell and cloud classes empty -> nem() returns 1 and writes no files ->
    # run_partitioning's IOError branch: every family 'U', no parameters (ppanggolin.py:1975-1976)
    out.append(dict(name="empty_class_all_U", x=np.ones((300, 20), np.uint8), nei=synth.contiguity_graph(300, 3),
                    beta=0.5, free_dispersion=False))
    return out


def main():
    run_partitioning = reference_run_partitioning()
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    manifest = []
    for cs in cases():
        x, nei = cs["x"], cs["nei"]
        n, d = x.shape
        prop, center, disp = synth.default_init(d)
        tmp = tempfile.mkdtemp(prefix="pypin_")
        names = ["fam%d" % (i + 1) for i in range(n)]
        nemfiles.write_nem_inputs(tmp, x, nei, prop, center, disp, names=names)
        labels, params = run_partitioning(tmp, d, cs["beta"], cs["free_dispersion"])
        rec = dict(name=cs["name"], n=n, d=d, beta=cs["beta"], free_dispersion=cs["free_dispersion"],
                   x_sha256=hashlib.sha256(np.packbits(x, axis=1, bitorder="little").tobytes()).hexdigest(),
                   labels="".join(labels[nm] for nm in names),
                   params={str(k): dict(mu=[bool(v) for v in p[0]], epsilon=[float(v) for v in p[1]],
                                        proportion=float(p[2])) for k, p in params.items()},
                   has_outputs=os.path.isfile(os.path.join(tmp, "nem_file.uf")))
        cdir = os.path.join(OUT, cs["name"])
        os.makedirs(cdir)
        np.savez_compressed(os.path.join(cdir, "inputs.npz"), xbits=np.packbits(x, axis=1, bitorder="little"),
                            n=n, d=d, has_graph=nei is not None,
                            nei_ptr=nei[0] if nei is not None else np.zeros(1, np.int32),
                            nei_idx=nei[1] if nei is not None else np.zeros(1, np.int32),
                            nei_w=nei[2] if nei is not None else np.zeros(1, np.float32))
        for ext in ("uf", "mf"):
            p = os.path.join(tmp, "nem_file." + ext)
            if os.path.isfile(p):
                with open(p, "rb") as f, gzip.GzipFile(os.path.join(cdir, "ref_%s.txt.gz" % ext), "wb", mtime=0) as g:
                    g.write(f.read())
        with open(os.path.join(cdir, "pin.json"), "w") as f:
            json.dump(rec, f)
        manifest.append(cs["name"])
        shutil.rmtree(tmp)
        print(cs["name"], {c: rec["labels"].count(c) for c in "PSCU"}, "outputs" if rec["has_outputs"] else "no files")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f)


if __name__ == "__main__":
    main()

"""Drop-in for the reference's Cython extension module ``nem`` (ppanggolin/NEM/nem.pyx:1-14):

    from nem import *          # ppanggolin/ppanggolin.py:20
    nem(Fname=b".../nem_file", nk=3, algo=b"ncem", beta=0.5, convergence=b"clas",
        convergence_th=1e-8, format=b"fuzzy", it_max=100, dolog=True, model_family=b"bern",
        proportion=b"pk", dispersion=b"sk_", init_mode=2)      # ppanggolin.py:1814-1826

Same keyword names, same bytes arguments, same int return value; the work is done by the C symbol
``nem`` of lib/libnem_mi355x.so (include/nem_mi355x.h) on the GPU.
"""
from .engine import load_library

__all__ = ["nem"]


def _b(s):
    return s if isinstance(s, bytes) else str(s).encode("ascii")


def nem(Fname, nk, algo, beta, convergence, convergence_th, format, it_max, dolog, model_family, proportion,
        dispersion, init_mode):
    lib = load_library()
    return int(lib.nem(_b(Fname), int(nk), _b(algo), float(beta), _b(convergence), float(convergence_th),
                       _b(format), int(it_max), int(bool(dolog)), _b(model_family), _b(proportion),
                       _b(dispersion), int(init_mode)))

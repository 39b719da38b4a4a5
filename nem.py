"""Top-level module ``nem``: PPanGGOLiN does ``from nem import *`` (ppanggolin/ppanggolin.py:20) and the
reference builds its Cython extension under this name (setup.py:55).  Re-exports the GPU-backed drop-in."""
from pangenomenem_amd.nem import nem  # noqa: F401

__all__ = ["nem"]

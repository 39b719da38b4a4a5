"""Many independent NEM problems on ONE GPU at once.

PPanGGOLiN partitions pangenomes of more than 500 organisms by solving many 500-organism chunks and voting
(ppanggolin/ppanggolin.py:995-1097); the reference runs them in a multiprocessing.Pool because its nem() is
neither re-entrant nor thread-safe.  A 20 000 x 500 problem fills a fraction of an MI355X (one wave per SIMD in
the density kernel, launch-bound everywhere else), so the native way to run chunks is concurrently, one engine
and one HIP stream per worker thread of a single process: the text parsing of the workers runs on different host
cores (the C ABI releases the GIL), their kernels interleave on the GPU.  This library's nem() keeps no global
state, so the same works through the drop-in entry point.
"""
from concurrent.futures import ThreadPoolExecutor

from .engine import solve
from .nem import nem


def nem_many(calls, workers=8):
    """Run nem(**kw) for every kw of `calls` with `workers` threads; returns the list of return codes."""
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        return list(pool.map(lambda kw: nem(**kw), calls))


def solve_many(problems, workers=8, **cfg):
    """In-memory variant: problems = [(x, nei, k, prop, center, disp), ...]; returns the list of solve() results."""
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        return list(pool.map(lambda p: solve(*p, **cfg), problems))

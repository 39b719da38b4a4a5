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

from .engine import solve_many as _solve_many
from .nem import nem


def nem_many(calls, workers=8):
    """Run nem(**kw) for every kw of `calls` with `workers` threads; returns the list of return codes."""
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        return list(pool.map(lambda kw: nem(**kw), calls))


def solve_many(problems, workers=8, group=32, device=0, **cfg):
    """problems = [(x, nei, k, prop, center, disp), ...]; returns their solve() results, each bit-identical to the
    problem solved alone.  ONE library call (nemgpu_solve_many, include/nem_mi355x.h): `workers` threads of the library
    build the engines (bit packing, uploads) and fetch the results, every `group` of problems runs in lock step, and
    while one group runs the next ones are being built."""
    return _solve_many(problems, workers=workers, group=group, device=device, **cfg)

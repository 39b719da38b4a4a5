"""Many independent NEM problems on ONE GPU at once.

PPanGGOLiN partitions pangenomes of more than 500 organisms by solving many 500-organism chunks and voting
(ppanggolin/ppanggolin.py:995-1097); the reference runs them in a multiprocessing.Pool because its nem() is
neither re-entrant nor thread-safe.  One 20 000 x 500 problem fills a fraction of an MI355X (one wave per SIMD in
the density kernel, launch-bound everywhere else), so the native way to run chunks is as ONE batch on the device:

  * ``solve_many``  -- the in-memory route: every problem gets its engine (uploads on ``workers`` host threads), then
    ``nemgpu_run_many`` runs all EM loops in LOCK STEP: each step of the loop -- M-step counts, density, relaxation
    rounds, criteria -- is one launch for all problems (problem = blockIdx.z), one host synchronisation per batch
    of iterations for everybody.  ``group`` bounds how many problems share a launch.
  * ``nem_many``    -- the drop-in route (five ASCII files per problem): nem() keeps no global state, so the calls
    run on worker threads, text parsing on different host cores, kernels interleaved on the GPU.
"""
from concurrent.futures import ThreadPoolExecutor

from .engine import NemEngine, run_many
from .nem import nem


def nem_many(calls, workers=8):
    """Run nem(**kw) for every kw of `calls` with `workers` threads; returns the list of return codes."""
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        return list(pool.map(lambda kw: nem(**kw), calls))


def _build(problem, device, cfg):
    x, nei, k, prop, center, disp = problem
    eng = NemEngine(x.shape[0], x.shape[1], k, device=device)
    try:
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        eng.configure(**cfg)
    except Exception:
        eng.close()
        raise
    return eng


def solve_many(problems, workers=8, group=64, device=0, **cfg):
    """problems = [(x, nei, k, prop, center, disp), ...]; returns their solve() results, each bit-identical to the
    problem solved alone.  workers: host threads that build the engines (bit-packing, uploads) and fetch the results;
    group: problems per lock-step batch.  While one group runs on the GPU the next one is being built."""
    out = []
    group = max(1, int(group))

    def finish(engines, metas):
        def fetch(pair):
            eng, meta = pair
            meta.update(eng.results())
            eng.close()
            return meta
        return list(pool.map(fetch, zip(engines, metas)))

    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        chunks = [problems[g0:g0 + group] for g0 in range(0, len(problems), group)]
        building = [pool.submit(_build, p, device, cfg) for p in chunks[0]] if chunks else []
        for ci in range(len(chunks)):
            engines = [f.result() for f in building]
            building = [pool.submit(_build, p, device, cfg) for p in chunks[ci + 1]] if ci + 1 < len(chunks) else []
            try:
                metas = run_many(engines, fetch=False)
            except Exception:
                for e in engines:
                    e.close()
                raise
            out.extend(finish(engines, metas))
    return out

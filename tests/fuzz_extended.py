#!/usr/bin/env python3
"""Extended differential fuzz (an evidence run next to the test suite, kept under tests/ because it uses the oracle):
the seeded random problems of tests/test_gpu_fuzz.py -- sizes, class counts, both algorithms, all dispersion / proportion models, tie rules, hand-made
parameters -- solved (a) alone by the HIP engine, (b) by the oracle, (c) in lock-step batches of 32 mixed problems
(nemgpu_run_many).  (a) = (b) as the tests define it, (c) = (a) bit for bit.  Prints one JSON object.

    python3 tests/fuzz_extended.py 20000 > gpurun_out/r02_fuzz_extended.json
    python3 tests/fuzz_extended.py 20000 100000 heavy     # round 4: tests/test_gpu_heavy_weights.py::heavy_problem -- edge weights
                                                          # U[1, 600] or sums within +-2 of 709.78 / 88.72; criteria compared too
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pangenomenem_amd.engine import NemEngine, run_many, solve  # noqa: E402
from tests.test_gpu_fuzz import TOL, random_problem  # noqa: E402
from tests.util import maxdiff  # noqa: E402


def same_as_oracle(got, want, cfg):
    if got["status"] != want["status"] or got["iters"] != want["iters"] or got["converged"] != want["converged"]:
        return False
    if want["status"] == 2 and got["emptyk"] != want["emptyk"]:
        return False
    if np.isfinite(want["c"]).all():
        if not np.array_equal(got["c"].argmax(1), want["c"].argmax(1)):
            return False
        if cfg["algo"] == "ncem" and not np.array_equal(got["c"], want["c"]):
            return False
        if maxdiff(got["c"], want["c"]) > TOL:
            return False
    elif not np.array_equal(np.isnan(got["c"]), np.isnan(want["c"])):
        return False
    if any(maxdiff(got[k], want[k]) > TOL for k in ("disp", "prop")):
        return False
    if not np.array_equal(np.nan_to_num(got["center"], nan=-7), np.nan_to_num(want["center"], nan=-7)):
        return False
    if got["n_zero_density"] != want["n_zero_density"]:
        return False
    if cfg.get("_crit"):                                      # (heavy problems: the same non-finite criteria, 1e-5 where finite)
        for g, w in zip(np.asarray(got["crit"], np.float64), np.asarray(want["crit"], np.float64)):
            if np.isnan(w):
                if not np.isnan(g):
                    return False
            elif np.isinf(w):
                if not (np.isinf(g) and (g > 0) == (w > 0)):
                    return False
            elif not (np.isfinite(g) and abs(g - w) <= 1e-5 * max(1.0, abs(w))):
                return False
    return True


def main():
    from oracle.pyoracle import Oracle
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000        # (seeds the test suite does not use)
    heavy = len(sys.argv) > 3 and sys.argv[3] == "heavy"
    if heavy:
        from tests.test_gpu_heavy_weights import heavy_problem
    oracle = Oracle()
    bad_oracle, bad_lockstep, done = [], [], 0
    t0 = time.time()
    for g0 in range(first, first + n_seeds, 32):
        seeds = list(range(g0, min(g0 + 32, first + n_seeds)))
        probs = [heavy_problem(s) if heavy else random_problem(s) for s in seeds]
        solo = []
        for s, (x, nei, k, prop, center, disp, cfg) in zip(seeds, probs):
            got = solve(x, nei, k, prop, center, disp, **cfg)
            want = oracle.run(x, nei, k, prop, center, disp, **cfg)
            if not same_as_oracle(got, want, dict(cfg, _crit=heavy and want["status"] == 0)):
                bad_oracle.append(s)
            solo.append(got)
        engines = []
        for x, nei, k, prop, center, disp, cfg in probs:
            e = NemEngine(x.shape[0], x.shape[1], k)
            e.set_matrix(x); e.set_graph(nei); e.set_params(prop, center, disp); e.configure(**cfg)
            engines.append(e)
        batch = run_many(engines)
        for e in engines:
            e.close()
        for s, a, b in zip(seeds, solo, batch):
            ok = a["status"] == b["status"] and a["iters"] == b["iters"] and a["converged"] == b["converged"]
            for f in ("c", "prop", "center", "disp", "crit"):
                ok = ok and np.array_equal(a[f], b[f], equal_nan=True)
            if not ok:
                bad_lockstep.append(s)
        done += len(seeds)
        if done % 3200 == 0:
            print("[fuzz] %d problems, %.0f s" % (done, time.time() - t0), file=sys.stderr, flush=True)
    print(json.dumps(dict(problems=done, first_seed=first, engine_vs_oracle_mismatches=bad_oracle,
                          lockstep_vs_solo_mismatches=bad_lockstep, seconds=round(time.time() - t0, 1),
                          tolerance_posteriors_and_parameters=TOL,
                          generator="heavy_problem (edge weights U[1, 600] / sums at the exp overflow edges)" if heavy else "random_problem",
                          what="tests/test_gpu_fuzz.py::random_problem; labels / NCEM posteriors / centres bit-exact, fuzzy "
                               "posteriors, epsilon, pi within the tolerance; lock-step batches of 32 mixed problems bit-identical "
                               "to the solo runs"), indent=1))
    return 1 if (bad_oracle or bad_lockstep) else 0


if __name__ == "__main__":
    sys.exit(main())

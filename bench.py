#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NEM hot path (BASELINE.json metric:
"EM iterations/sec + families x organisms/sec at K=3; HBM GB/s vs roofline").

    python bench.py --gpus N --steps K --warmup W

A *step* is one EM iteration (M-step + E-step + convergence test; one pass of the reference loop
nem_alg.c:1789-1840) over the whole presence/absence matrix.  The workload is BASELINE.json
configs[1] per GPU (20 000 families x 500 organisms, K=3, beta=0.5 with a contiguity graph,
ncem / sk_ / pk exactly as ppanggolin.py:1769-1826 calls nem()).  Inputs are resident in HBM
before the timed region.  The timed region restarts the EM from the initial parameters every
`cycle` iterations (cycle = the iterations this workload needs to converge), so the timed steps
are real pre-convergence iterations; the restart (reset + the two initial sweeps) is inside the
timed region and is NOT counted as steps.

For N > 1 (one process per GPU, torch.distributed / RCCL): families are sharded in contiguous
blocks, every rank holds configs[1]-sized shard (weak scaling); per iteration two all-gathers of the
label blocks (one per relaxation round), the second of which also carries every rank's partial integer
M-step statistics -- no separate all-reduce (pangenomenem_amd/distributed.py).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pangenomenem_amd import build as nem_build  # noqa: E402
from pangenomenem_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--families", type=int, default=20000, help="families per GPU")
    ap.add_argument("--organisms", type=int, default=500)
    ap.add_argument("--algo", default="ncem")
    ap.add_argument("--disper", default="sk_", help="dispersion model (sk_ is BASELINE's; skd = PPanGGOLiN's free_dispersion)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist", action="store_true", help="use the sharded torch.distributed path even with 1 GPU")
    ap.add_argument("--cpu-iters", type=int, default=24, help="reference iterations timed for the CPU baseline")
    return ap.parse_args()


def cpu_baseline(x, nei, prop, center, disp, beta, algo, iters):
    """Time the CPU checker on THIS host, 1 core: the compiled reference (oracle/_ref) when it is
    there, else the plain-C port (oracle/nem_oracle.c).  Bounded sample: the same workload, `iters`
    iterations with the convergence test off, minus a 0-iteration run (sort index + initial sweeps)."""
    from oracle import pyoracle
    n, d = x.shape
    if pyoracle.have_reference():
        ref = pyoracle.Reference()
        t0 = ref.classify(x, nei, 3, prop, center, disp, algo=algo, beta=beta, cvtest="none", it_max=0)["seconds"]
        t1 = ref.classify(x, nei, 3, prop, center, disp, algo=algo, beta=beta, cvtest="none", it_max=iters)["seconds"]
        per_it = max(t1 - t0, 1e-9) / iters
        kind = "reference"
    else:
        orc = pyoracle.Oracle()
        per_it = orc.run(x, nei, 3, prop, center, disp, algo=algo, beta=beta, cvtest="none", it_max=iters,
                         tie="hash")["loop_seconds"] / iters
        kind = "port"
    return dict(value=n * d / per_it, unit="cells/s", cores=1, kind=kind,
                sample="%d EM iterations of the same %dx%d workload, convergence test off, loop time only "
                       "(%.3f s/iteration); host has %d cores" % (iters, n, d, per_it, os.cpu_count() or 0),
                em_iterations_per_sec=1.0 / per_it)


def pmc_traffic(kernel, n_loc, d):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_density.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this very command,
    FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, checked with a known-size read).  Only valid
    for the workload it was measured on; null otherwise."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_density.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        for w in rec.get("workloads", []):
            if w.get("families") == n_loc and w.get("organisms") == d and kernel in w.get("kernels", {}):
                return w["kernels"][kernel]["traffic_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    return None


def main():
    args = parse_args()
    # stdout carries exactly ONE line, the JSON record: libraries that print banners to fd 1 (RCCL prints its
    # version when a communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    if world == 1:
        nem_build.build()                          # no-op when the in-tree library is up to date
    n_loc, d, k, beta = args.families, args.organisms, 3, 0.5
    n_tot = n_loc * world

    if world == 1 and not args.dist:
        from pangenomenem_amd.engine import NemEngine
        x, _ = synth.bernoulli_pa_matrix(n_tot, d, 2)
        nei = synth.contiguity_graph(n_tot, 2)
        prop, center, disp = synth.default_init(d)
        eng = NemEngine(n_tot, d, k, device=0)
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        # how many iterations does this workload need?  (reference call: clas, 1e-8, it_max 100)
        eng.configure(algo=args.algo, beta=beta, disper=args.disper, propor="pk", cvtest="clas", cvthres=1e-8, it_max=100)
        first = eng.run()
        # restart period of the timed loop: at least 5 iterations (what the reference needs on this
        # config per SURVEY.md §6) so the restart overhead is amortised the way a real solve amortises it
        cycle = max(5, int(first["iters"]))
        eng.configure(algo=args.algo, beta=beta, disper=args.disper, propor="pk", cvtest="none", it_max=100)

        def run_steps(count):
            done = 0
            rounds = 0
            while done < count:
                m = min(cycle, count - done)
                rounds += eng.restart_iterate(m)["sweep_rounds"]      # reset + initial sweeps + m iterations
                done += m
            return rounds

        run_steps(args.warmup)
        t0 = time.perf_counter()
        run_steps(args.steps)
        elapsed = time.perf_counter() - t0          # iterate() synchronises the stream before returning
        # E1 kernel duration: HIP events on the engine's stream around individual launches, on the state the
        # timed region just left (the timed region itself replays captured graphs, which carry no events)
        prof = eng.profile_density(100)
        dt_max = elapsed
        extra = dict(iters_to_converge=int(first["iters"]))
    else:
        from pangenomenem_amd import distributed as nd
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "RANK" not in os.environ:              # plain `python bench.py --dist`: a 1-rank group
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if world > 1:                             # one rank checks / rebuilds the library, the others wait for it
            if rank == 0:
                nem_build.build()
            dist.barrier()
        job = nd.ShardedNem.synthetic(n_loc, d, k, beta, rank, world, local_rank, algo=args.algo)
        cycle = job.iters_to_converge()
        job.run_steps(args.warmup, cycle)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.run_steps(args.steps, cycle)
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        prof = job.eng.profile_density(100)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())
        extra = dict(iters_to_converge_floor5=cycle)
        x = nei = None

    if rank == 0:
        ms_per_step = dt_max * 1e3 / args.steps
        cells_per_s = n_tot * d * args.steps / dt_max
        achieved = prof["algorithmic_bytes_per_launch"] / (prof["density_ms_avg"] * 1e-3) / 1e9 \
            if prof["density_ms_avg"] > 0 else 0.0
        out = {
            "metric": "em_family_x_organism_cells_per_sec",
            "value": cells_per_s,
            "unit": "cells/s",
            "em_iterations_per_sec": args.steps / dt_max,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 chains with f64 intermediates (reference arithmetic); int32 popcounts in the M-step",
            "data": "synthetic",
            "config": {
                "workload": "%s per GPU: %d families x %d organisms, K=3, beta=0.5, contiguity graph "
                            "(path + 5%% chords, weights 1..8), %s/%s/pk, default .m init"
                            % ({(20000, 500): "BASELINE configs[1]", (50000, 1000): "BASELINE configs[2] shape",
                                (200000, 5000): "BASELINE configs[3] shape"}.get((n_loc, d), "custom shape")
                               if args.disper == "sk_" else "custom model",
                               n_loc, d, args.algo, args.disper),
                "families_total": n_tot, "organisms": d, "K": k, "beta": beta,
                "cycle_iterations": cycle,
                "parallelism": "1 GPU" if world == 1 else "families sharded over %d GPUs; per EM iteration two RCCL "
                                                           "all-gathers of the label blocks (the second also carries the "
                                                           "ranks' int32 M-step statistics)" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": prof.get("kernel", "k_density") + " (E1 Bernoulli log-density chains)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(prof.get("kernel", "k_density"), n_loc, d),
                "algorithmic_bytes_per_launch": prof["algorithmic_bytes_per_launch"],
                "avg_launch_ms": prof["density_ms_avg"],
                "launches_timed": prof["density_launches"],
                "whole_iteration_algorithmic_GBps": None,
                "note": "HBM is the nominal bound (SURVEY.md 8d: bit scans, no dense contraction, no MFMA). The kernel is a "
                        "dependent chain per (family, class) that the reference's float rounding forbids re-associating; "
                        "at configs[1] one launch moves 2 MB (0.26 us at peak) and is latency-bound, see DESIGN.md section 4",
            },
        }
        # whole-iteration algorithmic traffic (SURVEY.md §8d formula), for context
        nnz = int(nei[0][-1]) if nei is not None else int(2.1 * n_tot)
        bytes_iter = 2 * ((d + 31) // 32) * 4 * n_tot + 12 * n_tot * k + (8 * nnz + 4 * k * nnz + 4 * (n_tot + 1)) \
            + 16 * k * d
        out["roofline"]["whole_iteration_algorithmic_GBps"] = bytes_iter * args.steps / dt_max / 1e9
        out.update(extra)
        if world == 1 and not args.dist and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(x, nei, prop, center, disp, beta, args.algo, args.cpu_iters)
                out["speedup_vs_cpu_baseline"] = cells_per_s / out["cpu_baseline"]["value"]
            except Exception as exc:   # the checker is optional on the box; the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 1, "kind": "unavailable",
                                       "sample": "failed: %r" % (exc,)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if world > 1 or args.dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

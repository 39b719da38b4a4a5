#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NEM hot path (BASELINE.json metric:
"EM iterations/sec + families x organisms/sec at K=3; HBM GB/s vs roofline").

    python bench.py --gpus N --steps K --warmup W

A *step* is one EM iteration (M-step + E-step + convergence test; one pass of the reference loop
nem_alg.c:1789-1840) over the whole presence/absence matrix.  Inputs are resident in HBM before the timed
region.  The EM is restarted from PPanGGOLiN's default initial parameters every `cycle` iterations, cycle =
the iterations the workload needs to converge under the reference's own test (clas, 1e-8) -- so every timed step
is an iteration a real solve performs; the restart (reset + the two initial sweeps) is inside the timed region
and is NOT counted as steps.  The matrix has a U-shaped family-frequency spectrum (synth.ushaped_pa_matrix): on
the three well-separated latent classes of SURVEY.md 8(d) NEM is at its fixed point after ONE iteration, which
would time the cheapest path (--spectrum latent3 selects that generator).

N = 1 (default): BASELINE configs[1], 20 000 families x 500 organisms, K=3, beta=0.5, contiguity graph,
ncem / sk_ / pk exactly as ppanggolin.py:1769-1826 calls nem().  The JSON line also carries `north_star_target`:
the 50 000 x 1 000 problem of BASELINE.json's north_star on this GPU next to the compiled reference's own loop on
one host core, timed in the same run (--no-north-star skips it).

N > 1: one process per GPU (torch.distributed over RCCL).  `python bench.py --gpus N` starts its own ranks
(a torch.distributed.run child, before this process touches the GPU); under an existing launcher (RANK /
WORLD_SIZE in the environment) it is one of the ranks.  Default: STRONG scaling of BASELINE configs[2] -- the
fixed 50 000 x 1 000 problem, families sharded in contiguous blocks; --scaling weak gives every rank a
configs[1]-sized shard instead.  Per EM iteration two all-gathers of the label blocks (one per relaxation round),
the second of which also carries every rank's partial integer M-step statistics -- no separate all-reduce
(pangenomenem_amd/distributed.py).

Before the timed region every batch shape it will enqueue is captured into its hipGraph (independent of
--warmup); `graphs_primed` in the output asserts that nothing was captured or sent as plain launches while timing.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

SHAPES = {(20000, 500): "BASELINE configs[1]", (50000, 1000): "BASELINE configs[2] shape",
          (200000, 5000): "BASELINE configs[3] shape"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--families", type=int, default=None,
                    help="families: of the problem (1 GPU, strong scaling) or per GPU (weak scaling)")
    ap.add_argument("--organisms", type=int, default=None)
    ap.add_argument("--k", type=int, default=None,
                    help="classes (default 3); given: BASELINE configs[4], the K sweep -- 10-latent-group data, K-class .m, skd")
    ap.add_argument("--algo", default="ncem")
    ap.add_argument("--disper", default=None, help="dispersion model (default sk_, BASELINE's; skd for --k != 3)")
    ap.add_argument("--spectrum", default="ushape", choices=["ushape", "latent3"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: host-staged collectives, several ranks may share one GPU (rehearsal only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--dist", action="store_true", help="use the sharded torch.distributed path even with 1 GPU")
    ap.add_argument("--cpu-iters", type=int, default=24, help="reference iterations timed for the CPU baseline")
    ap.add_argument("--ns-cpu-iters", type=int, default=2, help="reference iterations timed at 50 000 x 1 000")
    ap.add_argument("--ns-steps", type=int, default=220, help="GPU iterations timed at 50 000 x 1 000")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher around it
# ----------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """Start N ranks with torch.distributed.run as a CHILD process and pass its stdout (the one JSON line) on.
    Runs before this process imports torch or touches HIP: a process that has initialised the GPU must not exec."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    out = proc.stdout.decode(errors="replace")
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
    else:
        sys.stderr.write(out)
    sys.stdout.flush()
    return proc.returncode if proc.returncode != 0 or lines else 1


# ----------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------
def make_workload(n, d, k, spectrum, seed, ksweep=False):
    """(x, nei, prop, center, disp, disper, description)"""
    from pangenomenem_amd import synth
    nei = synth.contiguity_graph(n, seed)
    if ksweep:
        x, _ = synth.grouped_pa_matrix(n, d, 5, groups=10)
        prop, center, disp = synth.kclass_init(x, k)
        return x, nei, prop, center, disp, "skd", "10-latent-group matrix (SURVEY.md 8d, C5), deterministic K-class .m"
    if spectrum == "ushape":
        x, _ = synth.ushaped_pa_matrix(n, d, seed)
        what = "U-shaped family-frequency spectrum (Beta(0.3, 0.3) per family)"
    else:
        x, _ = synth.bernoulli_pa_matrix(n, d, seed)
        what = "3 latent classes P/S/C 0.30/0.20/0.50 at p = 0.97/0.5/0.03 (SURVEY.md 8d)"
    prop, center, disp = synth.default_init(d)
    return x, nei, prop, center, disp, "sk_", what + ", default .m init"


def whole_iteration_bytes(n, d, k, nnz):
    """algorithmic bytes of one EM iteration, SURVEY.md 8(d)"""
    return 2 * ((d + 31) // 32) * 4 * n + 12 * n * k + (8 * nnz + 4 * k * nnz + 4 * (n + 1)) + 16 * k * d


class EngineRun:
    """One problem resident on one GPU, timed the way the module docstring says."""

    def __init__(self, x, nei, k, prop, center, disp, algo, beta, disper, device=0):
        from pangenomenem_amd.engine import NemEngine
        n, d = x.shape
        self.algo, self.beta, self.disper = algo, beta, disper
        self.eng = eng = NemEngine(n, d, k, device=device)
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        # how many iterations does this workload need?  (the reference's call: clas, 1e-8, it_max 100)
        eng.configure(algo=algo, beta=beta, disper=disper, propor="pk", cvtest="clas", cvthres=1e-8, it_max=100)
        first = eng.run()
        self.first = first
        # (a run that ends with an empty class -- K = 9 of the K sweep -- stops in the M-step of its last iteration:
        #  the iterations before it are the ones that can be timed)
        self.cycle = max(1, int(first["iters"]) - (1 if first["status"] != 0 else 0))
        eng.configure(algo=algo, beta=beta, disper=disper, propor="pk", cvtest="none", it_max=100)
        eng.set_graph_policy(True)

    def run_steps(self, count):
        done, rounds = 0, 0
        while done < count:
            m = min(self.cycle, count - done)
            rounds += self.eng.restart_iterate(m)["sweep_rounds"]      # reset + initial sweeps + m iterations
            done += m
        return rounds

    def timed(self, steps, warmup):
        eng = self.eng
        # every batch shape the warm-up and the timed region enqueue gets its hipGraph NOW, whatever --warmup is
        for m in sorted({self.cycle, steps % self.cycle, warmup % self.cycle} - {0}):
            eng.restart_iterate(m)
        self.run_steps(warmup)
        c0 = eng.graph_counters()
        t0 = time.perf_counter()
        rounds = self.run_steps(steps)
        elapsed = time.perf_counter() - t0          # restart_iterate() synchronises the stream before returning
        c1 = eng.graph_counters()
        restarts = (steps + self.cycle - 1) // self.cycle
        info = dict(
            graphs_primed=(c1["captured"] == c0["captured"] and c1["plain"] == c0["plain"]),
            graph_replays_timed=c1["replayed"] - c0["replayed"],
            host_finished_sweeps_timed=c1["host_finished_sweeps"] - c0["host_finished_sweeps"],
            iters_to_converge=int(self.first["iters"]), cycle_iterations=self.cycle, run_status=int(self.first["status"]),
            # two initial sweeps per restart ride in the total; the rest are the iterations' own
            sweep_rounds_per_iteration=(rounds - 2 * restarts) / max(steps, 1) if rounds else None,
        )
        return elapsed, info


def cpu_baseline(x, nei, k, prop, center, disp, beta, algo, disper, iters):
    """Time the CPU checker on THIS host, 1 core: the compiled reference (oracle/_ref) when it is there, else the
    plain-C port (oracle/nem_oracle.c).  Bounded sample: the same workload, `iters` iterations with the convergence
    test off, minus a 0-iteration run (sort index + initial sweeps)."""
    from oracle import pyoracle
    n, d = x.shape
    if pyoracle.have_reference():
        ref = pyoracle.Reference()
        kw = dict(algo=algo, beta=beta, disper=disper, cvtest="none")
        t0 = ref.classify(x, nei, k, prop, center, disp, it_max=0, **kw)["seconds"]
        t1 = ref.classify(x, nei, k, prop, center, disp, it_max=iters, **kw)["seconds"]
        per_it = max(t1 - t0, 1e-9) / iters
        kind = "reference"
    else:
        orc = pyoracle.Oracle()
        per_it = orc.run(x, nei, k, prop, center, disp, algo=algo, beta=beta, disper=disper, cvtest="none",
                         it_max=iters, tie="hash")["loop_seconds"] / iters
        kind = "port"
    return dict(value=n * d / per_it, unit="cells/s", cores=1, kind=kind,
                sample="%d EM iterations of the same %dx%d workload, convergence test off, loop time only "
                       "(%.3f s/iteration); host has %d cores" % (iters, n, d, per_it, os.cpu_count() or 0),
                em_iterations_per_sec=1.0 / per_it, seconds_per_iteration=per_it)


def pmc_traffic(kernel, n_loc, d):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r0N_pmc_density.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this very command,
    FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, checked with a known-size read).  Only valid
    for the workload it was measured on; null otherwise."""
    for name in ("r02_pmc_density.json", "r01_pmc_density.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            for w in rec.get("workloads", []):
                if w.get("families") == n_loc and w.get("organisms") == d and kernel in w.get("kernels", {}):
                    return w["kernels"][kernel]["traffic_bytes_per_launch"]
        except (OSError, ValueError):
            pass
    return None


def roofline_block(prof, n_loc, d):
    achieved = prof["algorithmic_bytes_per_launch"] / (prof["density_ms_avg"] * 1e-3) / 1e9 \
        if prof["density_ms_avg"] > 0 else 0.0
    return {
        "bound": "hbm",
        "kernel": prof.get("kernel", "k_density") + " (E1 Bernoulli log-density chains)",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc_traffic(prof.get("kernel", "k_density"), n_loc, d),
        "algorithmic_bytes_per_launch": prof["algorithmic_bytes_per_launch"],
        "avg_launch_ms": prof["density_ms_avg"], "launches_timed": prof["density_launches"],
        "note": "HBM is the nominal bound (SURVEY.md 8d: bit scans, no dense contraction, no MFMA). The kernel is a "
                "dependent chain per (family, class) that the reference's float rounding forbids re-associating; "
                "one launch at configs[1] moves 2 MB (0.26 us at peak) and is latency-bound, see DESIGN.md section 4",
    }


def north_star_target(args):
    """BASELINE.json north_star: >= 50x the in-repo C NEM on 50 000 x 1 000 at K=3, beta=0.5 -- GPU and the
    compiled reference (1 host core), same workload, same run."""
    n, d, k, beta = 50000, 1000, 3, 0.5
    x, nei, prop, center, disp, disper, what = make_workload(n, d, k, args.spectrum, 3)
    run = EngineRun(x, nei, k, prop, center, disp, "ncem", beta, disper)
    steps = max(run.cycle, (args.ns_steps // run.cycle) * run.cycle)
    elapsed, info = run.timed(steps, run.cycle)
    prof = run.eng.profile_density(50)
    ms_it = elapsed * 1e3 / steps
    out = {
        "workload": "50000 families x 1000 organisms, K=3, beta=0.5, contiguity graph, ncem/sk_/pk; " + what,
        "gpu_ms_per_iteration": ms_it, "gpu_em_iterations_per_sec": 1e3 / ms_it,
        "gpu_cells_per_sec": n * d * 1e3 / ms_it, "gpu_steps_timed": steps,
        "e1_roofline_frac": roofline_block(prof, n, d)["frac"], "e1_avg_launch_ms": prof["density_ms_avg"],
        "whole_iteration_algorithmic_GBps": whole_iteration_bytes(n, d, k, int(nei[0][-1])) / (ms_it * 1e-3) / 1e9,
        "target_speedup": 50.0,
    }
    out.update(info)
    run.eng.close()
    try:
        cpu = cpu_baseline(x, nei, k, prop, center, disp, beta, "ncem", disper, args.ns_cpu_iters)
        out["reference_cpu_seconds_per_iteration"] = cpu["seconds_per_iteration"]
        out["reference_cpu"] = {kk: cpu[kk] for kk in ("kind", "cores", "sample")}
        out["speedup_vs_reference_cpu"] = cpu["seconds_per_iteration"] * 1e3 / ms_it
        out["meets_target"] = bool(out["speedup_vs_reference_cpu"] >= 50.0)
    except Exception as exc:
        out["reference_cpu"] = {"kind": "unavailable", "sample": "failed: %r" % (exc,)}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # stdout carries exactly ONE line, the JSON record: libraries that print banners to fd 1 (RCCL prints its
    # version when a communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    from pangenomenem_amd import build as nem_build
    ksweep = args.k is not None
    k, beta = (args.k if ksweep else 3), 0.5
    sharded = world > 1 or args.dist
    if args.disper is None:
        args.disper = "skd" if ksweep else "sk_"
    if sharded and ksweep:
        raise SystemExit("the sharded path benchmarks K = 3 (BASELINE configs[2])")

    if not sharded:
        nem_build.build()                          # no-op when the in-tree library is up to date
        n_tot, d = args.families or 20000, args.organisms or 500
        x, nei, prop, center, disp, _, what = make_workload(n_tot, d, k, args.spectrum, 2, ksweep)
        run = EngineRun(x, nei, k, prop, center, disp, args.algo, beta, args.disper)
        dt_max, extra = run.timed(args.steps, args.warmup)
        # E1 kernel duration: HIP events on the engine's stream around individual launches, on the state the
        # timed region just left (the timed region itself replays captured graphs, which carry no events)
        prof = run.eng.profile_density(100)
        n_loc, nnz = n_tot, int(nei[0][-1])
        scaling, parallelism = "weak", "1 GPU"
    else:
        from pangenomenem_amd import distributed as nd
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        device = local_rank % max(ndev, 1) if args.backend == "gloo" else local_rank
        torch.cuda.set_device(device)
        if "RANK" not in os.environ:              # plain `python bench.py --dist`: a 1-rank group
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
        if rank == 0:                             # one rank checks / rebuilds the library, the others wait for it
            nem_build.build()
        dist.barrier()
        if args.scaling == "strong":
            n_tot, d = args.families or 50000, args.organisms or 1000
            seed = 3 if (n_tot, d) == (50000, 1000) else 2
        else:
            d = args.organisms or 500
            n_tot, seed = (args.families or 20000) * world, 2
        what = make_workload(64, 32, 3, args.spectrum, 1)[6]
        job = nd.ShardedNem.synthetic(n_tot, d, k, beta, rank, world, device, algo=args.algo,
                                      spectrum=args.spectrum, seed=seed)
        cycle = job.iters_to_converge()
        # every batch shape of the timed region goes through the driver once or twice before the clock starts
        # (its graph, where graphs are on, is captured the second time a shape is seen)
        for m in sorted({cycle, args.steps % cycle, args.warmup % cycle} - {0}):
            job.run_steps(2 * m, m)
        job.run_steps(args.warmup, cycle)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.run_steps(args.steps, cycle)
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        prof = job.eng.profile_density(100)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())
        n_loc = job.hi - job.lo
        nnz = int(2.1 * n_tot)
        scaling = args.scaling if world > 1 else "weak"
        extra = dict(iters_to_converge=cycle, cycle_iterations=cycle, native_rccl=bool(job.native),
                     rccl_ranks=job.eng.rccl_ranks(), backend=args.backend,
                     batch_graphs=bool(job.use_graphs))
        if world > 1 and args.scaling == "strong" and rank == 0:
            # the same problem on ONE GPU, same run: what the strong-scaling values are to be compared with (the
            # N = 1 default of this script is configs[1], another problem)
            try:
                x1, nei1, p1, c1, d1, _, _ = make_workload(n_tot, d, k, args.spectrum, seed)
                solo = EngineRun(x1, nei1, k, p1, c1, d1, args.algo, beta, "sk_", device=device)
                st = max(solo.cycle, (min(args.steps, 220) // solo.cycle) * solo.cycle)
                t1, _ = solo.timed(st, solo.cycle)
                extra["single_gpu_same_workload"] = dict(ms_per_step=t1 * 1e3 / st, value=n_tot * d * st / t1,
                                                         em_iterations_per_sec=st / t1, steps=st)
                extra["speedup_vs_single_gpu_same_workload"] = (t1 / st) / (dt_max / args.steps)
                solo.eng.close()
            except Exception as exc:
                extra["single_gpu_same_workload"] = {"error": repr(exc)}
        if world > 1:
            dist.barrier()
        parallelism = ("families sharded over %d GPUs in contiguous blocks (%s scaling); per EM iteration two RCCL "
                       "all-gathers of the label blocks, the second also carrying the ranks' int32 M-step statistics"
                       % (world, scaling)) if world > 1 else "1 GPU through the sharded driver"

    if rank == 0:
        ms_per_step = dt_max * 1e3 / args.steps
        cells_per_s = n_tot * d * args.steps / dt_max
        shape = ("BASELINE configs[4] (K sweep)" if (n_tot, d) == (20000, 500) and args.disper == "skd" else "custom model") \
            if ksweep else (SHAPES.get((n_tot, d), "custom shape") if args.disper == "sk_" else "custom model")
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
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32 chains with f64 intermediates (reference arithmetic); int32 popcounts in the M-step",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d families x %d organisms, K=%d, beta=0.5, contiguity graph (path + 5%% chords, "
                            "weights 1..8), %s/%s/pk; %s" % (shape, n_tot, d, k, args.algo, args.disper, what),
                "families_total": n_tot, "families_per_gpu": n_loc, "organisms": d, "K": k, "beta": beta,
                "cycle_iterations": extra["cycle_iterations"],
                "parallelism": parallelism,
            },
            "roofline": roofline_block(prof, n_loc, d),
        }
        out["roofline"]["whole_iteration_algorithmic_GBps"] = \
            whole_iteration_bytes(n_tot, d, k, nnz) * args.steps / dt_max / 1e9
        out.update(extra)
        if not sharded and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(x, nei, k, prop, center, disp, beta, args.algo, args.disper,
                                                   args.cpu_iters)
                out["speedup_vs_cpu_baseline"] = cells_per_s / out["cpu_baseline"]["value"]
            except Exception as exc:   # the checker is optional on the box; the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 1, "kind": "unavailable",
                                       "sample": "failed: %r" % (exc,)}
        if not sharded and not args.no_north_star and not ksweep:
            try:
                run.eng.close()
                out["north_star_target"] = north_star_target(args)
            except Exception as exc:
                out["north_star_target"] = {"error": repr(exc)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if sharded:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

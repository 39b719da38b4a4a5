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

Timing: W warm-up steps, then `--repeats` (25) blocks of EXACTLY K steps, each bracketed by a barrier and a device
synchronisation on both sides, maximum over ranks per block; `ms_per_step` / `value` come from the MEDIAN block,
`ms_per_step_min` / `_max` give the spread (a block of 20 steps lasts under a millisecond: one scheduler hiccup
would otherwise move the headline by double digits).

N = 1 (default): BASELINE configs[1], 20 000 families x 500 organisms, K=3, beta=0.5, contiguity graph,
ncem / sk_ / pk exactly as ppanggolin.py:1769-1826 calls nem().  The JSON line also carries `north_star_target`:
the 50 000 x 1 000 problem of BASELINE.json's north_star on this GPU next to the compiled reference's own loop on
one host core, timed in the same run (--no-north-star skips it).

N > 1: one process per GPU (torch.distributed over RCCL).  `python bench.py --gpus N` starts its own ranks
(a torch.distributed.run child, before this process touches the GPU); under an existing launcher (RANK /
WORLD_SIZE in the environment) it is one of the ranks.  --scaling selects what the N GPUs do:
  strong   (default) BASELINE configs[2]: ONE 50 000 x 1 000 problem (--families / --organisms: any), families
           sharded in contiguous blocks; per EM iteration two RCCL all-gathers of the label blocks (one per relaxation
           round), the second also carrying every rank's partial integer M-step statistics -- no separate all-reduce
           (pangenomenem_amd/distributed.py).  Rank 0 also times the same problem on ONE GPU in the same run
           (`single_gpu_same_workload`);
  weak     the same sharded EM with a configs[1]-sized shard per rank (one problem of N x 20 000 families);
  replicas N independent configs[1]-sized problems, one per GPU, NO collective on the data path -- the reference's own
           form of parallelism (one NEM problem per organism chunk, ppanggolin.py:1039-1095).
An N > 1 line also carries `collective` (time of one all-gather, measured on the job's own path) and, unless
--no-extras, `also`: the replicas figure and the strong-scaling figure of 200 000 x 5 000 (BASELINE configs[3]), the
shape large enough for sharding to pay, each with its own single-GPU reference.

Before a timed region every batch shape it will enqueue is captured into its hipGraph (independent of --warmup);
`graphs_primed` in the output asserts that nothing was captured or sent as plain launches while timing.

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
    ap.add_argument("--repeats", type=int, default=25, help="timed blocks of --steps steps each (median reported)")
    ap.add_argument("--families", type=int, default=None,
                    help="families: of the problem (1 GPU, strong scaling) or per GPU (weak scaling, replicas)")
    ap.add_argument("--organisms", type=int, default=None)
    ap.add_argument("--k", type=int, default=None,
                    help="classes (default 3); given: BASELINE configs[4], the K sweep -- 10-latent-group data, K-class .m, skd")
    ap.add_argument("--algo", default="ncem")
    ap.add_argument("--disper", default=None, help="dispersion model (default sk_, BASELINE's; skd for --k != 3)")
    ap.add_argument("--spectrum", default="ushape", choices=["ushape", "latent3"])
    ap.add_argument("--weights", default="small", choices=["small", "coverage", "adjacency"],
                    help="edge weights of the contiguity graph: small = integers 1..8 (SURVEY.md 8d); coverage = integers "
                         "uniform in 1..D, the range of what PPanGGOLiN itself writes (ppanggolin.py:866-878: the organisms "
                         "carrying the adjacency); adjacency = such counts with a pangenome graph's structure (a family's "
                         "weights add up to at most twice its organisms: synth.contiguity_graph)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak", "replicas"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: host-staged collectives, several ranks may share one GPU (rehearsal only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the `also` block (N = 1: coverage weights, 64 problems in lock "
                    "step, 50 random starts on the tie stream, 200 000 x 5 000, 256 whole chunks; N > 1: replicas, 200 000 x 5 000)")
    ap.add_argument("--extras-strong-shape", default="200000x5000", help="families x organisms of the `also` block's strong-scaling problem")
    ap.add_argument("--dist", action="store_true", help="use the sharded torch.distributed path even with 1 GPU")
    ap.add_argument("--cpu-iters", type=int, default=24, help="reference iterations timed for the CPU baseline")
    ap.add_argument("--ns-cpu-iters", type=int, default=4, help="reference iterations timed at 50 000 x 1 000")
    ap.add_argument("--ns-steps", type=int, default=220, help="GPU iterations per block at 50 000 x 1 000")
    ap.add_argument("--phase-timeout", type=float, default=300.0,
                    help="N > 1: seconds a phase (a mode's measurement) may take before the job's watchdog ends it and rank 0 "
                         "prints the record it has")
    ap.add_argument("--watchdog-selftest", action="store_true", help=argparse.SUPPRESS)
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
def make_workload(n, d, k, spectrum, seed, ksweep=False, weights="small"):
    """(x, nei, prop, center, disp, disper, description)"""
    from pangenomenem_amd import synth

    def graph(x):
        return synth.contiguity_graph(n, seed, weights=weights, d=d, counts=x.sum(axis=1) if weights == "adjacency" else None)
    if ksweep:
        x, _ = synth.grouped_pa_matrix(n, d, 5, groups=10)
        prop, center, disp = synth.kclass_init(x, k)
        return x, graph(x), prop, center, disp, "skd", "10-latent-group matrix (SURVEY.md 8d, C5), deterministic K-class .m"
    if spectrum == "ushape":
        x, _ = synth.ushaped_pa_matrix(n, d, seed)
        what = "U-shaped family-frequency spectrum (Beta(0.3, 0.3) per family)"
    else:
        x, _ = synth.bernoulli_pa_matrix(n, d, seed)
        what = "3 latent classes P/S/C 0.30/0.20/0.50 at p = 0.97/0.5/0.03 (SURVEY.md 8d)"
    prop, center, disp = synth.default_init(d)
    return x, graph(x), prop, center, disp, "sk_", what + ", default .m init"


def whole_iteration_bytes(n, d, k, nnz):
    """algorithmic bytes of one EM iteration, SURVEY.md 8(d)"""
    return 2 * ((d + 31) // 32) * 4 * n + 12 * n * k + (8 * nnz + 4 * k * nnz + 4 * (n + 1)) + 16 * k * d


def median(v):
    s = sorted(v)
    m = len(s) // 2
    return s[m] if len(s) % 2 else 0.5 * (s[m - 1] + s[m])


def timing_fields(blocks, steps):
    """blocks: seconds of each timed block of `steps` steps (max over ranks already)"""
    med = median(blocks)
    return dict(ms_per_step=med * 1e3 / steps, repeats=len(blocks), ms_per_step_min=min(blocks) * 1e3 / steps,
                ms_per_step_max=max(blocks) * 1e3 / steps, timed_region_s=sum(blocks)), med


class EngineRun:
    """One problem resident on one GPU, timed the way the module docstring says."""

    def __init__(self, x, nei, k, prop, center, disp, algo, beta, disper, device=0, tie="hash"):
        from pangenomenem_amd.engine import NemEngine
        n, d = x.shape
        self.algo, self.beta, self.disper = algo, beta, disper
        self.eng = eng = NemEngine(n, d, k, device=device)
        eng.set_matrix(x)
        eng.set_graph(nei)
        eng.set_params(prop, center, disp)
        # how many iterations does this workload need?  (the reference's call: clas, 1e-8, it_max 100)
        eng.configure(algo=algo, beta=beta, disper=disper, propor="pk", cvtest="clas", cvthres=1e-8, it_max=100, tie=tie, seed=1)
        first = eng.run()
        self.first = first
        # (a run that ends with an empty class -- K = 9 of the K sweep -- stops in the M-step of its last iteration:
        #  the iterations before it are the ones that can be timed)
        self.cycle = max(1, int(first["iters"]) - (1 if first["status"] != 0 else 0))
        eng.configure(algo=algo, beta=beta, disper=disper, propor="pk", cvtest="none", it_max=100, tie=tie, seed=1)
        eng.set_graph_policy(True)

    def run_steps(self, count):
        done, rounds = 0, 0
        while done < count:
            m = min(self.cycle, count - done)
            rounds += self.eng.restart_iterate(m)["sweep_rounds"]      # reset + initial sweeps + m iterations
            done += m
        return rounds

    def prime(self, steps, warmup):
        # every batch shape the warm-up and the timed region enqueue gets its hipGraph NOW, whatever --warmup is
        # (repeated until a pass neither captures nor sends plain launches: the engine adapts the rounds it enqueues for a
        #  start over its first restarts and re-captures the batch shapes when it does)
        for _ in range(6):
            c0 = self.eng.graph_counters()
            for m in sorted({self.cycle, steps % self.cycle, warmup % self.cycle} - {0}):
                self.eng.restart_iterate(m)
            self.run_steps(warmup)
            c1 = self.eng.graph_counters()
            if (c1["plain"], c1["captured"]) == (c0["plain"], c0["captured"]):
                break

    def block(self, steps):
        """one timed block: restart_iterate() synchronises the stream before returning"""
        t0 = time.perf_counter()
        rounds = self.run_steps(steps)
        return time.perf_counter() - t0, rounds

    def timed(self, steps, warmup, repeats, sync=None):
        """sync(seconds) -> seconds: barrier + maximum over ranks around a block (multi-rank callers)"""
        eng = self.eng
        self.prime(steps, warmup)
        c0 = eng.graph_counters()
        blocks, rounds = [], 0
        for _ in range(max(1, repeats)):
            if sync is not None:
                sync(None)
            dt, r = self.block(steps)
            blocks.append(sync(dt) if sync is not None else dt)
            rounds += r
        c1 = eng.graph_counters()
        total = steps * len(blocks)
        restarts = len(blocks) * ((steps + self.cycle - 1) // self.cycle)
        info = dict(
            graphs_primed=(c1["captured"] == c0["captured"] and c1["plain"] == c0["plain"]),
            graph_replays_timed=c1["replayed"] - c0["replayed"],
            host_finished_sweeps_timed=c1["host_finished_sweeps"] - c0["host_finished_sweeps"],
            iters_to_converge=int(self.first["iters"]), cycle_iterations=self.cycle, run_status=int(self.first["status"]),
            # two initial sweeps per restart ride in the total; the rest are the iterations' own
            sweep_rounds_per_iteration=(rounds - 2 * restarts) / max(total, 1) if rounds else None,
        )
        return blocks, info


def host_facts():
    """what the CPU figure was taken on (it moved from box to box with nothing to explain it): model, cores
    visible to this process, frequency governor and current clock of core 0, load average"""
    out = {"cores_visible": os.cpu_count()}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                out["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    for key, path in (("governor", "/sys/devices/system/cpu/cpu0/cpufreq/scaling_governor"),
                      ("cpu0_khz", "/sys/devices/system/cpu/cpu0/cpufreq/scaling_cur_freq")):
        try:
            out[key] = open(path).read().strip()
        except OSError:
            out[key] = None
    try:
        out["affinity_cores"] = len(os.sched_getaffinity(0))
        out["loadavg_1min"] = os.getloadavg()[0]
    except (AttributeError, OSError):
        pass
    return out


def cpu_baseline(x, nei, k, prop, center, disp, beta, algo, disper, iters):
    """Time the CPU checker on THIS host, 1 core: the compiled reference (oracle/_ref) when it is there, else the
    plain-C port (oracle/nem_oracle.c).  Bounded sample: the same workload, `iters` iterations with the convergence
    test off, minus a 0-iteration run (sort index + initial sweeps)."""
    from oracle import pyoracle
    n, d = x.shape
    extra = {}
    if pyoracle.have_reference():
        ref = pyoracle.Reference()
        kw = dict(algo=algo, beta=beta, disper=disper, cvtest="none")
        t0 = ref.classify(x, nei, k, prop, center, disp, it_max=0, **kw)["seconds"]
        t1 = ref.classify(x, nei, k, prop, center, disp, it_max=iters, **kw)["seconds"]
        per_it = max(t1 - t0, 1e-9) / iters
        kind = "reference"
        extra = dict(reference_wall_s_0_iterations=t0, reference_wall_s_n_iterations=t1, reference_iterations=iters)
    else:
        orc = pyoracle.Oracle()
        per_it = orc.run(x, nei, k, prop, center, disp, algo=algo, beta=beta, disper=disper, cvtest="none",
                         it_max=iters, tie="hash")["loop_seconds"] / iters
        kind = "port"
    out = dict(value=n * d / per_it, unit="cells/s", cores=1, kind=kind,
               sample="%d EM iterations of the same %dx%d workload, convergence test off, loop time only "
                      "(%.3f s/iteration = the difference of a %d-iteration and a 0-iteration call of the reference's "
                      "ClassifyByNem, divided by %d); host has %d cores" % (iters, n, d, per_it, iters, iters, os.cpu_count() or 0),
               em_iterations_per_sec=1.0 / per_it, seconds_per_iteration=per_it, host=host_facts())
    out.update(extra)
    return out


def pmc_traffic(kernel, n_loc, d):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes
    (profiles/r0N_pmc_density.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this very command,
    FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, checked with a known-size read).  Only valid
    for the workload it was measured on; null otherwise."""
    for name in ("r04_pmc_kernels.json", "r03_pmc_density.json", "r02_pmc_density.json", "r01_pmc_density.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            for w in rec.get("workloads", []):
                if w.get("families") != n_loc or w.get("organisms") != d or w.get("problems_per_launch"):
                    continue
                # (template instances carry their arguments in the name: the instance with the most launches is the loop's)
                hits = [(v.get("launches", 0), v["traffic_bytes_per_launch"]) for name, v in w.get("kernels", {}).items()
                        if name == kernel or name.startswith(kernel + "<")]
                if hits:
                    return max(hits)[1]
        except (OSError, ValueError):
            pass
    return None


def rocprof_average_ms(kernel, n_loc, d):
    """the committed `rocprofv3 --kernel-trace --stats` average of this kernel on this workload (profiles/
    r03_kernel_averages.json, written by profiles/summarize.py from the kernel-stats CSVs), or null"""
    try:
        name = "r04_kernel_averages.json" if os.path.isfile(os.path.join(ROOT, "profiles", "r04_kernel_averages.json")) else "r03_kernel_averages.json"
        with open(os.path.join(ROOT, "profiles", name)) as f:
            rec = json.load(f)
        for w in rec.get("workloads", []):
            if w.get("families") == n_loc and w.get("organisms") == d:
                for name, v in w.get("kernels", {}).items():
                    if name == kernel or name.startswith(kernel + "<"):
                        return v["avg_ms"]
    except (OSError, ValueError):
        pass
    return None


def roofline_block(kernels, n_loc, d, profiled_workload=True):
    """kernels: engine.profile_kernels() -- {E1, one relaxation round, M-step counts}, each timed live with HIP events
    around a string of back-to-back launches on the engine's stream (no event pair per launch).  The block's own fields
    are the dominant kernel's (E1, the longest launch of the iteration); `kernels` lists all three."""
    rows = []
    for kr in kernels:
        ach = kr["algorithmic_bytes_per_launch"] / (kr["avg_launch_ms"] * 1e-3) / 1e9 if kr["avg_launch_ms"] > 0 else 0.0
        rows.append(dict(kernel=kr["kernel"], what=kr["what"], algorithmic_bytes_per_launch=kr["algorithmic_bytes_per_launch"],
                         avg_launch_ms=kr["avg_launch_ms"], launches_timed=kr["launches_timed"], achieved=ach, unit="GB/s",
                         frac=ach / HBM_PEAK_GBS,
                         # (the committed traces and counters are of the default model on the default data)
                         rocprof_avg_launch_ms=rocprof_average_ms(kr["kernel"], n_loc, d) if profiled_workload else None,
                         traffic=pmc_traffic(kr["kernel"], n_loc, d) if profiled_workload else None))
    e1 = rows[0]
    return {
        "bound": "hbm",
        "kernel": e1["kernel"] + " (" + e1["what"] + ")",
        "achieved": e1["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": e1["frac"],
        "traffic": e1["traffic"],
        "algorithmic_bytes_per_launch": e1["algorithmic_bytes_per_launch"],
        "avg_launch_ms": e1["avg_launch_ms"], "launches_timed": e1["launches_timed"],
        "kernels": rows,
        "note": "HBM is the nominal bound (SURVEY.md 8d: bit scans, no dense contraction, no MFMA). E1 is a dependent "
                "chain per (family, class) that the reference's float rounding forbids re-associating; a launch at "
                "configs[1] moves 2 MB (0.26 us at peak) and is latency-bound, as are the sweep rounds and the counts "
                "(DESIGN.md section 4). Durations: HIP events around back-to-back launches; rocprof_avg_launch_ms: the "
                "committed rocprofv3 kernel-trace average of the same kernel on the same workload",
    }


def kernel_probe(eng, reps=50):
    try:
        return eng.profile_kernels(reps)
    except Exception:                                      # fuzzy / sharded engines: E1 alone
        p = eng.profile_density(reps)
        return [dict(kernel=p["kernel"], what="E1: Bernoulli log-density chains", avg_launch_ms=p["density_ms_avg"],
                     algorithmic_bytes_per_launch=p["algorithmic_bytes_per_launch"], launches_timed=p["density_launches"])]


def graph_facts(nei, beta):
    """what the sweep's exp(beta * context) table can serve on this graph: a site whose weight sum is below 64 can only
    have contexts the 64-entry table holds; the others take it when their three contexts happen to be small integers"""
    import numpy as np
    ptr, idx, w = nei
    rows = np.add.reduceat(np.concatenate([w, [0.0]]), ptr[:-1])[: len(ptr) - 1] if len(w) else np.zeros(len(ptr) - 1)
    rows = np.where(np.diff(ptr) > 0, rows, 0.0)
    return dict(max_weight=float(w.max()) if len(w) else 0.0, max_site_weight_sum=float(rows.max()) if len(rows) else 0.0,
                max_beta_times_weight_sum=float(beta * rows.max()) if len(rows) else 0.0,
                sites_with_weight_sum_below_64=float((rows < 64).mean()) if len(rows) else 1.0,
                sites_with_beta_weight_sum_above_88=float((beta * rows > 88.72).mean()) if len(rows) else 0.0)


def also_single_gpu(args, k, beta):
    """The regimes of this path besides the headline's, measured in the same run: the edge weights the
    reference's caller writes, many problems per launch, the shape where the kernels approach a bandwidth regime, whole
    chunks host arrays -> host arrays.  Each entry is its own try: a failure is reported, the line still prints."""
    import numpy as np
    from pangenomenem_amd import synth
    also = {}
    reps = min(args.repeats, 7)
    # (1) configs[1] with the edge weights the reference's caller writes: counts of organisms, up to D.  Two forms:
    # "adjacency" (a family's weights add up to at most twice its organisms, as in a pangenome graph: beta * sum(w) stays
    # below 709, the rows stay finite, the run converges, M = -inf) and "coverage" (uniform in 1..D on every edge: the
    # sites with three or four heavy edges pass 709, their rows are NaN, ComputeMAP's ties are redrawn in every sweep and
    # the run never converges -- in the reference as here, tests/golden/cov500_w1500_ncem).
    for key, wk, what in (("adjacency_weights", "adjacency", "weights = 60-100 % (path) / 1-15 % (chords) of the smaller organism count of the two families"),
                          ("coverage_weights", "coverage", "weights uniform in 1..500 on every edge")):
        try:
            x, nei, prop, center, disp, disper, _ = make_workload(20000, 500, k, "ushape", 2, weights=wk)
            run = EngineRun(x, nei, k, prop, center, disp, "ncem", beta, "sk_")
            steps = max(run.cycle, (140 // run.cycle) * run.cycle)
            blocks, info = run.timed(steps, run.cycle, reps)
            tf, med = timing_fields(blocks, steps)
            kern = kernel_probe(run.eng)
            sweep = [r for r in kern if r["kernel"] == "k_sweep"]
            also[key] = dict(
                workload="BASELINE configs[1] shape, " + what + " (ppanggolin.py:866-878)",
                ms_per_step=tf["ms_per_step"], ms_per_step_min=tf["ms_per_step_min"], ms_per_step_max=tf["ms_per_step_max"],
                steps=steps, repeats=tf["repeats"], cells_per_sec=20000 * 500 * steps / med, iters_to_converge=info["iters_to_converge"],
                converged=bool(run.first["converged"]), zero_density_sites=int(run.first["n_zero_density"]),
                sweep_rounds_per_iteration=info["sweep_rounds_per_iteration"], host_finished_sweeps_timed=info["host_finished_sweeps_timed"],
                sweep_round_avg_ms=sweep[0]["avg_launch_ms"] if sweep else None, e1_avg_ms=kern[0]["avg_launch_ms"],
                final_criteria_M_is_minus_inf=bool(np.isneginf(run.first["crit"][3])), graph=graph_facts(nei, beta))
            run.eng.close()
        except Exception as exc:
            also[key] = {"error": repr(exc)}
    # (1b) the headline workload under the reference's own tie rule (TIE_LIBC, the drop-in's default: ComputeMAP's ties are
    # broken by the draws of glibc's random() in the sequential sweep's order).  The headline itself runs the stateless hash
    # rule; here the sweeps carry the draw bookkeeping and a restart's two initial sweeps are completed from the host.
    try:
        x, nei, prop, center, disp, disper, _ = make_workload(20000, 500, k, "ushape", 2)
        run = EngineRun(x, nei, k, prop, center, disp, "ncem", beta, "sk_", tie="libc")
        steps = max(run.cycle, (140 // run.cycle) * run.cycle)
        blocks, info = run.timed(steps, run.cycle, reps)
        tf, med = timing_fields(blocks, steps)
        also["reference_tie_stream"] = dict(
            workload="BASELINE configs[1], the headline's matrix and graph, tie rule TIE_LIBC",
            ms_per_step=tf["ms_per_step"], ms_per_step_min=tf["ms_per_step_min"], ms_per_step_max=tf["ms_per_step_max"],
            steps=steps, repeats=tf["repeats"], cells_per_sec=20000 * 500 * steps / med, iters_to_converge=info["iters_to_converge"],
            tie_draws_first_run=int(run.first["tie_draws"]), sweep_rounds_per_iteration=info["sweep_rounds_per_iteration"],
            host_finished_sweeps_timed=info["host_finished_sweeps_timed"])
        run.eng.close()
    except Exception as exc:
        also["reference_tie_stream"] = {"error": repr(exc)}
    # (2) 64 configs[1]-sized problems in lock step (one launch per EM step for all of them)
    try:
        from pangenomenem_amd.engine import NemEngine, Result, profile_density_many
        import ctypes as C
        B = 64
        x0, _ = synth.ushaped_pa_matrix(20000, 500, 100)
        nei0 = synth.contiguity_graph(20000, 100)
        prop, center, disp = synth.default_init(500)
        rng = np.random.default_rng(0)
        engs = []
        for p in range(B):                                     # (another sample of organisms per problem: a column shuffle)
            e = NemEngine(20000, 500, 3)
            e.set_matrix(np.ascontiguousarray(x0[:, rng.permutation(500)])); e.set_graph(nei0); e.set_params(prop, center, disp)
            e.configure(algo="ncem", beta=beta, disper="sk_", tie="hash", seed=1)
            engs.append(e)
        lib = engs[0].lib
        handles = (C.c_void_p * B)(*[e._h for e in engs])
        res = (Result * B)()
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            rc = lib.nemgpu_run_many(handles, B, res)
            dt = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError("nemgpu_run_many: status %d" % rc)
            best = dt if best is None else min(best, dt)
        iters = sum(r.iters for r in res)
        e1_ms, e1_bytes = profile_density_many(engs, 30)
        nnz = int(nei0[0][-1])
        also["lockstep_64_configs1"] = dict(
            problems=B, em_iterations=iters, batch_seconds=best, us_per_problem_iteration=best * 1e6 / max(iters, 1),
            whole_problems_per_sec=B / best, cells_per_sec=iters * 20000 * 500 / best,
            whole_iteration_algorithmic_GBps=whole_iteration_bytes(20000, 500, 3, nnz) * iters / best / 1e9,
            e1_launch_ms=e1_ms, e1_algorithmic_bytes_per_launch=e1_bytes, e1_achieved_GBps=e1_bytes / (e1_ms * 1e-3) / 1e9,
            e1_frac=e1_bytes / (e1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            note="EM only (start + iterations to convergence + final criteria), inputs resident, best of 5 calls of "
                 "nemgpu_run_many; E1: the 64 members' density kernels in one launch, 30 launches between one pair of HIP events")
        for e in engs:
            e.close()
    except Exception as exc:
        also["lockstep_64_configs1"] = {"error": repr(exc)}
    # (2b) RandNemAlgo's 50 random starts (init_mode = 1) under the reference's own tie stream: their initial sweeps in
    # stream order, their iterations in lock step, against the same starts one after the other
    try:
        from pangenomenem_amd.engine import NemEngine, Result
        import ctypes as C
        x0, _ = synth.ushaped_pa_matrix(20000, 500, 9)
        nei0 = synth.contiguity_graph(20000, 9)
        rec = {}
        for mode, key in (("0", "one_after_the_other_ms"), ("1", "lockstep_ms")):
            old_env = os.environ.get("NEM_MI355X_BATCH_STARTS")
            os.environ["NEM_MI355X_BATCH_STARTS"] = mode
            try:
                e = NemEngine(20000, 500, 3)
                e.set_matrix(x0); e.set_graph(nei0)
                e.configure(algo="ncem", beta=beta, disper="sk_", propor="pk", it_max=100, tie="libc", seed=3)
                r, bst = Result(), C.c_int(-1)
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    rc = e.lib.nemgpu_run_random(e._h, 50, C.c_uint32(3), C.byref(r), C.byref(bst))
                    dt = time.perf_counter() - t0
                    if rc != 0:
                        raise RuntimeError("nemgpu_run_random: status %d" % rc)
                    best = dt if best is None else min(best, dt)
                rec[key] = best * 1e3
                rec["best_start_" + mode] = int(bst.value)
                rec["tie_draws"] = int(r.tie_draws)
                if mode == "1":
                    rec["how"] = e.random_start_counters()
                e.close()
            finally:
                if old_env is None:
                    os.environ.pop("NEM_MI355X_BATCH_STARTS", None)
                else:
                    os.environ["NEM_MI355X_BATCH_STARTS"] = old_env
        rec["same_best_start"] = rec.pop("best_start_0") == rec.pop("best_start_1")
        rec["note"] = ("50 starts of 20 000 x 500, K = 3, NCEM, srandom(3); TIE_LIBC: the starts' centre draws and every sweep's tie "
                       "draws are one random() stream; best of 3 calls each; how = lock-step rounds / starts in them / alone / redone")
        also["random_starts_50_tie_stream"] = rec
    except Exception as exc:
        also["random_starts_50_tie_stream"] = {"error": repr(exc)}
    # (3) 200 000 x 5 000 on this one GPU (BASELINE configs[3]'s matrix)
    try:
        x, nei, prop, center, disp, disper, _ = make_workload(200000, 5000, k, "ushape", 2)
        run = EngineRun(x, nei, k, prop, center, disp, "ncem", beta, "sk_")
        del x
        steps = max(run.cycle, (40 // run.cycle) * run.cycle)
        blocks, info = run.timed(steps, run.cycle, min(reps, 5))
        tf, med = timing_fields(blocks, steps)
        kern = kernel_probe(run.eng, 20)
        rows = roofline_block(kern, 200000, 5000, profiled_workload=False)["kernels"]
        also["single_gpu_200000x5000"] = dict(
            ms_per_step=tf["ms_per_step"], ms_per_step_min=tf["ms_per_step_min"], ms_per_step_max=tf["ms_per_step_max"], steps=steps,
            repeats=tf["repeats"], cells_per_sec=200000.0 * 5000 * steps / med,
            whole_iteration_algorithmic_GBps=whole_iteration_bytes(200000, 5000, 3, int(nei[0][-1])) * steps / med / 1e9,
            kernels=[{kk: r[kk] for kk in ("kernel", "avg_launch_ms", "algorithmic_bytes_per_launch", "achieved", "frac")} for r in rows],
            e1_frac=rows[0]["frac"], counts_frac=rows[2]["frac"] if len(rows) > 2 else None)
        run.eng.close()
    except Exception as exc:
        also["single_gpu_200000x5000"] = {"error": repr(exc)}
    # (4) 256 whole chunks, host arrays in, host arrays out (uploads included)
    try:
        from pangenomenem_amd.batch import solve_many
        P = 256
        x0, _ = synth.ushaped_pa_matrix(20000, 500, 2)
        nei0 = synth.contiguity_graph(20000, 2)
        prop, center, disp = synth.default_init(500)
        rng = np.random.default_rng(1)
        probs = []
        for p in range(P):
            bits = np.packbits(np.ascontiguousarray(x0[:, rng.permutation(500)]), axis=1, bitorder="little")   # 63 bytes per row -> 64
            rows = np.zeros((bits.shape[0], (bits.shape[1] + 3) // 4 * 4), np.uint8)
            rows[:, :bits.shape[1]] = bits
            probs.append((rows.view(np.uint32), nei0, 3, prop, center, disp))
        solve_many(probs[:64], 8, algo="ncem", beta=beta, disper="sk_")                         # (fills the library's pools)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = solve_many(probs, 8, algo="ncem", beta=beta, disper="sk_")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if not all(r["status"] == 0 for r in res):
            raise RuntimeError("a chunk did not solve")
        also["chunks_256"] = dict(problems=P, workers=8, seconds=best, whole_problems_per_sec=P / best,
                                  em_iterations=int(sum(r["iters"] for r in res)), cells_per_sec=sum(r["iters"] for r in res) * 1e7 / best,
                                  note="nemgpu_solve_many: bit rows + graph + parameters uploaded, solved in lock-step groups, labels "
                                       "and parameters fetched; best of 3 jobs")
    except Exception as exc:
        also["chunks_256"] = {"error": repr(exc)}
    # (5) the same kind of job the way partition()'s voting loop poses it (ppanggolin.py:1045-1086): 256 samples of 500
    # organisms out of ONE 20 000 x 5 000 pangenome that lives on the device -- every sample's problem (kept families,
    # columns, coverage-weighted graph) is formed there, nothing but initial parameters and results crosses PCIe
    try:
        from pangenomenem_amd.chunks import Master
        P = 256
        x, (ptr, idx), eb = synth.master_pangenome(20000, 5000, 2)
        rng = np.random.default_rng(0)
        subs = [rng.permutation(5000)[:500] for _ in range(P)]
        m = Master(x, ptr, idx, eb)
        del x
        kw = dict(workers=8, group=64, want_params=False, algo="ncem", beta=beta, disper="sk_", tie="hash", seed=1)
        m.solve_chunks(subs[:64], **kw)
        best, res = None, None
        for _ in range(3):
            res = m.solve_chunks(subs, **kw)
            best = m.last_call_seconds if best is None else min(best, m.last_call_seconds)
        short = None
        for _ in range(3):
            m.solve_chunks(subs, **dict(kw, it_max=1))
            short = m.last_call_seconds if short is None else min(short, m.last_call_seconds)
        m.close()
        iters = int(sum(r["iters"] for r in res))
        also["chunks_256_from_resident_master"] = dict(
            problems=P, master=[20000, 5000], sample_organisms=500, workers=8, group=64, seconds=best, whole_problems_per_sec=P / best,
            em_iterations=iters, us_per_problem_iteration=best * 1e6 / max(iters, 1),
            mean_families=float(np.mean([r["n"] for r in res])), mean_directed_edges=float(np.mean([r["nnz"] for r in res])),
            one_iteration_whole_problems_per_sec=P / short,
            note="nemgpu_solve_chunks (library call alone, best of 3): coverage weights as the reference computes them (a chunk needs "
                 "~16 EM iterations on them, against 7 on chunks_256's weights 1..8); one_iteration: the same job cut off after one "
                 "EM iteration = the pipeline around the EM (formation, start, results)")
    except Exception as exc:
        also["chunks_256_from_resident_master"] = {"error": repr(exc)}
    return also


def north_star_target(args):
    """BASELINE.json north_star: >= 50x the in-repo C NEM on 50 000 x 1 000 at K=3, beta=0.5 -- GPU and the
    compiled reference (1 host core), same workload, same run."""
    n, d, k, beta = 50000, 1000, 3, 0.5
    x, nei, prop, center, disp, disper, what = make_workload(n, d, k, args.spectrum, 3)
    run = EngineRun(x, nei, k, prop, center, disp, "ncem", beta, disper)
    steps = max(run.cycle, (args.ns_steps // run.cycle) * run.cycle)
    blocks, info = run.timed(steps, run.cycle, min(args.repeats, 9))
    tf, med = timing_fields(blocks, steps)
    roof = roofline_block(kernel_probe(run.eng), n, d)
    ms_it = tf["ms_per_step"]
    out = {
        "workload": "50000 families x 1000 organisms, K=3, beta=0.5, contiguity graph, ncem/sk_/pk; " + what,
        "gpu_ms_per_iteration": ms_it, "gpu_em_iterations_per_sec": 1e3 / ms_it,
        "gpu_cells_per_sec": n * d * 1e3 / ms_it, "gpu_steps_per_block": steps, "gpu_blocks": tf["repeats"],
        "gpu_ms_per_iteration_min": tf["ms_per_step_min"], "gpu_ms_per_iteration_max": tf["ms_per_step_max"],
        "e1_roofline_frac": roof["frac"], "e1_avg_launch_ms": roof["avg_launch_ms"],
        "kernels": [{kk: r[kk] for kk in ("kernel", "avg_launch_ms", "frac")} for r in roof["kernels"]],
        "whole_iteration_algorithmic_GBps": whole_iteration_bytes(n, d, k, int(nei[0][-1])) / (ms_it * 1e-3) / 1e9,
        "target_speedup": 50.0,
    }
    out.update(info)
    run.eng.close()
    try:
        cpu = cpu_baseline(x, nei, k, prop, center, disp, beta, "ncem", disper, args.ns_cpu_iters)
        out["reference_cpu_seconds_per_iteration"] = cpu["seconds_per_iteration"]
        out["reference_cpu"] = {kk: cpu[kk] for kk in ("kind", "cores", "sample", "reference_wall_s_0_iterations",
                                                       "reference_wall_s_n_iterations", "reference_iterations") if kk in cpu}
        out["speedup_vs_reference_cpu"] = cpu["seconds_per_iteration"] * 1e3 / ms_it
        out["meets_target"] = bool(out["speedup_vs_reference_cpu"] >= 50.0)
    except Exception as exc:
        out["reference_cpu"] = {"kind": "unavailable", "sample": "failed: %r" % (exc,)}
    return out


# ----------------------------------------------------------------------------------------------------------------
# multi-rank pieces
# ----------------------------------------------------------------------------------------------------------------
class Watchdog:
    """First-run safety of the multi-rank modes: every phase of an N > 1 run has a deadline.  A phase
    that passes it -- a collective that never completes, a wedged replay -- ends the job from a helper thread: rank 0
    prints the line it has (what was measured before the phase, or a line that says what hung) and every rank leaves
    with os._exit, so that the launcher returns and the driver has a record.  The main thread is not needed for any of
    it: it may be stuck inside a C call."""

    def __init__(self, json_fd, rank, enabled=True):
        import threading
        self.json_fd, self.rank = json_fd, rank
        self.deadline, self.name = None, ""
        self.fallback = None
        self.lock = threading.Lock()
        if enabled:
            t = threading.Thread(target=self._watch, daemon=True)
            t.start()

    def phase(self, name, seconds):
        with self.lock:
            self.name, self.deadline = name, (time.monotonic() + seconds if seconds else None)

    def done(self):
        self.phase("", None)

    def set_fallback(self, line):
        with self.lock:
            self.fallback = dict(line) if line is not None else None

    def _watch(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                name, line = self.name, self.fallback
            if not late:
                continue
            if self.rank == 0:
                if line is None:
                    line = {"metric": "em_family_x_organism_cells_per_sec", "value": None, "unit": "cells/s", "higher_is_better": True,
                            "data": "synthetic", "vs_baseline": None}
                line["error"] = "phase %r did not finish within its deadline; the job was ended by its watchdog" % name
                try:
                    os.write(self.json_fd, (json.dumps(line) + "\n").encode())
                except OSError:
                    pass
            os._exit(0)


class Ranks:
    """barrier + device synchronisation + maximum over ranks, the bracket of every timed block"""

    def __init__(self, args, rank, world, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.on_gpu = args.backend == "nccl"

    def sync(self, seconds):
        """seconds None: the bracket ahead of a block; else: the bracket behind it, returns the maximum over ranks"""
        self.torch.cuda.synchronize()
        if seconds is None:
            self.dist.barrier()
            self.torch.cuda.synchronize()
            return None
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda" if self.on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device="cuda" if self.on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())


def sharded_run(args, ranks, n_tot, d, k, beta, seed, steps, warmup, repeats, want_solo):
    """ONE n_tot x d problem sharded over the ranks: timed blocks (max over ranks each), the job's facts, and -- rank 0,
    want_solo -- the same problem on one GPU in the same run."""
    from pangenomenem_amd import distributed as nd
    rank, world, device = ranks.rank, ranks.world, ranks.device
    job = nd.ShardedNem.synthetic(n_tot, d, k, beta, rank, world, device, algo=args.algo, spectrum=args.spectrum, seed=seed)
    cycle = job.iters_to_converge()
    steps = max(steps, 1)
    # every batch shape of the timed region goes through the driver once or twice before the clock starts
    # (its graph, where graphs are on, is captured the second time a shape is seen)
    for m in sorted({cycle, steps % cycle, warmup % cycle} - {0}):
        job.run_steps(2 * m, m)
    job.run_steps(warmup, cycle)
    blocks = []
    nb0, nh0 = job.n_batches, job.n_host_sweeps
    for _ in range(max(1, repeats)):
        ranks.sync(None)
        t0 = time.perf_counter()
        job.run_steps(steps, cycle)
        ranks.torch.cuda.synchronize()
        blocks.append(ranks.sync(time.perf_counter() - t0))
    nb1, nh1 = job.n_batches, job.n_host_sweeps
    coll = job.time_collective(100)
    stride_bytes = int(job.stride)
    facts = dict(iters_to_converge=cycle, cycle_iterations=cycle, native_rccl=bool(job.native), rccl_ranks=job.eng.rccl_ranks(),
                 batches_timed=nb1 - nb0, host_finished_sweeps_timed=nh1 - nh0, pipe_depth=job.PIPE_DEPTH,
                 backend=args.backend, batch_graphs=bool(job.use_graphs or job.library_graphs),
                 collective=dict(what="in-place all-gather of the ranks' label blocks (labels, flag bytes, int32 M-step statistics)",
                                 bytes_per_rank=stride_bytes, per_allgather_ms=coll * 1e3, per_iteration=2,
                                 per_iteration_ms=2 * coll * 1e3, how="100 back-to-back all-gathers between one pair of HIP events"
                                 if job.native else "100 all-gathers through torch.distributed, wall clock"))
    facts["library_graph_counters"] = job.eng.graph_counters()
    kernels = kernel_probe(job.eng)
    n_loc = job.hi - job.lo
    solo = None
    if want_solo and rank == 0:
        try:
            x1, nei1, p1, c1, d1, _, _ = make_workload(n_tot, d, k, args.spectrum, seed)
            one = EngineRun(x1, nei1, k, p1, c1, d1, args.algo, beta, "sk_", device=device)
            st = max(one.cycle, (min(steps, 220) // one.cycle) * one.cycle)
            b1, _ = one.timed(st, one.cycle, min(repeats, 9))
            tf1, med1 = timing_fields(b1, st)
            solo = dict(ms_per_step=tf1["ms_per_step"], value=n_tot * d * st / med1, em_iterations_per_sec=st / med1, steps=st,
                        repeats=tf1["repeats"], ms_per_step_min=tf1["ms_per_step_min"], ms_per_step_max=tf1["ms_per_step_max"])
            one.eng.close()
            del x1
        except Exception as exc:
            solo = {"error": repr(exc)}
    if want_solo and world > 1:
        ranks.dist.barrier()
    return blocks, facts, kernels, n_loc, solo


def replicas_run(args, ranks, n, d, k, beta, steps, warmup, repeats):
    """N independent problems, one per GPU, no collective on the data path (the barrier brackets the timed blocks only):
    every rank solves its own n x d problem -- another seed, another chunk of organisms."""
    x, nei, prop, center, disp, disper, what = make_workload(n, d, k, args.spectrum, 2 + ranks.rank)
    run = EngineRun(x, nei, k, prop, center, disp, args.algo, beta, args.disper or disper, device=ranks.device)
    blocks, info = run.timed(steps, warmup, repeats, sync=ranks.sync)
    solves = ranks.sum(steps / run.cycle)                 # whole solves all ranks complete per block
    kernels = kernel_probe(run.eng)
    run.eng.close()
    return blocks, info, kernels, solves, what


def assemble_line(args, world, k, beta, n_tot, n_loc, d, nnz, blocks, extra, kernels, scaling, parallelism, what, ksweep, multi):
    """the JSON record of a finished measurement (rank 0)"""
    tf, med = timing_fields(blocks, args.steps)
    cells_per_s = n_tot * d * args.steps / med
    shape = ("BASELINE configs[4] (K sweep)" if (n_tot, d) == (20000, 500) and args.disper == "skd" else "custom model") \
        if ksweep else (SHAPES.get((n_tot, d), "custom shape") if args.disper == "sk_" else "custom model")
    out = {
        "metric": "em_family_x_organism_cells_per_sec",
        "value": cells_per_s,
        "unit": "cells/s",
        "em_iterations_per_sec": args.steps / med,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": tf["ms_per_step"],
        "repeats": tf["repeats"], "ms_per_step_min": tf["ms_per_step_min"], "ms_per_step_max": tf["ms_per_step_max"],
        "timed_region_s": tf["timed_region_s"],
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32 chains with f64 intermediates (reference arithmetic); int32 popcounts in the M-step",
        "data": "synthetic",
        "config": {
            "workload": "%s: %d families x %d organisms, K=%d, beta=0.5, contiguity graph (path + 5%% chords, "
                        "weights %s), %s/%s/pk; %s" % (shape, n_tot, d, k, "1..8" if args.weights == "small" else
                                                       ("1..%d (%s)" % (d, args.weights)), args.algo, args.disper, what),
            "families_total": n_tot, "families_per_gpu": n_loc, "organisms": d, "K": k, "beta": beta,
            "cycle_iterations": extra.get("cycle_iterations"),
            "parallelism": parallelism,
        },
        "roofline": roofline_block(kernels, n_loc, d, profiled_workload=(args.algo == "ncem" and args.disper == "sk_" and not ksweep
                                                                          and args.spectrum == "ushape" and not multi
                                                                          and args.weights == "small")),
    }
    out["roofline"]["whole_iteration_algorithmic_GBps"] = whole_iteration_bytes(n_tot, d, k, nnz) * args.steps / med / 1e9
    out.update(extra)
    return out, cells_per_s


def main():
    args = parse_args()
    if args.watchdog_selftest:                    # (tests/test_bench_guard.py: a phase that never ends)
        json_fd = os.dup(1)
        wd = Watchdog(json_fd, 0)
        wd.set_fallback({"metric": "em_family_x_organism_cells_per_sec", "value": 1.0, "fallback_reason": "selftest"})
        wd.phase("selftest", 1.0)
        time.sleep(60)
        raise SystemExit(9)
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
    multi = world > 1 or args.dist
    if args.disper is None:
        args.disper = "skd" if ksweep else "sk_"
    if multi and ksweep:
        raise SystemExit("the multi-GPU modes benchmark K = 3")
    extra, also = {}, None
    x = nei = prop = center = disp = run = None
    wd = Watchdog(json_fd, rank, enabled=multi and world > 1)

    if not multi:
        nem_build.build()                          # no-op when the in-tree library is up to date
        n_tot, d = args.families or 20000, args.organisms or 500
        # (50 000 x 1 000 is BASELINE configs[2]'s problem: the same matrix -- seed 3 -- as the sharded runs and the
        #  north-star block time)
        x, nei, prop, center, disp, _, what = make_workload(n_tot, d, k, args.spectrum, 3 if (n_tot, d) == (50000, 1000) else 2, ksweep,
                                                            weights=args.weights)
        run = EngineRun(x, nei, k, prop, center, disp, args.algo, beta, args.disper)
        blocks, extra = run.timed(args.steps, args.warmup, args.repeats)
        # kernel durations: HIP events on the engine's stream around strings of launches, on the state the timed region
        # just left (the timed region itself replays captured graphs, which carry no events)
        kernels = kernel_probe(run.eng)
        n_loc, nnz = n_tot, int(nei[0][-1])
        scaling, parallelism = "weak", "1 GPU"
    else:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        device = local_rank % max(ndev, 1) if args.backend == "gloo" else local_rank
        torch.cuda.set_device(device)
        if "RANK" not in os.environ:              # plain `python bench.py --dist`: a 1-rank group
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        wd.phase("process group", args.phase_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
        if rank == 0:                             # one rank checks / rebuilds the library, the others wait for it
            nem_build.build()
        dist.barrier()
        ranks = Ranks(args, rank, world, device)
        what = make_workload(64, 32, 3, args.spectrum, 1)[6]
        replicas_line = None

        def replicas_measure(steps, warmup, repeats):
            n_r, d_r = args.families or 20000, args.organisms or 500
            if args.scaling != "replicas":
                n_r, d_r = 20000, 500
            b, info, kern, solves, w = replicas_run(args, ranks, n_r, d_r, k, beta, steps, warmup, repeats)
            info = dict(info)
            info.update(whole_solves_per_block=solves, rccl_ranks=0, backend=args.backend,
                        collective=dict(what="none on the data path (independent problems); a barrier brackets each timed block",
                                        per_iteration=0, per_iteration_ms=0.0))
            par = ("%d independent %d x %d problems, one per GPU, no collective on the data path (the reference's own "
                   "parallel form: one NEM problem per organism chunk, ppanggolin.py:1039-1095)" % (world, n_r, d_r))
            return n_r, d_r, b, info, kern, solves, w, par

        if args.scaling == "replicas":
            wd.phase("replicas", args.phase_timeout)
            n_loc, d, blocks, extra, kernels, solves, what, parallelism = replicas_measure(args.steps, args.warmup, args.repeats)
            n_tot = n_loc * world
            nnz = int(2.1 * n_tot)
            scaling = "weak"
        else:
            # The mode with NO collective on the data path goes first (N > 1): it cannot hang on a collective, and its
            # record is what rank 0 prints if the sharded mode -- whose native RCCL path has never run with more than one
            # rank on the development box -- fails or passes its deadline.
            if world > 1 and not args.no_extras:
                saved = args.steps
                try:
                    wd.phase("replicas (also)", args.phase_timeout)
                    st2 = max(args.steps, 140)
                    args.steps = st2
                    n_r, d_r, b2, i2, kern2, solves, w2, par2 = replicas_measure(st2, 14, min(args.repeats, 9))
                    tf2, med2 = timing_fields(b2, st2)
                    also = {"replicas_20000x500_per_gpu": dict(
                        value=world * n_r * d_r * st2 / med2, unit="cells/s", ms_per_step=tf2["ms_per_step"],
                        whole_solves_per_sec=solves / med2, scaling="weak", collectives_per_iteration=0,
                        repeats=tf2["repeats"], ms_per_step_min=tf2["ms_per_step_min"], ms_per_step_max=tf2["ms_per_step_max"],
                        note="N independent configs[1]-sized problems, one per GPU; the N = 1 default line of this script is the one-GPU figure")}
                    if rank == 0:
                        replicas_line, _ = assemble_line(args, world, k, beta, n_r * world, n_r, d_r, int(2.1 * n_r * world), b2, i2, kern2,
                                                         "weak", par2, w2, False, True)
                        replicas_line["fallback_reason"] = ("the sharded mode (--scaling %s) did not finish: this is the replicas "
                                                            "measurement of the same run" % args.scaling)
                        wd.set_fallback(replicas_line)
                except Exception as exc:
                    also = {"replicas_20000x500_per_gpu": {"error": repr(exc)}}
                finally:
                    args.steps = saved
            if args.scaling == "strong":
                n_tot, d = args.families or 50000, args.organisms or 1000
                seed = 3 if (n_tot, d) == (50000, 1000) else 2
            else:
                d = args.organisms or 500
                n_tot, seed = (args.families or 20000) * world, 2
            wd.phase("sharded EM (%s scaling)" % args.scaling, args.phase_timeout)
            fallback_reason = None
            try:
                blocks, extra, kernels, n_loc, solo = sharded_run(args, ranks, n_tot, d, k, beta, seed, args.steps, args.warmup,
                                                                  args.repeats, want_solo=(world > 1 and args.scaling == "strong"))
            except Exception as exc:
                # every rank that got an exception here takes the same second route: all collectives through
                # torch.distributed (a rank that did NOT get one is stuck in a collective; the deadline ends the job)
                fallback_reason = "native RCCL path failed (%r): collectives through torch.distributed" % (exc,)
                os.environ["NEM_DIST_NATIVE"] = "0"
                wd.phase("sharded EM (%s scaling), torch collectives" % args.scaling, args.phase_timeout)
                blocks, extra, kernels, n_loc, solo = sharded_run(args, ranks, n_tot, d, k, beta, seed, args.steps, args.warmup,
                                                                  args.repeats, want_solo=(world > 1 and args.scaling == "strong"))
            if fallback_reason:
                extra["fallback_reason"] = fallback_reason
            nnz = int(2.1 * n_tot)
            scaling = args.scaling if world > 1 else "weak"
            if solo is not None:
                extra["single_gpu_same_workload"] = solo
                if "ms_per_step" in solo:
                    extra["speedup_vs_single_gpu_same_workload"] = solo["ms_per_step"] / (median(blocks) * 1e3 / args.steps)
            parallelism = ("families sharded over %d GPUs in contiguous blocks (%s scaling); per EM iteration two RCCL "
                           "all-gathers of the label blocks, the second also carrying the ranks' int32 M-step statistics"
                           % (world, scaling)) if world > 1 else "1 GPU through the sharded driver"

    out = None
    if rank == 0:
        out, cells_per_s = assemble_line(args, world, k, beta, n_tot, n_loc, d, nnz, blocks, extra, kernels, scaling, parallelism,
                                         what, ksweep, multi)
        if also is not None:
            out["also"] = dict(also)
        wd.set_fallback(out)                      # from here on a phase that hangs costs its own entry only

    # ---- what else N GPUs can do with this path, in the same run: the strong-scaling figure of the shape where it can pay
    if multi and world > 1 and not args.no_extras and args.scaling != "replicas":
        n3, d3 = (int(v) for v in args.extras_strong_shape.lower().split("x"))
        if (n_tot, d) != (n3, d3):
            key3 = "strong_%dx%d" % (n3, d3)
            try:
                wd.phase(key3, args.phase_timeout)
                st3 = 60
                b3, f3, _, _, solo3 = sharded_run(args, ranks, n3, d3, k, beta, 2, st3, 6, min(args.repeats, 7), want_solo=True)
                tf3, med3 = timing_fields(b3, st3)
                rec = dict(value=float(n3) * d3 * st3 / med3, unit="cells/s", ms_per_step=tf3["ms_per_step"], scaling="strong",
                           families=n3, organisms=d3,
                           repeats=tf3["repeats"], ms_per_step_min=tf3["ms_per_step_min"], ms_per_step_max=tf3["ms_per_step_max"],
                           collective=f3["collective"], rccl_ranks=f3["rccl_ranks"], single_gpu_same_workload=solo3)
                if solo3 and "ms_per_step" in solo3:
                    rec["speedup_vs_single_gpu_same_workload"] = solo3["ms_per_step"] / tf3["ms_per_step"]
            except Exception as exc:
                rec = {"error": repr(exc)}
            if rank == 0:
                out.setdefault("also", {})[key3] = rec
    wd.done()

    if rank == 0:
        if not multi and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(x, nei, k, prop, center, disp, beta, args.algo, args.disper,
                                                   args.cpu_iters)
                out["speedup_vs_cpu_baseline"] = cells_per_s / out["cpu_baseline"]["value"]
            except Exception as exc:   # the checker is optional on the box; the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 1, "kind": "unavailable",
                                       "sample": "failed: %r" % (exc,)}
        if not multi and not args.no_north_star and not ksweep:
            try:
                run.eng.close()
                out["north_star_target"] = north_star_target(args)
            except Exception as exc:
                out["north_star_target"] = {"error": repr(exc)}
        if not multi and not args.no_extras and not ksweep and args.algo == "ncem":
            try:
                run.eng.close()
            except Exception:
                pass
            t_also = time.perf_counter()
            out["also"] = also_single_gpu(args, k, beta)
            out["also"]["seconds"] = time.perf_counter() - t_also
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if multi:
        import torch.distributed as dist
        wd.phase("leaving the process group", 60)
        dist.barrier()
        dist.destroy_process_group()
        wd.done()


if __name__ == "__main__":
    main()

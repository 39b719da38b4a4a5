#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs merged back under gpurun_out/ into the small, tracked summaries in profiles/.

    gpurun --timeout 1100 -- 'bash profiles/collect_r02.sh'   # on the GPU box
    python profiles/summarize.py r02                           # here

Writes <round>_<cfg>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim), <round>_pmc_by_kernel_<cfg>.csv,
<round>_pmc_density.json (what bench.py reports as roofline.traffic), <round>_fetch_calibration.json, the bench lines
and the other measurements, and <round>_batch16_e1_roofline.json (the E1 kernel of a lock-step batch of 16 problems:
full-launch duration from the kernel trace, algorithmic bytes, HBM traffic from the counters).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
HERE = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
Q = "--no-cpu-baseline" + ("" if R == "r01" else " --no-north-star")
SHAPES = {"c2": (20000, 500, "python3 bench.py --steps 56 --warmup 7 " + Q),
          "c4": (200000, 5000, "python3 bench.py --families 200000 --organisms 5000 --steps 10 --warmup 2 " + Q),
          "b16": (20000, 500, "python3 profiles/batch_lockstep.py 20000 500 16   (16 problems per launch)")}
HBM_PEAK = 8000.0


def newest(pattern):
    files = glob.glob(os.path.join(OUT, pattern))
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    return name.split("(")[0].replace("void ", "").replace("nemk::", "")


def per_kernel(counter_csv, stat="mean"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(counter_csv)):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    if stat == "max":
        return {k: (len(v), max(v)) for k, v in agg.items()}
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def full_launch_us(trace_csv, kernel):
    """duration of the kernel's FULL launches in a lock-step batch: the median of the launches within 10 % of the
    longest one (members converge at different iterations, so later launches of a batch serve fewer and fewer of its
    members; launches that return at the stop word are a few microseconds)"""
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace_csv))
               if short(r["Kernel_Name"]) == kernel)
    if not d:
        return None, 0, 0
    top = [x for x in d if x >= 0.9 * d[-1]]
    return top[len(top) // 2], len(d), len(top)


def main():
    averages = []
    for cfg in ("c2", "c3", "c4", "c2_fuzzy", "b16", "dist1", "c2_adjacency"):
        f = newest("%s_%s/*/*_kernel_stats.csv" % (R, cfg))
        if f:
            shutil.copy(f, os.path.join(HERE, "%s_%s_kernel_stats.csv" % (R, cfg)))
            shape = {"c2": (20000, 500), "c3": (50000, 1000), "c4": (200000, 5000)}.get(cfg)
            if shape:
                # what bench.py quotes as rocprof_avg_launch_ms: the kernel-trace average of every launch of the run
                ks = {}
                for r in csv.DictReader(open(f)):
                    ks[short(r["Name"])] = {"avg_ms": float(r["AverageNs"]) * 1e-6, "calls": int(r["Calls"]),
                                            "min_ms": float(r["MinNs"]) * 1e-6, "max_ms": float(r["MaxNs"]) * 1e-6}
                averages.append({"families": shape[0], "organisms": shape[1], "source": "%s_%s_kernel_stats.csv" % (R, cfg),
                                 "kernels": ks})
    if averages:
        json.dump({"what": "rocprofv3 --kernel-trace --stats averages per kernel (all launches of the profiled bench run, "
                           "the early returns at the stop word included)", "workloads": averages},
                  open(os.path.join(HERE, "%s_kernel_averages.json" % R), "w"), indent=1)
    calib = newest("%s_calib/*/*_counter_collection.csv" % R)
    factor = None
    if calib:
        c = per_kernel(calib).get("k_calib_read16")
        if c:
            known_kb = (1 << 30) / 1024.0
            factor = known_kb / c[1]
            json.dump({"kernel": "k_calib_read16", "bytes_read_known": 1 << 30, "FETCH_SIZE_KB_reported": c[1],
                       "true_over_reported": factor,
                       "note": "16 bytes per lane, lanes consecutive (E1's pattern): MI355X_MICROARCH.md says FETCH_SIZE "
                               "reports half the bytes of such reads on gfx950; this is the check"},
                      open(os.path.join(HERE, "%s_fetch_calibration.json" % R), "w"), indent=1)
    scale = factor or 2.0
    workloads = []
    for cfg, (n, d, cmd) in SHAPES.items():
        fetch = newest("%s_%s_fetch/*/*_counter_collection.csv" % (R, cfg))
        write = newest("%s_%s_write/*/*_counter_collection.csv" % (R, cfg))
        if not (fetch and write):
            continue
        # a batch's launches differ in size (members converge at different iterations): its full launches are the max
        stat = "max" if cfg == "b16" else "mean"
        fk, wk = per_kernel(fetch, stat), per_kernel(write, stat)
        rec = {}
        with open(os.path.join(HERE, "%s_pmc_by_kernel_%s.csv" % (R, cfg)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KB_%s_raw" % stat, "WRITE_SIZE_KB_%s" % stat, "traffic_bytes_per_launch"])
            for k in sorted(set(fk) | set(wk)):
                fr = fk.get(k, (0, 0.0)); wr = wk.get(k, (0, 0.0))
                traffic = (fr[1] * scale + wr[1]) * 1024.0
                w.writerow([k, fr[0], "%.2f" % fr[1], "%.2f" % wr[1], "%.0f" % traffic])
                rec[k] = dict(fetch_kb_raw=fr[1], write_kb=wr[1], traffic_bytes_per_launch=traffic, launches=fr[0])
        entry = {"families": n, "organisms": d, "command": cmd, "fetch_scale": scale,
                 "kernels": {k: rec[k] for k in rec if k.startswith("k_density")}, "all_kernels": rec}
        if cfg == "b16":
            entry["problems_per_launch"] = 16
        workloads.append(entry)
    if workloads:
        # (round 4: every kernel of the run, not the density kernels only -- bench.py's roofline.kernels[].traffic for the
        #  sweep rounds and the counts comes from here)
        every = [dict({k: v for k, v in w.items() if k not in ("kernels", "all_kernels")}, kernels=w["all_kernels"]) for w in workloads]
        json.dump({"what": "HBM-side bytes per launch, every kernel of the profiled run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                           "separate passes, FETCH_SIZE scaled by the calibration factor (MI355X_MICROARCH.md, HBM / rocprofv3 section)",
                   "workloads": every}, open(os.path.join(HERE, "%s_pmc_kernels.json" % R), "w"), indent=1)
        for w in workloads:
            w.pop("all_kernels", None)
        json.dump({"workloads": [w for w in workloads if "problems_per_launch" not in w] +
                                [w for w in workloads if "problems_per_launch" in w]},
                  open(os.path.join(HERE, "%s_pmc_density.json" % R), "w"), indent=1)
    # E1 of the lock-step batch of 16: roofline block from the trace + counters
    trace = newest("%s_b16/*/*_kernel_trace.csv" % R)
    if trace:
        out = {"workload": "16 independent configs[1]-sized problems (20000 x 500, K=3) per launch, nemgpu_run_many",
               "kernels": {}}
        for kern in ("k_density_b", "k_density_fused_b"):
            us, calls, full = full_launch_us(trace, kern)
            if us is None:
                continue
            n, d, k, B = 20000, 500, 3, 16
            alg = B * (4 * ((d + 31) // 32) * n + 24 * k * d + 12 * n * k)
            tr = None
            for w in workloads:
                if w.get("problems_per_launch") == 16 and kern in w["kernels"]:
                    tr = w["kernels"][kern]["traffic_bytes_per_launch"]
            out["kernels"][kern] = {"bound": "hbm", "full_launch_us": us, "launches_in_trace": calls, "full_launches": full,
                                    "algorithmic_bytes_per_launch": alg, "achieved": alg / (us * 1e-6) / 1e9,
                                    "peak": HBM_PEAK, "unit": "GB/s", "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK,
                                    "traffic": tr}
        json.dump(out, open(os.path.join(HERE, "%s_batch16_e1_roofline.json" % R), "w"), indent=1)
    for cfg in ("c2", "c3", "c4", "c2_fuzzy"):
        b = os.path.join(OUT, "%s_%s_prof.json" % (R, cfg))
        if os.path.isfile(b) and os.path.getsize(b) > 0:
            shutil.copy(b, os.path.join(HERE, "%s_bench_%s_profiled.json" % (R, cfg)))
    # the other measurements the collection script leaves under gpurun_out/ (copied as they are; the drop-in record is
    # written by tests/test_gpu_dropin_fullsize.py whenever the GPU tests run)
    for name in ("bench_c2.json", "bench_c2_driver.json", "bench_20000x500.json", "bench_50000x1000.json",
                 "bench_200000x5000.json", "bench_20000x500_latent3.json", "bench_20000x500_fuzzy.json",
                 "bench_20000x500_skd.json", "bench_dist_world1.json", "bench_2ranks_gloo_one_gpu.json", "dist_world1.json",
                 "dist_world1_20000x500.json", "dist_2ranks_gloo_strong.json", "dist_2ranks_gloo_replicas.json", "dist_2ranks_gloo_weak.json",
                 "pcie_inclusive.json", "batch_chunks.json", "batch_chunks_256.json", "batch_lockstep.json", "random_starts.json",
                 "bench_20000x500_adjacency.json", "bench_20000x500_coverage.json", "chunks_device.json", "sweep_phases.json", "gpu_suite.txt",
                 "dropin_whole_call.json", "dropin_logged.json", "fuzzy_mstep.txt", "fuzzy_mstep_lane_per_chain.txt", "fuzzy_mstep_wave_per_chain.txt"):
        b = os.path.join(OUT, "%s_%s" % (R, name))
        if os.path.isfile(b) and os.path.getsize(b) > 0:
            shutil.copy(b, os.path.join(HERE, "%s_%s" % (R, name)))
    b = os.path.join(OUT, "dropin_whole_call.json")     # (written by tests/test_gpu_dropin_fullsize.py, no round in its name)
    if os.path.isfile(b) and os.path.getsize(b) > 0:
        shutil.copy(b, os.path.join(HERE, "%s_dropin_whole_call.json" % R))
    ks = os.path.join(OUT, "%s_ksweep.jsonl" % R)
    if os.path.isfile(ks):
        rows = []
        for line in open(ks):
            line = line.strip()
            if line.startswith("{"):
                rec = json.loads(line)
                rows.append({"K": rec["config"]["K"], "ms_per_iteration": rec["ms_per_step"],
                             "em_iterations_per_sec": rec["em_iterations_per_sec"], "cells_per_sec": rec["value"],
                             "iters_to_converge": rec.get("iters_to_converge"),
                             "e1_kernel": rec["roofline"]["kernel"], "e1_avg_launch_ms": rec["roofline"]["avg_launch_ms"],
                             "e1_roofline_frac": rec["roofline"]["frac"]})
        if rows:
            json.dump({"workload": "BASELINE configs[4]: 20000 x 500, skd, ncem, beta 0.5, 10-latent-group matrix, "
                                   "python3 bench.py --k K --steps 200 --warmup 20", "rows": rows},
                      open(os.path.join(HERE, "%s_ksweep.json" % R), "w"), indent=1)


if __name__ == "__main__":
    main()

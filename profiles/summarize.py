#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs merged back under gpurun_out/ into the small, tracked summaries in profiles/.

    gpurun --timeout 1100 -- 'bash profiles/collect.sh'      # on the GPU box
    python profiles/summarize.py                              # here

Writes r01_<cfg>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim), r01_pmc_by_kernel_<cfg>.csv,
r01_pmc_density.json (what bench.py reports as roofline.traffic), r01_fetch_calibration.json and the bench lines.
"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
HERE = os.path.join(ROOT, "profiles")
SHAPES = {"c2": (20000, 500, "python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline"),
          "c4": (200000, 5000, "python3 bench.py --families 200000 --organisms 5000 --steps 10 --warmup 2 --no-cpu-baseline")}


def newest(pattern):
    files = glob.glob(os.path.join(OUT, pattern))
    return max(files, key=os.path.getmtime) if files else None


def per_kernel(counter_csv):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(counter_csv)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nemk::", "")
        agg[name].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    for cfg in ("c2", "c3", "c4"):
        f = newest("r01_%s/*/*_kernel_stats.csv" % cfg)
        if f:
            shutil.copy(f, os.path.join(HERE, "r01_%s_kernel_stats.csv" % cfg))
    calib = newest("r01_calib/*/*_counter_collection.csv")
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
                      open(os.path.join(HERE, "r01_fetch_calibration.json"), "w"), indent=1)
    scale = factor or 2.0
    workloads = []
    for cfg, (n, d, cmd) in SHAPES.items():
        fetch = newest("r01_%s_fetch/*/*_counter_collection.csv" % cfg)
        write = newest("r01_%s_write/*/*_counter_collection.csv" % cfg)
        if not (fetch and write):
            continue
        fk, wk = per_kernel(fetch), per_kernel(write)
        rec = {}
        with open(os.path.join(HERE, "r01_pmc_by_kernel_%s.csv" % cfg), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg_raw", "WRITE_SIZE_KB_avg", "traffic_bytes_per_launch"])
            for k in sorted(set(fk) | set(wk)):
                fr = fk.get(k, (0, 0.0)); wr = wk.get(k, (0, 0.0))
                traffic = (fr[1] * scale + wr[1]) * 1024.0
                w.writerow([k, fr[0], "%.2f" % fr[1], "%.2f" % wr[1], "%.0f" % traffic])
                rec[k] = dict(fetch_kb_raw=fr[1], write_kb=wr[1], traffic_bytes_per_launch=traffic, launches=fr[0])
        workloads.append({"families": n, "organisms": d, "command": cmd, "fetch_scale": scale,
                          "kernels": {k: rec[k] for k in rec if k.startswith("k_density")}})
    if workloads:
        json.dump({"workloads": workloads}, open(os.path.join(HERE, "r01_pmc_density.json"), "w"), indent=1)
    b = os.path.join(OUT, "r01_bench_c2.json")
    if os.path.isfile(b):
        shutil.copy(b, os.path.join(HERE, "r01_bench_c2.json"))
    for cfg in ("c2", "c3", "c4"):
        b = os.path.join(OUT, "r01_%s_prof.json" % cfg)
        if os.path.isfile(b):
            shutil.copy(b, os.path.join(HERE, "r01_bench_%s_profiled.json" % cfg))
    # the other measurements collect.sh leaves under gpurun_out/ (copied as they are; the drop-in record is written
    # by tests/test_gpu_dropin_fullsize.py whenever the GPU tests run)
    for name in ("r01_bench_20000x500.json", "r01_bench_50000x1000.json", "r01_bench_200000x5000.json",
                 "r01_bench_dist_world1.json", "r01_pcie_inclusive.json", "r01_batch_chunks.json",
                 "r01_dropin_whole_call.json"):
        b = os.path.join(OUT, name)
        if os.path.isfile(b) and os.path.getsize(b) > 0:
            shutil.copy(b, os.path.join(HERE, name))
    f = newest("r01_c2_fuzzy/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(HERE, "r01_c2_fuzzy_kernel_stats.csv"))


if __name__ == "__main__":
    main()

#!/bin/bash
# Round-1 profile collection.  Runs on the GPU box from the repo root:
#     gpurun --timeout 1100 -- 'bash profiles/collect.sh'
# then, back in the container:   python profiles/summarize.py
# Kernel timing (--kernel-trace --stats) and each PMC counter are separate runs, as MI355X_MICROARCH.md's
# HBM / rocprofv3 section prescribes; the program itself follows `--` (no env/bash hop).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
# default bench line (configs[1], with the CPU baseline leg)
python3 bench.py > $O/r01_bench_c2.json 2> $O/r01_bench_c2.err
# per-kernel durations: C2 (default command), C3, C4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01_c2 -- python3 bench.py --no-cpu-baseline > $O/r01_c2_prof.json 2> $O/r01_c2_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01_c3 -- python3 bench.py --families 50000 --organisms 1000 --steps 100 --warmup 10 --no-cpu-baseline > $O/r01_c3_prof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01_c4 -- python3 bench.py --families 200000 --organisms 5000 --steps 20 --warmup 4 --no-cpu-baseline > $O/r01_c4_prof.json 2> /dev/null
# HBM traffic counters, one counter per run: C2 and C4
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r01_c2_fetch -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r01_c2_write -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r01_c4_fetch -- python3 bench.py --families 200000 --organisms 5000 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r01_c4_write -- python3 bench.py --families 200000 --organisms 5000 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
# FETCH_SIZE calibration: a known 1 GiB read, one dword per lane
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r01_calib -- python3 -c "from pangenomenem_amd.engine import calibrate_fetch; calibrate_fetch(1 << 30, 3)" > /dev/null 2>&1

# unprofiled bench lines of the same build (C3, C4), the fuzzy path's kernels, the family-sharded driver on one GPU
python3 bench.py --families 50000 --organisms 1000 --steps 1000 --warmup 100 --no-cpu-baseline > $O/r01_bench_50000x1000.json 2> /dev/null
python3 bench.py --families 200000 --organisms 5000 --steps 300 --warmup 30 --no-cpu-baseline > $O/r01_bench_200000x5000.json 2> /dev/null
python3 bench.py --no-cpu-baseline > $O/r01_bench_20000x500.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01_c2_fuzzy -- python3 bench.py --algo nem --steps 50 --warmup 5 --no-cpu-baseline > $O/r01_c2_fuzzy_prof.json 2> /dev/null
python3 bench.py --dist --no-cpu-baseline > $O/r01_bench_dist_world1.json 2> /dev/null
python3 profiles/pcie_inclusive.py > $O/r01_pcie_inclusive.json 2> /dev/null
python3 profiles/batch_chunks.py > $O/r01_batch_chunks.json 2> /dev/null
cat $O/r01_bench_c2.json

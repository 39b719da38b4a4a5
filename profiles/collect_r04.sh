#!/bin/bash
# Round-4 profile collection.  Runs on the GPU box from the repo root:
#     gpurun --timeout 1150 -- 'bash profiles/collect_r04.sh A'     (bench lines, kernel traces, counters)
#     gpurun --timeout 1150 -- 'bash profiles/collect_r04.sh B'     (bench variants, multi-GPU modes, K sweep, batches, drop-in)
# then, back in the container:   python profiles/summarize.py r04
# Kernel timing (--kernel-trace --stats) and each PMC counter are separate runs, as MI355X_MICROARCH.md's
# HBM / rocprofv3 section prescribes; the program itself follows `--` (no env/bash hop).
# (no set -e: a step that fails -- rocprofv3 itself segfaulted once on a long C2 run -- must not cost the steps behind it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
R=r04
mkdir -p $O
PART=${1:-A}
Q="--no-cpu-baseline --no-north-star"
X="--no-extras"
if [ "$PART" = "A" ]; then
# the default bench line (configs[1], CPU baseline and north-star block) and the line as the driver asks for it
python3 bench.py > $O/${R}_bench_c2.json 2> $O/${R}_bench_c2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_c2_driver.json 2> /dev/null
echo "[collect] bench lines done"
# per-kernel durations: C2 (default command), C3, C4, the fuzzy path, a lock-step batch of 16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_c2 -- python3 bench.py --steps 210 --warmup 21 --repeats 3 $Q $X > $O/${R}_c2_prof.json 2> $O/${R}_c2_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_c3 -- python3 bench.py --families 50000 --organisms 1000 --steps 220 --warmup 22 --repeats 5 $Q $X > $O/${R}_c3_prof.json 2> $O/${R}_c3_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_c4 -- python3 bench.py --families 200000 --organisms 5000 --steps 20 --warmup 4 --repeats 3 $Q $X > $O/${R}_c4_prof.json 2> $O/${R}_c4_prof.err
echo "[collect] kernel traces c2 c3 c4 done"; ls $O | grep -c ${R}_c
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_c2_fuzzy -- python3 bench.py --algo nem --steps 50 --warmup 5 --repeats 3 $Q $X > $O/${R}_c2_fuzzy_prof.json 2> $O/${R}_c2_fuzzy_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_b16 -- python3 profiles/batch_lockstep.py 20000 500 16 > $O/${R}_b16_prof.json 2> $O/${R}_b16_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_dist1 -- python3 bench.py --dist --steps 220 --warmup 22 --repeats 5 --no-cpu-baseline > $O/${R}_dist1_prof.json 2> $O/${R}_dist1_prof.err
echo "[collect] kernel traces done"
# HBM traffic counters, one counter per run: C2, C4, the lock-step batch of 16
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_c2_fetch -- python3 bench.py --steps 56 --warmup 7 --repeats 3 $Q $X > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_c2_write -- python3 bench.py --steps 56 --warmup 7 --repeats 3 $Q $X > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_c4_fetch -- python3 bench.py --families 200000 --organisms 5000 --steps 10 --warmup 2 --repeats 2 $Q $X > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_c4_write -- python3 bench.py --families 200000 --organisms 5000 --steps 10 --warmup 2 --repeats 2 $Q $X > /dev/null 2>&1
echo "[collect] counters c2 c4 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_b16_fetch -- python3 profiles/batch_lockstep.py 20000 500 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_b16_write -- python3 profiles/batch_lockstep.py 20000 500 16 > /dev/null 2>&1
# FETCH_SIZE calibration: a known 1 GiB read, 16 bytes per lane
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_calib -- python3 -c "from pangenomenem_amd.engine import calibrate_fetch; calibrate_fetch(1 << 30, 3)" > /dev/null 2>&1
echo "[collect] counters done"
fi
if [ "$PART" = "B" ]; then
# unprofiled bench lines of the same build (C3, C4), the fuzzy path, free dispersion
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --families 50000 --organisms 1000 --steps 1100 --warmup 110 $Q $X > $O/${R}_bench_50000x1000.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --families 200000 --organisms 5000 --steps 300 --warmup 30 --repeats 9 $Q $X > $O/${R}_bench_200000x5000.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --spectrum latent3 $Q --no-extras > $O/${R}_bench_20000x500_latent3.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --weights adjacency $Q --no-extras > $O/${R}_bench_20000x500_adjacency.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --weights coverage --steps 200 --warmup 100 --repeats 9 $Q --no-extras > $O/${R}_bench_20000x500_coverage.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_c2_adjacency -- python3 bench.py --weights adjacency --steps 200 --warmup 20 --repeats 3 $Q --no-extras > /dev/null 2>&1
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --algo nem --steps 300 --warmup 30 --repeats 9 $Q $X > $O/${R}_bench_20000x500_fuzzy.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --disper skd $Q $X > $O/${R}_bench_20000x500_skd.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
echo "[collect] bench variants done"
# the multi-GPU modes on the one GPU of the box: one rank over RCCL through the sharded driver (the 1-rank ratio against
# the single engine on the same problem), two ranks over gloo (host-staged collectives: a rehearsal of the protocol,
# not of its speed) in the three modes
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --dist --steps 1100 --warmup 110 --no-cpu-baseline > $O/${R}_dist_world1.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --dist --families 20000 --organisms 500 --steps 700 --warmup 70 --no-cpu-baseline > $O/${R}_dist_world1_20000x500.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --gpus 2 --backend gloo --steps 110 --warmup 11 --repeats 5 --extras-strong-shape 100000x2500 > $O/${R}_dist_2ranks_gloo_strong.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --gpus 2 --backend gloo --scaling replicas --steps 140 --warmup 14 --repeats 5 > $O/${R}_dist_2ranks_gloo_replicas.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 bench.py --gpus 2 --backend gloo --scaling weak --steps 70 --warmup 7 --repeats 5 > $O/${R}_dist_2ranks_gloo_weak.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
echo "[collect] multi-GPU modes done"
# BASELINE configs[4]: the K sweep (20 000 x 500, skd), one line per K
rm -f $O/${R}_ksweep.jsonl
for k in 2 3 4 5 6 7 8 9 10; do PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 120 python3 bench.py --k $k --steps 200 --warmup 20 --repeats 9 $Q $X >> $O/${R}_ksweep.jsonl 2>> $O/${R}_partB.err; done; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
echo "[collect] K sweep done"
# lock-step batches, random starts, whole chunks (one device, and the device list), host-buffer-inclusive solve, the fuzzy M-step alone
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/batch_lockstep.py > $O/${R}_batch_lockstep.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/random_starts.py > $O/${R}_random_starts.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/batch_chunks.py > $O/${R}_batch_chunks.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/batch_chunks.py 256 > $O/${R}_batch_chunks_256.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/pcie_inclusive.py > $O/${R}_pcie_inclusive.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/fuzzy_mstep.py > $O/${R}_fuzzy_mstep.txt 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/chunks_device.py 256 8 > $O/${R}_chunks_device.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
NEM_MI355X_FUSED_SWEEP=1 python3 profiles/sweep_phases.py > $O/${R}_sweep_phases.json 2> /dev/null
echo "[collect] batches done"
# the drop-in nem() against the reference's own nem() on the same files (writes gpurun_out/dropin_whole_call.json)
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 -m pytest tests/test_gpu_dropin_fullsize.py -q -m gpu -k whole_call > $O/${R}_dropin_test.log 2>&1 || true; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
PYTHONFAULTHANDLER=1 timeout -s ABRT -k 15 280 python3 profiles/dropin_logged.py > $O/${R}_dropin_logged.json 2>> $O/${R}_partB.err; rc=$?; echo "[collect] step done: $rc"; [ $rc -lt 124 ] || { echo "[collect] a step was killed at its limit: stopping here"; exit 1; }
echo "[collect] done"
fi

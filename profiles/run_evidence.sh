cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03ev && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --timeout 600 > $O/full_suite.log 2>&1; rc=$?; echo full rc=$rc; tail -3 $O/full_suite.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
NEM_FUZZ_LARGE_SEEDS=80 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -k large --timeout 600 > $O/r03_fuzz_large.txt 2>&1; rc=$?; echo large rc=$rc; tail -2 $O/r03_fuzz_large.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 900 python3 tests/fuzz_extended.py 60000 > $O/r03_fuzz_extended.json 2> $O/fuzz_extended.err; echo ext rc=$?; head -c 600 $O/r03_fuzz_extended.json

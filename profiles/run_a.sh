cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
timeout -k 10 900 python3 -m pytest tests/test_chain.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_dropin_fullsize.py -x -q -m gpu --timeout 400 > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4prof -- python3 bench.py --families 200000 --organisms 5000 --steps 20 --warmup 4 --repeats 3 $Q > $O/c4p.json 2>/dev/null
python3 profiles/dropin_logged.py > $O/dropin_logged.json 2>$O/dropin_logged.err; cat $O/dropin_logged.json | tr -d '\n' | cut -c1-1500; echo

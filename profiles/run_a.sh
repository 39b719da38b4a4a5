cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
python3 profiles/halfsum_phases.py > $O/halfsum_phases.json 2>&1; cat $O/halfsum_phases.json | tr -d '\n ' ; echo
timeout -k 10 600 python3 -m pytest tests/test_halfsum.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 400 > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -5 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python3 bench.py --families 200000 --organisms 5000 --steps 100 --warmup 10 --repeats 7 $Q > $O/c4.json 2>$O/c4.err; echo c4 rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4prof -- python3 bench.py --families 200000 --organisms 5000 --steps 20 --warmup 4 --repeats 3 $Q > $O/c4p.json 2>/dev/null
echo done

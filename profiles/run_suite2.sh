cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/$1 && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -v -m gpu --timeout 400 > $O/tests.log 2>&1; echo tests rc=$?; tail -3 $O/tests.log
python3 bench.py --families 200000 --organisms 5000 --steps 100 --warmup 10 --repeats 7 $Q > $O/c4.json 2>$O/c4.err; echo c4 rc=$?
python3 bench.py --families 50000 --organisms 1000 --steps 220 --warmup 22 --repeats 9 $Q > $O/c3.json 2>/dev/null; echo c3 rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4prof -- python3 bench.py --families 200000 --organisms 5000 --steps 20 --warmup 4 --repeats 3 $Q > $O/c4p.json 2>/dev/null
echo done

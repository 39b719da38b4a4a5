cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/$1 && mkdir -p $O
timeout -k 10 1000 python3 -u -m pytest tests -x -v -m gpu --timeout 400 -o faulthandler_timeout=380 > $O/gpu_all.log 2>&1; echo all rc=$?; tail -4 $O/gpu_all.log
python3 profiles/random_starts.py > $O/random_starts.json 2> $O/random_starts.err; echo rs rc=$?
python3 bench.py --dist --steps 700 --warmup 70 --families 20000 --organisms 500 --no-cpu-baseline --no-north-star > $O/dist_c2.json 2>/dev/null; echo rc=$?
echo done

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/$1 && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu --timeout 400 > $O/tests.log 2>&1; echo tests rc=$?; tail -3 $O/tests.log
for tag in after before; do
  if [ $tag = before ]; then export NEM_MI355X_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libnem_before.so; else unset NEM_MI355X_LIB; fi
  python3 bench.py --steps 200 --warmup 20 --repeats 15 $Q > $O/c2_$tag.json 2>/dev/null
  python3 profiles/batch_lockstep.py 20000 500 16,64 > $O/batch_$tag.json 2>/dev/null
  echo $tag done
done
echo done

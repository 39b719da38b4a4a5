cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/$1 && mkdir -p $O
try() {
  tag=$1; shift
  hung=0
  for i in 1 2 3 4 5 6; do
    env "$@" python3 bench.py --dist --steps 700 --warmup 70 --families 20000 --organisms 500 --no-cpu-baseline --no-north-star > $O/$tag$i.json 2> $O/$tag$i.err &
    PID=$!
    for t in $(seq 1 40); do sleep 1; kill -0 $PID 2>/dev/null || break; done
    if kill -0 $PID 2>/dev/null; then hung=$((hung+1)); kill -9 $PID; wait $PID 2>/dev/null; else wait $PID; fi
  done
  echo "$tag: hung $hung of 6"
}
try base X=1
python3 bench.py --dist --steps 220 --warmup 22 --no-cpu-baseline > $O/dist_c3.json 2>/dev/null; echo c3 rc=$?
echo done

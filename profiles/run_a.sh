cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu --timeout 400 > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python3 bench.py --disper skd $Q > $O/skd.json 2>/dev/null; echo skd rc=$?
NEM_MI355X_FF=0 python3 bench.py --disper skd $Q > $O/skd_noff.json 2>/dev/null
python3 bench.py $Q > $O/c2.json 2>/dev/null
for k in 2 5 10; do python3 bench.py --k $k --steps 200 --warmup 20 --repeats 9 $Q > $O/k$k.json 2>/dev/null; NEM_MI355X_FF=0 python3 bench.py --k $k --steps 200 --warmup 20 --repeats 9 $Q > $O/k${k}_noff.json 2>/dev/null; done
python3 - <<'PY'
import json
for f in ['skd','skd_noff','c2','k2','k2_noff','k5','k5_noff','k10','k10_noff']:
    try:
        d=json.load(open('gpurun_out/r03z/%s.json'%f)); print(f, round(d['ms_per_step'],5), [(k['kernel'],round(k['avg_launch_ms']*1e3,2)) for k in d['roofline']['kernels']])
    except Exception as e: print(f, 'ERR', e)
PY

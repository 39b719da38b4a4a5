cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
for r in 1 2 3; do
python3 bench.py --families 200000 --organisms 5000 --steps 100 --warmup 10 --repeats 7 $Q > $O/c4_new_$r.json 2>/dev/null
NEM_MI355X_SWEEP_SPB=0 python3 bench.py --families 200000 --organisms 5000 --steps 100 --warmup 10 --repeats 7 $Q > $O/c4_old_$r.json 2>/dev/null
done
python3 bench.py --families 100000 --organisms 1000 --steps 200 --warmup 20 --repeats 7 $Q > $O/c100_new.json 2>/dev/null
NEM_MI355X_SWEEP_SPB=0 python3 bench.py --families 100000 --organisms 1000 --steps 200 --warmup 20 --repeats 7 $Q > $O/c100_old.json 2>/dev/null
python3 - <<'PY'
import json
for f in ['c4_new_1','c4_old_1','c4_new_2','c4_old_2','c4_new_3','c4_old_3','c100_new','c100_old']:
    d=json.load(open('gpurun_out/r03z/%s.json'%f)); print(f, round(d['ms_per_step'],5), round(d['ms_per_step_min'],5), [(k['kernel'],round(k['avg_launch_ms']*1e3,2)) for k in d['roofline']['kernels']])
PY

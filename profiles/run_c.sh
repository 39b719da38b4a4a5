cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
Q="--no-north-star --no-cpu-baseline"
for s in 100 300 100 300; do
python3 bench.py --families 200000 --organisms 5000 --steps $s --warmup $((s/10)) --repeats 7 $Q > $O/c4_$s.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/c4_$s.json')); print($s, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], [(k['kernel'], round(k['avg_launch_ms']*1e3,2)) for k in d['roofline']['kernels']])"
done

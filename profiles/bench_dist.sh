cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/$1 && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/tests.log 2>&1; echo tests rc=$?; tail -5 $O/tests.log
python3 bench.py --steps 20 --warmup 5 > $O/c2_driver.json 2>$O/c2_driver.err; echo driver rc=$?
python3 bench.py --dist --steps 220 --warmup 22 --no-cpu-baseline > $O/dist1.json 2>$O/dist1.err; echo dist rc=$?
python3 bench.py --gpus 2 --backend gloo --steps 110 --warmup 11 --repeats 5 --extras-strong-shape 60000x1500 > $O/gloo2.json 2>$O/gloo2.err; echo gloo2 rc=$?
echo done

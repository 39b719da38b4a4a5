cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "many or batch or robust" > $O/batch_tests.txt 2>&1; tail -2 $O/batch_tests.txt
python3 profiles/batch_chunks.py > $O/chunks.json 2> $O/chunks_prof.txt
python3 profiles/batch_chunks.py 256 > $O/chunks256.json 2> $O/chunks256.err

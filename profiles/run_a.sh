cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
NEM_MI355X_BATCH_PROF=1 python3 profiles/batch_chunks.py > $O/chunks.json 2> $O/chunks_prof.txt
NEM_MI355X_BATCH_PROF=1 python3 profiles/batch_chunks.py 256 > $O/chunks256.json 2> $O/chunks256.err; grep "solve_many\] 256 problems, 8 w" $O/chunks256.err | tail -3

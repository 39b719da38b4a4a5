cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
NEM_MI355X_BATCH_PROF=1 python3 profiles/batch_chunks.py > $O/chunks.json 2> $O/chunks_prof.txt
grep -B9 "solve_many\] 64 problems, 8 workers" $O/chunks_prof.txt | tail -22 | cut -c1-230

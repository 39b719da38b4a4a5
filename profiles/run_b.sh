cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r03z && mkdir -p $O
cp pangenomenem_amd/lib/libnem_mi355x.so /tmp/libnem_keep.so
NEM_EXTRA_HIPCC_FLAGS=-DNEM_PHASE_PROF python3 pangenomenem_amd/build.py --force > $O/build_phase.log 2>&1
python3 profiles/finish_phases.py 200000 5000 > $O/finish_phases.txt 2>&1
cat $O/finish_phases.txt
cp /tmp/libnem_keep.so pangenomenem_amd/lib/libnem_mi355x.so

mkdir -p gpurun_out/fm
export TMPDIR=/tmp
cd /tmp
for n in 8 32; do
N=$n STEPS=10 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fm/prof_fm_b$n --output-format csv -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $GRAFT_REPO_ROOT/gpurun_out/fm/prof_fm_b$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/fm/prof_fm_b8/*/ | head

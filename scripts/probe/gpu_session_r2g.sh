mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2g
python bench.py > $R/bench_v14.json 2> $R/bench_v14.err; echo "bench rc=$?"
cd /tmp
for n in 8 32; do
N=$n STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_fm_b$n -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/prof_fm_b$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
N=64 STEPS=6 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, "

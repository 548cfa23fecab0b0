set -x
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2e
python bench.py > $R/bench_v12.json 2> $R/bench_v12.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $R/bench_v12_driverlike.json 2> $R/bench_v12_driverlike.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_v12 -- python $GRAFT_REPO_ROOT/bench.py --steps 800 --warmup 40 --no-cpu-baseline --no-extras > $R/prof_v12.json 2> $R/prof_v12.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b1024k6 -- python $GRAFT_REPO_ROOT/bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/prof_b1024k6.json 2> $R/prof_b1024k6.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b128k32 -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/prof_b128k32.json 2> $R/prof_b128k32.err
for n in 8 32; do
N=$n STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_fm_b$n -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/prof_fm_b$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
ls $R

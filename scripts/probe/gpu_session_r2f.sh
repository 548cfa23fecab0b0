set -x
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2f
( time python bench.py > $R/bench_v13.json 2> $R/bench_v13.err ) 2> $R/bench_time.txt; echo "bench rc=$?"; tail -3 $R/bench_time.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/prof_v13 -- python $GRAFT_REPO_ROOT/bench.py --steps 800 --warmup 40 --no-cpu-baseline --no-extras > $R/prof_v13.json 2> $R/prof_v13.err
rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fetch_v13 -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras > $R/pmc_fetch_v13.json 2> $R/pmc_fetch_v13.err
rocprofv3 --pmc WRITE_SIZE -d $R/pmc_write_v13 -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras > $R/pmc_write_v13.json 2> $R/pmc_write_v13.err
rocprofv3 --kernel-trace --stats -d $R/prof_b1024k6 -- python $GRAFT_REPO_ROOT/bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/prof_b1024k6.json 2> $R/prof_b1024k6.err
rocprofv3 --kernel-trace --stats -d $R/prof_b128k32 -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/prof_b128k32.json 2> $R/prof_b128k32.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py stats $R/prof_v13 > $R/r02_v13_kernel_stats.csv
python scripts/pmc_summary.py hbm $R/pmc_fetch_v13 $R/pmc_write_v13 > $R/r02_v13_pmc_hbm.csv
python scripts/pmc_summary.py stats $R/prof_b1024k6 > $R/r02_v13_b1024_k6_kernel_stats.csv
python scripts/pmc_summary.py stats $R/prof_b128k32 > $R/r02_v13_b128_k32_kernel_stats.csv
cat $R/r02_v13_kernel_stats.csv | head; cat $R/r02_v13_pmc_hbm.csv

set -x
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/r2/t13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/t9.log; tail -4 gpurun_out/r2/t9.log
python bench.py > gpurun_out/r2/bench_v11.json 2> gpurun_out/r2/bench_v11.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_v11 -- python bench.py --steps 800 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/r2/prof_v11.json 2> gpurun_out/r2/prof_v11.err
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2/pmc_fetch_v11 -- python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras > gpurun_out/r2/pmc_fetch_v11.json 2> gpurun_out/r2/pmc_fetch_v11.err
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2/pmc_write_v11 -- python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras > gpurun_out/r2/pmc_write_v11.json 2> gpurun_out/r2/pmc_write_v11.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_b1024k6_v11 -- python bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r2/prof_b1024k6_v11.json 2> gpurun_out/r2/prof_b1024k6_v11.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_b128k32_v11 -- python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r2/prof_b128k32_v11.json 2> gpurun_out/r2/prof_b128k32_v11.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_b128k6_v11 -- python bench.py --batch 128 --steps 200 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r2/prof_b128k6_v11.json 2> gpurun_out/r2/prof_b128k6_v11.err
ls gpurun_out/r2 | head -80

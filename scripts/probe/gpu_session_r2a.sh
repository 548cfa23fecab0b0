set -x
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/r2/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/t2.log; tail -4 gpurun_out/r2/t2.log
python scripts/probe/h2d_probe.py > gpurun_out/r2/h2d.log 2>&1
for cfg in "64 6" "1024 6"; do set -- $cfg; for w in pa pb; do B=$1 K=$2 WHICH=$w python scripts/diag_stamps.py > gpurun_out/r2/stamps_${w}_b$1.log 2>&1; done; done
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_b1024k6 -- python bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r2/prof_b1024k6.json 2> gpurun_out/r2/prof_b1024k6.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_b128k32 -- python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r2/prof_b128k32.json 2> gpurun_out/r2/prof_b128k32.err
SKIP_LIB=1 N=1024 rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_enc -- python scripts/bench_encoder.py > gpurun_out/r2/prof_enc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/r2/pmc_sq_b1024 -- python bench.py --batch 1024 --steps 16 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r2/pmc_sq_b1024.json 2> gpurun_out/r2/pmc_sq_b1024.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/r2/pmc_sq_b64 -- python bench.py --steps 64 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r2/pmc_sq_b64.json 2> gpurun_out/r2/pmc_sq_b64.err
ls gpurun_out/r2

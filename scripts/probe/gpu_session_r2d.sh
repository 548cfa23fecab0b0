set -x
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
# 8 waves per tile: parity first, then A/B against 4 waves
NDP_PHASE_WAVES=8 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_gan.py -m gpu -q -x > gpurun_out/r2/t_w8.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2/t_w8.log
for w in 4 8 4 8; do
  NDP_PHASE_WAVES=$w python bench.py --steps 800 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/r2/w${w}.json 2> gpurun_out/r2/w${w}.err; python -c "
import json,sys
d=json.loads(open('gpurun_out/r2/w${w}.json').read().strip().splitlines()[-1]); print('waves ${w}:', d['value'], d['ms_per_step'])"
done
export NDP_PHASE_WAVES=8
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_w8 -- python bench.py --steps 800 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/r2/prof_w8.json 2> gpurun_out/r2/prof_w8.err
python - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/r2/prof_w8/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY

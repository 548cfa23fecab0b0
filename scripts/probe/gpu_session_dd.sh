mkdir -p gpurun_out/fm
export TMPDIR=/tmp
for dd in 0 1 0 1; do
NDP_STEP_CODE_ROWS=$dd python bench.py --batch 8 --num-sample 32 --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('code rows $dd: B=8 K=32 (112 tiles)', d['ms_per_step'], {n: k[n]['avg_us'] for n in k})"
done
for dd in 0 1; do
NDP_STEP_CODE_ROWS=$dd python bench.py --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('code rows $dd: config 2', d['value'], d['ms_per_step'])"
done

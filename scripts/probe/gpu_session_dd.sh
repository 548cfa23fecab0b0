mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/fm/t_dd.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed\|Error" gpurun_out/fm/t_dd.log | head -20
for dd in 0 1 0 1; do
NDP_STEP_SHARED_CODE_ROW=$dd python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('shared code row $dd: B=128 K=32', d['ms_per_step'], {n: k[n]['avg_us'] for n in k})"
done
NDP_STEP_SHARED_CODE_ROW=0 python bench.py --batch 64 --num-sample 16 --steps 64 --warmup 16 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off: B=64 K=16', d['ms_per_step'])"
NDP_STEP_SHARED_CODE_ROW=1 python bench.py --batch 64 --num-sample 16 --steps 64 --warmup 16 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('on : B=64 K=16', d['ms_per_step'])"

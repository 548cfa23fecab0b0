mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/fm/t_dd.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed\|Error" gpurun_out/fm/t_dd.log | head -20
for i in 1 2 3; do
python bench.py --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('config 2', d['value'], d['ms_per_step'], {n: k[n]['avg_us'] for n in k})"
done
python bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1024', d['ms_per_step'])"
WHICH=pa B=64 K=6 WARM=3 timeout -k 10 300 python scripts/diag_stamps.py 2>&1 | grep "phase  0\|prologue\|first start"
WHICH=pb B=64 K=6 WARM=3 timeout -k 10 300 python scripts/diag_stamps.py 2>&1 | grep "phase  0\|first start"

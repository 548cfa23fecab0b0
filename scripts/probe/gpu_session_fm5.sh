export TMPDIR=/tmp
mkdir -p gpurun_out/fm
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_gan.py tests/test_gpu_p2p.py -m gpu -q > gpurun_out/fm/t15.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/fm/t15.log
python bench.py --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2', d['value'], d['ms_per_step'])"

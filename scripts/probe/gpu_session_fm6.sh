mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py tests/test_gpu_encoder.py -m gpu -q > gpurun_out/fm/t4.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed" gpurun_out/fm/t4.log | head -20
N=8 python scripts/probe/fm_time.py > gpurun_out/fm/time_b8.log 2>&1; head -14 gpurun_out/fm/time_b8.log
N=32 STEPS=10 python scripts/probe/fm_time.py > gpurun_out/fm/time_b32.log 2>&1; head -12 gpurun_out/fm/time_b32.log

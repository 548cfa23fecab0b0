mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -m gpu -q > gpurun_out/fm/t7.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed\|Error" gpurun_out/fm/t7.log | head -20

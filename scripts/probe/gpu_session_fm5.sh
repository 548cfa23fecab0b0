export TMPDIR=/tmp
mkdir -p gpurun_out/fm
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -m gpu -q -k "cli_entry" > gpurun_out/fm/t13.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed\|Error" gpurun_out/fm/t13.log | head

mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/fm/t_all.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed" gpurun_out/fm/t_all.log | head -20

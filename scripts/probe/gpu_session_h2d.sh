mkdir -p gpurun_out/fm
export TMPDIR=/tmp
for i in 1 2 3; do
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "uploads_from_pinned" > gpurun_out/fm/h2d_$i.log 2>&1; echo "alone run $i rc=$?"
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/fm/h2d_file.log 2>&1; echo "whole file rc=$?"; tail -3 gpurun_out/fm/h2d_file.log

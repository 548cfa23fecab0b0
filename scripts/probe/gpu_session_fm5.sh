mkdir -p gpurun_out/fm
export TMPDIR=/tmp
GRAPH=1 N=8 python scripts/probe/fm_time.py > gpurun_out/fm/time_b8.log 2>&1; grep "ms/step" gpurun_out/fm/time_b8.log
GRAPH=1 N=32 STEPS=10 python scripts/probe/fm_time.py > gpurun_out/fm/time_b32.log 2>&1; grep "ms/step" gpurun_out/fm/time_b32.log

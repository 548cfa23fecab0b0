mkdir -p gpurun_out/fm
export TMPDIR=/tmp
N=8 python scripts/probe/fm_time.py > gpurun_out/fm/time_b8.log 2>&1; head -30 gpurun_out/fm/time_b8.log
N=32 STEPS=10 python scripts/probe/fm_time.py > gpurun_out/fm/time_b32.log 2>&1; head -45 gpurun_out/fm/time_b32.log

mkdir -p gpurun_out/fm
export TMPDIR=/tmp
NDP_FM_SIDE_STREAM=0 N=8 python scripts/probe/fm_time.py > gpurun_out/fm/time_b8.log 2>&1; cat gpurun_out/fm/time_b8.log | head -70

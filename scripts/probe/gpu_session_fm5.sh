mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -m gpu -q > gpurun_out/fm/t14.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed\|Error" gpurun_out/fm/t14.log | head
NDP_FM_SIDE_STREAM=0 N=8 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, \|conv3x3\|refine1"
NDP_FM_SIDE_STREAM=0 N=32 STEPS=10 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, \|conv3x3\|refine1"
N=8 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, "
N=32 STEPS=10 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, "

mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -m gpu -q > gpurun_out/fm/t12.log 2>&1; echo "pytest rc=$?"; grep -n "^E   \|^FAILED\|passed\|failed\|Error" gpurun_out/fm/t12.log | head
for v in 0 1 0 1; do
echo "BN64=$v"
NDP_FM_BN64=$v NDP_FM_SIDE_STREAM=0 N=8 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, \|gemm\[conv1\]\|gemm\[deconv5\]\|dgrad\[conv2\]"
NDP_FM_BN64=$v NDP_FM_SIDE_STREAM=0 N=32 STEPS=10 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, \|gemm\[conv1\]\|gemm\[deconv5\]\|dgrad\[conv2\]"
done

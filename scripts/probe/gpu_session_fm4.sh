mkdir -p gpurun_out/fm
export TMPDIR=/tmp
N=3 SEED=14 DATA_SEED=23 FOCUS=decoder.deconv1.weight timeout -k 10 400 python scripts/probe/fm_debug.py > gpurun_out/fm/debug_n3b.log 2>&1; echo "rc=$?"
grep -v "bias  \|_bn" gpurun_out/fm/debug_n3b.log | tail -48

set -x
mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 400 python scripts/probe/fm_debug.py > gpurun_out/fm/debug1.log 2>&1; echo "rc=$?"
tail -120 gpurun_out/fm/debug1.log

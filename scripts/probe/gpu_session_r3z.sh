R=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -x -q > $R/t.log 2>&1; echo "fm tests rc=$?"; tail -5 $R/t.log
N=8 STEPS=30 timeout -k 10 300 python scripts/probe/fm_time.py > $R/fm8.log 2>&1 && head -24 $R/fm8.log
N=32 STEPS=20 timeout -k 10 300 python scripts/probe/fm_time.py > $R/fm32.log 2>&1 && head -24 $R/fm32.log

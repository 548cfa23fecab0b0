set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3u
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -x -q > $R/t_fm.log 2>&1; echo "fm rc=$?"; tail -30 $R/t_fm.log
N=8 STEPS=20 timeout -k 10 300 python scripts/probe/fm_time.py > $R/fm8.log 2>&1; head -8 $R/fm8.log

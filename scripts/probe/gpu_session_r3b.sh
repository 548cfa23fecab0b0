# round 3, session b: forward-model tests after the epilogue-statistics change, then its timing
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_forward_model.py -x -q > $R/tests_fm.log 2>&1; rc=$?; echo "fm tests rc=$rc"; tail -25 $R/tests_fm.log
[ $rc -eq 0 ] || exit $rc
N=8 STEPS=20 timeout -k 10 300 python scripts/probe/fm_time.py > $R/fm8.log 2>&1 && N=32 STEPS=10 timeout -k 10 300 python scripts/probe/fm_time.py > $R/fm32.log 2>&1
head -24 $R/fm8.log; head -22 $R/fm32.log

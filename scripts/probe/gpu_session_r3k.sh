set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_forward_model.py -x -q -s -k "second_iteration or golden" > $R/t_fm.log 2>&1; echo "fm rc=$?"; grep -n "per iteration\|passed\|failed\|Error\|assert" $R/t_fm.log | head -20
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "fifty" > $R/t_50.log 2>&1; echo "50 rc=$?"; grep -n "worst\|passed\|failed" $R/t_50.log | head

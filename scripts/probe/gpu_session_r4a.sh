R=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $R/t.log 2>&1; echo "gpu suite rc=$?"; tail -4 $R/t.log

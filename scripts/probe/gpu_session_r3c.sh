# round 3, session c: forward-model tests (bucketed exchange), the bench launcher rehearsal, the whole GPU suite
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $R/tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -30 $R/tests.log

mkdir -p gpurun_out/fm
export TMPDIR=/tmp
WHICH=pa B=64 K=6 WARM=3 timeout -k 10 300 python scripts/diag_stamps.py > gpurun_out/fm/stamps_pa.log 2>&1; echo rc=$?; tail -22 gpurun_out/fm/stamps_pa.log
WHICH=pb B=64 K=6 WARM=3 timeout -k 10 300 python scripts/diag_stamps.py > gpurun_out/fm/stamps_pb.log 2>&1; echo rc=$?; tail -16 gpurun_out/fm/stamps_pb.log

R=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $R
export TMPDIR=/tmp
for e in 0 1 2 3 4 6; do EXTRA=$e timeout -k 10 120 python scripts/probe/queues_probe.py 2>/dev/null | tail -1; done
for e in 2 3 4 6; do NDP_FM_SIDE_PRIORITY=normal EXTRA=$e timeout -k 10 120 python scripts/probe/queues_probe.py 2>/dev/null | tail -1; done
for e in 3 6; do GPU_MAX_HW_QUEUES=2 EXTRA=$e timeout -k 10 120 python scripts/probe/queues_probe.py 2>/dev/null | tail -1; done
for e in 3 6; do GPU_MAX_HW_QUEUES=3 EXTRA=$e timeout -k 10 120 python scripts/probe/queues_probe.py 2>/dev/null | tail -1; done

R=$GRAFT_REPO_ROOT/gpurun_out/r3x
mkdir -p $R
export TMPDIR=/tmp
for pr in low normal; do for n in 8 32; do
NDP_FM_SIDE_PRIORITY=$pr N=$n STEPS=20 timeout -k 10 300 python scripts/probe/fm_time.py 2>/dev/null | head -1 | sed "s/^/side priority $pr: /"
done; done
NDP_FM_SIDE_STREAM=0 N=8 STEPS=20 timeout -k 10 300 python scripts/probe/fm_time.py 2>/dev/null | head -1 | sed "s/^/no side stream: /"
NDP_FM_SIDE_STREAM=0 N=32 STEPS=20 timeout -k 10 300 python scripts/probe/fm_time.py 2>/dev/null | head -1 | sed "s/^/no side stream: /"

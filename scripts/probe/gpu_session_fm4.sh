mkdir -p gpurun_out/fm
export TMPDIR=/tmp
for n in 2 3; do
N=$n timeout -k 10 400 python scripts/probe/fm_debug.py > gpurun_out/fm/debug_n${n}d.log 2>&1; echo "rc=$?"
grep -A6 "post-ReLU map\|^d encoder" gpurun_out/fm/debug_n${n}d.log | grep -v "^--$" | head -16
done

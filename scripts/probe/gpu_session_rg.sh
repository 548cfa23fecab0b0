mkdir -p gpurun_out/fm
export TMPDIR=/tmp
for lib in libndp_hip.so libndp_rg24.so libndp_rg16.so libndp_hip.so libndp_rg24.so; do
for cfgs in "--batch 1024" "--batch 128 --num-sample 32"; do
NDP_LIB_PATH=$GRAFT_REPO_ROOT/ndivplanning_amd/lib/$lib python bench.py $cfgs --steps 64 --warmup 16 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$cfgs', d['ms_per_step'], d['roofline']['kernels']['k_phase_a']['avg_us'])"
done; done

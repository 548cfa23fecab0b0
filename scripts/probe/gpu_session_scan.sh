export TMPDIR=/tmp
run() { NDP_LIB_PATH=$GRAFT_REPO_ROOT/ndivplanning_amd/lib/$1 python bench.py --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do run libndp_hip.so; run libndp_wpf6.so; run libndp_wpf10.so; done

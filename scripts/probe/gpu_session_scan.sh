export TMPDIR=/tmp
run() { python bench.py --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do run base; NDP_WGRAD_HG=24 NDP_WGRAD_LG=12 run "LG=12"; NDP_WGRAD_HG=24 NDP_WGRAD_LG=16 run "LG=16"; done

export TMPDIR=/tmp
run() { python bench.py --steps 1600 --warmup 160 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('$1', d['value'], d['ms_per_step'], k['k_wgrad[D]']['avg_us'], k['k_wgrad[G]']['avg_us'], k['k_reduce_adam']['avg_us'])"; }
run base
for hd in 10 12 16; do NDP_WGRAD_HD=$hd NDP_WGRAD_LD=8 run "HD=$hd LD=8"; done
NDP_WGRAD_HD=12 NDP_WGRAD_LD=6 run "HD=12 LD=6"
NDP_WGRAD_HD=8 NDP_WGRAD_LD=4 run "HD=8 LD=4"
for hg in 16 20 28 32; do NDP_WGRAD_HG=$hg NDP_WGRAD_LG=8 run "HG=$hg LG=8"; done
NDP_WGRAD_HG=24 NDP_WGRAD_LG=4 run "HG=24 LG=4"
NDP_WGRAD_HG=24 NDP_WGRAD_LG=12 run "HG=24 LG=12"
run base

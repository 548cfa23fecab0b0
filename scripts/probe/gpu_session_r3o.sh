set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3o
mkdir -p $R
export TMPDIR=/tmp
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], {n: v["avg_us"] for n, v in k.items()})
PY
}
for cfg in "8 8" "16 8" "12 8" "16 4" "24 8" "8 4" "12 4" "8 8"; do
set -- $cfg
NDP_WGRAD_HD=$1 NDP_WGRAD_LD=$2 timeout -k 10 200 python bench.py --steps 2000 --warmup 200 --no-extras --no-cpu-baseline > $R/hd$1_ld$2.json 2> $R/hd$1_ld$2.err; show $R/hd$1_ld$2.json
done

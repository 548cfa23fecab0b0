set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $R
export TMPDIR=/tmp
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["roofline"]["whole_step"]["frac"], {n: v["avg_us"] for n, v in k.items()})
PY
}
for cfg in "0 0" "128 0" "128 64" "128 48" "192 64" "96 32" "160 32"; do
set -- $cfg
if [ "$2" = "0" ]; then unset NDP_WGRAD_CHUNKS; else export NDP_WGRAD_CHUNKS=$2; fi
NDP_WGRAD_WIDE=$1 timeout -k 10 200 python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/w$1_c$2.json 2> $R/w$1_c$2.err; show $R/w$1_c$2.json
done

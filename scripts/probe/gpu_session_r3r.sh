set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $R
export TMPDIR=/tmp
rc=0
[ $rc -eq 0 ] || exit $rc
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["roofline"]["whole_step"]["frac"], {n: v["avg_us"] for n, v in k.items()})
PY
}
for w in 0 on; do
if [ $w = 0 ]; then export NDP_WGRAD_WIDE=0; else unset NDP_WGRAD_WIDE; fi
for sh in "128 32" "1024 6" "768 6"; do
set -- $sh
timeout -k 10 200 python bench.py --batch $1 --num-sample $2 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b$1k$2_$w.json 2> $R/b$1k$2_$w.err; show $R/b$1k$2_$w.json
done
done

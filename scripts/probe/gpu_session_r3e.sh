set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $R
export TMPDIR=/tmp
for ex in forward_model config4,forward_model large_m,forward_model h2d_per_launch,forward_model; do
NDP_BENCH_EXTRAS=$ex timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/b_$ex.json 2> $R/b_$ex.err
python - $R/b_$ex.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm=d.get("forward_model",{})
print(sys.argv[1].split("/")[-1], d.get("extras_failed"), [(k, fm[k]["ms_per_step"], fm[k]["repeat_ms_per_step"]) for k in ("batch8","batch32") if k in fm])
PY
done

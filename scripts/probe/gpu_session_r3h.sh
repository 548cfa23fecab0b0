set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $R
export TMPDIR=/tmp
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm=d.get("forward_model",{})
print(sys.argv[1].split("/")[-1], d.get("extras_failed"), [(k, fm[k]["ms_per_step"]) for k in ("batch8","batch32") if k in fm])
PY
}
export NDP_BENCH_EXTRAS=h2d_per_launch,config4,forward_model
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/q8.json 2> $R/q8.err; show $R/q8.json
GPU_MAX_HW_QUEUES=2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/q2.json 2> $R/q2.err; show $R/q2.json
NDP_FM_SIDE_STREAM=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/noside.json 2> $R/noside.err; show $R/noside.json

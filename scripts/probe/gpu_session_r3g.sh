set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $R
export TMPDIR=/tmp
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm=d.get("forward_model",{})
print(sys.argv[1].split("/")[-1], d.get("extras_failed"), [(k, fm[k]["ms_per_step"]) for k in ("batch8","batch32") if k in fm])
PY
}
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/all_nocpu.json 2> $R/all_nocpu.err; show $R/all_nocpu.json
NDP_BENCH_EXTRAS=forward_model timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $R/fm_cpu.json 2> $R/fm_cpu.err; show $R/fm_cpu.json
NDP_BENCH_EXTRAS=h2d_per_launch,config4,forward_model timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/hcf.json 2> $R/hcf.err; show $R/hcf.json

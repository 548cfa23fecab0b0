set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/b_drv.json 2> $R/b_drv.err; echo rc=$?
python - $R/b_drv.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm=d.get("forward_model",{})
print(d["value"], d.get("extras_failed"), [(k, fm[k]["ms_per_step"], fm[k]["frac_of_fp32_mfma_peak"], fm[k]["launches_per_step"]) for k in ("batch8","batch32") if k in fm])
print({k: v.get("ms_per_step") for k, v in d.get("large_m", {}).items()}, d["config4"]["image_step_ms_with_upload"], d["config4"]["image_step_ms"], d["h2d_per_launch"]["steps_per_sec"])
PY

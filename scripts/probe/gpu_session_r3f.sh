set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $R
export TMPDIR=/tmp
for tag in drv def; do
if [ $tag = drv ]; then A="--gpus 1 --steps 20 --warmup 5"; else A=""; fi
timeout -k 10 400 python bench.py $A > $R/b_$tag.json 2> $R/b_$tag.err
python - $R/b_$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm=d.get("forward_model",{})
print(sys.argv[1].split("/")[-1], d["value"], d.get("extras_failed"), [(k, fm[k]["ms_per_step"], fm[k]["repeat_ms_per_step"]) for k in ("batch8","batch32") if k in fm])
print({k: v.get("ms_per_step") for k, v in d.get("large_m", {}).items()}, d["config4"]["image_step_ms_with_upload"], d["h2d_per_launch"]["steps_per_sec"])
PY
done

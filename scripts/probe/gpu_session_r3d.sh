# round 3, session d: byte-frame tests, then the default bench (config4 with upload, forward model, headline)
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_forward_model.py -x -q > $R/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $R/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"; tail -3 $R/bench.err
python - <<'PY'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r3d/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("extras_failed"))
print(json.dumps(d.get("config4"), indent=0)[:1800])
fm=d.get("forward_model",{})
for k in ("batch8","batch32"):
    if k in fm: print(k, fm[k]["ms_per_step"], fm[k]["frac_of_fp32_mfma_peak"], fm[k]["launches_per_step"])
PY

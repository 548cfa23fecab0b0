# round 3: whole GPU suite, smoke, and the driver's bench invocation
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $R/tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -6 $R/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $R/smoke.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_drv.json 2> $R/bench_drv.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3w/bench_drv.json").read().strip().splitlines()[-1])
fm=d.get("forward_model",{})
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("mfma_busy_counter"), d.get("extras_failed"))
print([(k, fm[k]["ms_per_step"], fm[k]["frac_of_fp32_mfma_peak"], fm[k]["launches_per_step"]) for k in ("batch8","batch32") if k in fm])
print({k: (v.get("ms_per_step"), v.get("whole_step_frac_of_fp32_mfma_peak")) for k, v in d.get("large_m", {}).items()}, d["config4"]["image_step_ms_with_upload"], d["h2d_per_launch"]["steps_per_sec"], d["cpu_baseline"]["value"])
PY

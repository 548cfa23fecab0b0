R=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_bench_launcher.py -x -q > $R/t.log 2>&1; echo "launcher rc=$?"; tail -5 $R/t.log
env NDP_BENCH_ONE_GPU=1 NDP_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > $R/n2.json 2> $R/n2.err; echo "n2 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3y/n2.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["config"]["gradient_exchange"], d["config"]["replicas_bit_identical"], d.get("extras_failed"))
print(json.dumps(d["forward_model_dp"], indent=0)[:1500])
PY

set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $R/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -5 $R/parity.log
[ $rc -eq 0 ] || exit $rc
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["roofline"]["whole_step"]["frac"], {n: (v["avg_us"], v["tflops"]) for n, v in k.items()})
PY
}
for w in 0 64 128 256; do
NDP_WGRAD_WIDE=$w timeout -k 10 200 python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b128k32_w$w.json 2> $R/b128k32_w$w.err; show $R/b128k32_w$w.json
done
for w in 0 128; do
NDP_WGRAD_WIDE=$w timeout -k 10 200 python bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b1024_w$w.json 2> $R/b1024_w$w.err; show $R/b1024_w$w.json
done

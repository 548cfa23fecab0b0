set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $R/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 $R/parity.log
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
timeout -k 10 200 python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b128k32_$w.json 2> $R/b128k32_$w.err; show $R/b128k32_$w.json
timeout -k 10 200 python bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b1024_$w.json 2> $R/b1024_$w.err; show $R/b1024_$w.json
timeout -k 10 200 python bench.py --batch 512 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b512_$w.json 2> $R/b512_$w.err; show $R/b512_$w.json
done

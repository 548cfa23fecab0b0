set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "free_running or config5 or decoder_forward or discriminator_forward" > $R/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 $R/parity.log
[ $rc -eq 0 ] || exit $rc
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["roofline"]["whole_step"]["frac"], {n: v["avg_us"] for n, v in k.items()})
PY
}
NDP_WGRAD_WIDE_SPLIT=1 timeout -k 10 200 python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/split1.json 2> $R/split1.err; show $R/split1.json
for w in 48 64 80 96 128 160; do
NDP_WGRAD_WIDE=$w timeout -k 10 200 python bench.py --batch 128 --num-sample 32 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/s2_w$w.json 2> $R/s2_w$w.err; show $R/s2_w$w.json
done
for w in 64 96; do
NDP_WGRAD_WIDE=$w timeout -k 10 200 python bench.py --batch 1024 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $R/b1024_s2_w$w.json 2> $R/b1024_s2_w$w.err; show $R/b1024_s2_w$w.json
done

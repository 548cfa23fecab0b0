mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
export NDP_BENCH_ONE_GPU=1 NDP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
NDP_DP_EXCHANGE=p2p timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 64 --warmup 16 > gpurun_out/r2h/n2.json 2> gpurun_out/r2h/n2.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r2h/n2.json').read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'], d['config']['gradient_exchange'], d.get('extras_failed'))
print(d.get('forward_model_dp'))
PY

mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
export NDP_BENCH_ONE_GPU=1 NDP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for ex in p2p rccl; do
NDP_DP_EXCHANGE=$ex timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 64 --warmup 16 > gpurun_out/r2f/n2_$ex.json 2> gpurun_out/r2f/n2_$ex.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r2f/n2_$ex.json').read().strip().splitlines()[-1])
    print('$ex', d['value'], d['n_gpus'], d['config']['gradient_exchange'], d['config'].get('exchange_note'), d['config']['replicas_bit_identical'], list(k for k in d if k in ('strong_config3','config5_shard','headline_through_rccl','extras_failed')))
    for k in ('strong_config3','config5_shard'):
        if k in d: print(' ', k, d[k]['ms_per_step'], d[k].get('gradient_exchange'), d[k].get('replicas_bit_identical'))
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/r2f/n2_$ex.err').read()[-1500:])
PY
done

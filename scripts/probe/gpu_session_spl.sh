export TMPDIR=/tmp
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-like', d['value'], d['ms_per_step'], d['config']['steps_per_graph_launch'], d['config']['repeats'])"
done

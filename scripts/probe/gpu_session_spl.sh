export TMPDIR=/tmp
for spl in 8 16 32 64; do
python bench.py --steps 1920 --warmup 192 --steps-per-launch $spl --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spl $spl', d['value'], d['ms_per_step'])"
done

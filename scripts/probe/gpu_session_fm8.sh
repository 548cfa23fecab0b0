mkdir -p gpurun_out/fm
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 400 --warmup 40 > gpurun_out/fm/bench_fm.json 2> gpurun_out/fm/bench_fm.err; echo "bench rc=$?"; tail -3 gpurun_out/fm/bench_fm.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/fm/bench_fm.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('extras_failed'))
print(json.dumps(d.get('forward_model'), indent=1)[:6000])
PY

set -x
mkdir -p gpurun_out/fm
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_forward_model.py -m gpu -q > gpurun_out/fm/t1.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/fm/t1.log
N=8 python scripts/probe/fm_time.py 2>&1 | tail -3
N=32 STEPS=10 python scripts/probe/fm_time.py 2>&1 | tail -3
cd /tmp && N=8 STEPS=10 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fm/prof_b8 --output-format csv -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $GRAFT_REPO_ROOT/gpurun_out/fm/prof_b8.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/fm/prof_b8/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:25]:
        print('%-70s %6s calls avg %9.1f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY

export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
export NDP_FM_SIDE_STREAM=0
N=32 STEPS=2 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/fm32 -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/fm32.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc/fm32/**/*counter_collection.csv', recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    name=r['Kernel_Name'][:40]; g=r.get('Grid_Size') or r.get('Grid_Size_X'); key=(name, g)
    agg[key][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[key]+=1
for key,c in sorted(agg.items(), key=lambda kv:-kv[1].get('SQ_BUSY_CYCLES',0))[:14]:
    n=max(cnt[key],1)
    busy=c['SQ_BUSY_CYCLES']/n; 
    print(key, 'n',n, 'busy %.0f'%busy, 'mfma_busy/busy %.3f'% (c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1)), 'wait_inst_any/wave %.3f'%(c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1)), 'wait_lds/wave %.3f'%(c['SQ_WAIT_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1)), 'vmem %.3f'%(c['SQ_ACTIVE_INST_VMEM']/max(c['SQ_WAVE_CYCLES'],1)), 'lds %.3f'%(c['SQ_ACTIVE_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1)), 'bankconf %.0f'%(c['SQ_LDS_BANK_CONFLICT']/n))
PY

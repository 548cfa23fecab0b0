# round 3, session 4b: evidence for profiles/ after the forward-model / encoder work of the second half of the round
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $R
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $R/t.log 2>&1; echo "gpu suite rc=$?"; tail -3 $R/t.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_steps20.json 2> $R/bench_steps20.err; echo "bench rc=$?"
cd /tmp
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ_B="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
TC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
N=8 STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ks_fm8 -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/ks_fm8.log 2>&1
N=32 STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ks_fm32 -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/ks_fm32.log 2>&1
SKIP_LIB=1 N=1024 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ks_enc -- python $GRAFT_REPO_ROOT/scripts/bench_encoder.py > $R/ks_enc.log 2>&1
for pass in A B C; do
  case $pass in A) CTR=$SQ_A;; B) CTR=$SQ_B;; C) CTR=$TC;; esac
  NDP_FM_SIDE_STREAM=0 N=8 STEPS=2 rocprofv3 --pmc $CTR --output-format csv -d $R/fm8_$pass -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/fm8_$pass.log 2>&1 || echo "fm8 $pass failed"
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py sq $R/fm8_A $R/fm8_B $R/fm8_C > $R/r03_end_fm8_pmc_sq.csv
cp $(find $R/ks_fm8 -name "*kernel_stats.csv" | head -1) $R/r03_end_fm_b8_kernel_stats.csv
cp $(find $R/ks_fm32 -name "*kernel_stats.csv" | head -1) $R/r03_end_fm_b32_kernel_stats.csv
cp $(find $R/ks_enc -name "*kernel_stats.csv" | head -1) $R/r03_end_encoder_kernel_stats.csv
find $R -name "*.db" -size +30M -delete; find $R -name "*kernel_trace.csv" -size +20M -delete; find $R -name "*counter_collection.csv" -size +20M -delete
head -3 $R/ks_fm8.log; head -3 $R/ks_fm32.log; cat $R/ks_enc.log | tail -9

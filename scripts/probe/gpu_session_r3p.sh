# round 3, session p: evidence for profiles/ -- kernel stats and counter passes of the round's final kernel set
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ_B="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
TC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
# kernel stats (durations)
rocprofv3 --kernel-trace --stats -d $R/ks_cfg2 -- $B --steps 800 --warmup 40 > $R/ks_cfg2.json 2> $R/ks_cfg2.err
rocprofv3 --kernel-trace --stats -d $R/ks_b128k32 -- $B --batch 128 --num-sample 32 --steps 64 --warmup 16 > $R/ks_b128k32.json 2> $R/ks_b128k32.err
rocprofv3 --kernel-trace --stats -d $R/ks_b1024 -- $B --batch 1024 --steps 64 --warmup 16 > $R/ks_b1024.json 2> $R/ks_b1024.err
N=8 STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ks_fm8 -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/ks_fm8.log 2>&1
N=32 STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ks_fm32 -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/ks_fm32.log 2>&1
# HBM bytes of the config-2 kernels
rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fetch -- $B --steps 128 --warmup 16 > $R/pmc_fetch.json 2> $R/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/pmc_write -- $B --steps 128 --warmup 16 > $R/pmc_write.json 2> $R/pmc_write.err
# SQ / cache counters: config-5 shard with the WIDE job, forward model with the epilogue statistics
for pass in A B C; do
  case $pass in A) CTR=$SQ_A;; B) CTR=$SQ_B;; C) CTR=$TC;; esac
  rocprofv3 --pmc $CTR --output-format csv -d $R/b128k32_$pass -- $B --batch 128 --num-sample 32 --steps 16 --warmup 4 > $R/b128k32_$pass.log 2>&1 || echo "b128k32 $pass failed"
  NDP_FM_SIDE_STREAM=0 N=8 STEPS=2 rocprofv3 --pmc $CTR --output-format csv -d $R/fm8_$pass -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/fm8_$pass.log 2>&1 || echo "fm8 $pass failed"
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py stats $R/ks_cfg2 > $R/r03_final_kernel_stats.csv
python scripts/pmc_summary.py stats $R/ks_b128k32 > $R/r03_final_b128_k32_kernel_stats.csv
python scripts/pmc_summary.py stats $R/ks_b1024 > $R/r03_final_b1024_k6_kernel_stats.csv
python scripts/pmc_summary.py hbm $R/pmc_fetch $R/pmc_write > $R/r03_final_pmc_hbm.csv
python scripts/pmc_summary.py sq $R/b128k32_A $R/b128k32_B $R/b128k32_C > $R/r03_final_b128k32_pmc_sq.csv
python scripts/pmc_summary.py sq $R/fm8_A $R/fm8_B $R/fm8_C > $R/r03_final_fm8_pmc_sq.csv
cp $(find $R/ks_fm8 -name "*kernel_stats.csv" | head -1) $R/r03_fm_b8_kernel_stats.csv
cp $(find $R/ks_fm32 -name "*kernel_stats.csv" | head -1) $R/r03_fm_b32_kernel_stats.csv
find $R -name "*.db" -size +30M -delete; find $R -name "*kernel_trace.csv" -size +20M -delete; find $R -name "*counter_collection.csv" -size +20M -delete
cat $R/r03_final_kernel_stats.csv; cat $R/r03_final_b128_k32_kernel_stats.csv; cat $R/r03_final_pmc_hbm.csv

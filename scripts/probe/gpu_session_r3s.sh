set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ_B="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
TC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
for pass in A B C; do
  case $pass in A) CTR=$SQ_A;; B) CTR=$SQ_B;; C) CTR=$TC;; esac
  SKIP_LIB=1 N=1024 rocprofv3 --pmc $CTR --output-format csv -d $R/enc_$pass -- python $GRAFT_REPO_ROOT/scripts/bench_encoder.py > $R/enc_$pass.log 2>&1 || echo "enc $pass failed"
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py sq $R/enc_A $R/enc_B $R/enc_C > $R/r03_encoder_pmc_sq.csv
cut -c1-200 $R/r03_encoder_pmc_sq.csv | head -12

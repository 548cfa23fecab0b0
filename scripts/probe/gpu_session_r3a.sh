# round 3, session a: full GPU test suite (new: bench.py --gpus 2 self-launch), then SQ / cache counter passes for the
# GAN kernels (config 2, config-5 shard) and the forward model (batch 8).  Each PMC pass is its own run (no trace
# domains beside --pmc); the program stands directly behind `--`.
set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $R/tests.log 2>&1; echo "tests rc=$?"; tail -5 $R/tests.log
rocprofv3 -L > $R/counters_full.txt 2>&1
cd /tmp
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ_B="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
TC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
for pass in A B C; do
  case $pass in A) CTR=$SQ_A;; B) CTR=$SQ_B;; C) CTR=$TC;; esac
  rocprofv3 --pmc $CTR --output-format csv -d $R/cfg2_$pass -- $B --steps 64 --warmup 16 > $R/cfg2_$pass.log 2>&1 || echo "cfg2 $pass failed"
  rocprofv3 --pmc $CTR --output-format csv -d $R/b128k32_$pass -- $B --batch 128 --num-sample 32 --steps 16 --warmup 4 > $R/b128k32_$pass.log 2>&1 || echo "b128k32 $pass failed"
  NDP_FM_SIDE_STREAM=0 N=8 STEPS=2 rocprofv3 --pmc $CTR --output-format csv -d $R/fm8_$pass -- python $GRAFT_REPO_ROOT/scripts/probe/fm_time.py > $R/fm8_$pass.log 2>&1 || echo "fm8 $pass failed"
done
cd $GRAFT_REPO_ROOT
for w in cfg2 b128k32 fm8; do
  python scripts/pmc_summary.py sq $R/${w}_A $R/${w}_B $R/${w}_C > $R/r03_${w}_pmc_sq.csv 2> $R/${w}_sq.err || echo "summary $w failed"
done
find $R -name "*counter_collection.csv" -size +20M -delete
cut -c1-160 $R/r03_cfg2_pmc_sq.csv | head -12

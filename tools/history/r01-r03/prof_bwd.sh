# usage: tools/prof_bwd.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
KB2="$R/long-context-attention_amd/kbench bwd 2 8192 8192 16 16 128 1 0 0 3"
rocprofv3 --kernel-trace --stats -d $OUT/bwd -o bwd -- $KB2 > $OUT/bwd.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq_bwd -o pmc -- $KB2 > $OUT/pmc_sq_bwd.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq2_bwd -o pmc -- $KB2 > $OUT/pmc_sq2_bwd.log 2>&1

# usage: tools/prof_fwd.sh <tag>   -- PMC passes on the C2 forward kernel via kbench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
KB="$R/long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 1 0 0 5"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o pmc -- $KB > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq2 -o pmc -- $KB > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_VALU_TRANS -d $OUT/pmc_sq3 -o pmc -- $KB > $OUT/pmc_sq3.log 2>&1
tail -3 $OUT/pmc_sq2.log $OUT/pmc_sq3.log
if [ ! -f $R/gpurun_out/counters_list.txt ]; then rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1; fi

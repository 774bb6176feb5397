#!/bin/bash
# usage (GPU box, repo root): bash tools/r04/prof_fwd.sh <tag> [causal(1)|full(0)]
# rocprofv3 kernel trace + PMC passes (SQ / GRBM in separate passes) of the forward at the C2 shape: the 4x64 kernel
# (default policy) and the 8x32 kernel (USP_FWD_WAVES=8); summary -> gpurun_out/prof_<tag>/summary.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r04fwd}; CAUSAL=${2:-1}
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"
for w in 64 8; do
  export USP_FWD_WAVES=$w
  KB="$R/long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 $CAUSAL 0 0"
  rocprofv3 --kernel-trace --stats -d $OUT/w$w/trace -o t -- $KB 60 > $OUT/w${w}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/w$w/pmc_sq -o pmc -- $KB 5 > $OUT/w${w}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 -d $OUT/w$w/pmc_sq2 -o pmc -- $KB 5 > $OUT/w${w}_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/w$w/pmc_grbm -o pmc -- $KB 5 > $OUT/w${w}_grbm.log 2>&1
  python $R/tools/prof_summary.py $OUT/w$w $OUT/summary_w$w.txt > /dev/null
done
cat $OUT/summary_w64.txt $OUT/summary_w8.txt > $OUT/summary.txt
rm -rf $OUT/w64 $OUT/w8

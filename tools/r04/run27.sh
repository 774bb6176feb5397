#!/bin/bash
# round 4, GPU call 27: backward kernels at the C2 shape through kbench: kernel trace + two PMC passes -> summary.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; OUT=$R/gpurun_out/prof_bwd27
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="$K bwd 2 8192 8192 16 16 128 1 0 0 10"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc2 -o p2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc3 -o p3 -- $CMD > $OUT/pmc3.log 2>&1
python3 $R/tools/prof_summary.py $OUT $OUT/summary.txt 2 8192 16 128 > /dev/null 2>&1
head -60 $OUT/summary.txt
find $OUT -name "*.db" -size +20M -delete

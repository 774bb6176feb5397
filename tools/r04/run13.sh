#!/bin/bash
# round 4, GPU call 14: dkdv64 without the per-tile memory round trip of the statistics wave.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for shape in "2 512 512 4 2 128 1 0" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 2048 2048 4 2 128 1 0" "1 1000 1300 3 3 128 0 0"; do
  timeout 300 $K bwd $shape 1 0 | cut -c1-170 || echo "RC=$? for $shape"
done
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_BWD_WAVES=$w
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME)"
done; done
unset USP_BWD_WAVES
export TMPDIR=/tmp; cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
KB="$K bwd 2 8192 8192 16 16 128 1 0 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04d/trace -o t -- $KB 12 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $R/gpurun_out/prof_r04d/pmc_sq -o pmc -- $KB 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_r04d/pmc_grbm -o pmc -- $KB 3 > /dev/null 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_r04d $R/gpurun_out/prof_r04d/summary.txt > /dev/null; rm -rf $R/gpurun_out/prof_r04d/trace $R/gpurun_out/prof_r04d/pmc_sq $R/gpurun_out/prof_r04d/pmc_grbm

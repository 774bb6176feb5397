#!/bin/bash
# round 4, GPU call 12: the one-wave-per-SIMD dK/dV kernel (usp_flash_bwd64.hip): kbench parity + A/B timing.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for shape in "1 256 256 1 1 128 0 0" "1 256 256 2 2 128 1 0" "2 512 512 4 2 128 1 0" "1 384 640 4 2 128 0 0" \
             "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 1 1 1 1 128 1 0" "2 77 77 2 2 128 1 0" \
             "1 2048 2048 4 2 128 1 0" "1 3000 5000 6 2 128 1 0" "1 5000 3000 4 4 128 1 0" "1 1000 1300 3 3 128 0 0"; do
  timeout 300 $K bwd $shape 1 0 | cut -c1-170 || echo "RC=$? for $shape"
done
USP_KBENCH_BWD_SPLITS=2,3 timeout 300 $K bwd 1 2048 2048 4 2 128 1 0 1 0 | cut -c1-170
USP_KBENCH_BWD_SPLITS=1,4 timeout 300 $K bwd 1 1000 1300 3 3 128 0 0 1 0 | cut -c1-170
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_BWD_WAVES=$w
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME)"
done; done
unset USP_BWD_WAVES
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04b/trace -o t -- $K bwd 2 8192 8192 16 16 128 1 0 0 12 > /dev/null 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_r04b $R/gpurun_out/prof_r04b/summary.txt > /dev/null; rm -rf $R/gpurun_out/prof_r04b/trace
grep -A8 "calls" $R/gpurun_out/prof_r04b/summary.txt | head -14

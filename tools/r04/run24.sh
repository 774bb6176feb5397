#!/bin/bash
# round 4, GPU call 24: the one-wave-per-SIMD dQ kernel (usp_flash_bwd_dq64.hip), first run: suite, timing against the
# 8-wave dQ kernel (USP_BWD_WAVES=8 forces both 8-wave backward kernels).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1500 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -40
for rep in 1 2 3; do
  echo "[new    ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[8-wave ] $(USP_BWD_WAVES=8 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new 64K ] $(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
echo "[new full] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME | cut -c60-150)"
timeout 300 $K bwd 2 8192 8192 16 16 128 1 0 1 0 | cut -c1-200
timeout 300 $K bwd 1 4096 4096 8 2 128 0 0 1 0 | cut -c1-200

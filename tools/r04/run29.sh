#!/bin/bash
# round 4, GPU call 29: forward 4x64 with the first-tile reference path and row-counted descriptors: suite, timing vs previous.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite fwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
for rep in 1 2 3; do
  echo "[new  causal] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
  echo "[prev causal] $(LD_LIBRARY_PATH=$R/abl/f_prev timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
done
echo "[new  full] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 100 | grep TIME | cut -c60-150)"
echo "[prev full] $(LD_LIBRARY_PATH=$R/abl/f_prev timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 100 | grep TIME | cut -c60-150)"
echo "[new  64K] $(timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME | cut -c60-150)"
echo "[prev 64K] $(LD_LIBRARY_PATH=$R/abl/f_prev timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME | cut -c60-150)"

#!/bin/bash
# round 4, GPU call 30: prologues reordered (first tiles' DMA in front of the resident-operand loads) in the three 64-row
# kernels: suite, timing against the previous library.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1500 $K suite 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
for rep in 1 2 3; do
  echo "[new  fwd] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
  echo "[prev fwd] $(LD_LIBRARY_PATH=$R/abl/prev2 timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
  echo "[new  bwd] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[prev bwd] $(LD_LIBRARY_PATH=$R/abl/prev2 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done

#!/bin/bash
# round 4, GPU calls 19, 20: dkdv64 with the chains started from the row constants (C operand), K pre-scaled; then cheaper
# descriptors, role B reading its first chain ahead of the barrier, the NaN tail behind every kbench tensor.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | head -30
for rep in 1 2 3; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[base] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new 64K ] $(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
echo "[base 64K] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
timeout 300 $K bwd 2 8192 8192 16 16 128 1 0 1 0 | cut -c1-200

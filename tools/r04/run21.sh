#!/bin/bash
# round 4, GPU call 21: (1) the suite with the NaN tail behind every kbench tensor (the kbench of call 20 was stale);
# (2) dkdv64's iteration anatomy from s_memtime stamps around the DMA drain and the barrier (b_tm build).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
echo "== anatomy (C2 shape) =="
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TM" | sort -k3n -k5n | awk 'NR<=64'
for rep in 1 2; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[base] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done

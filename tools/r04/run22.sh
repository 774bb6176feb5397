#!/bin/bash
# round 4, GPU call 22: dkdv64 after the issue-slot work (packing lag, one LDS wait per phase): suite, timing, anatomy.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
echo "== anatomy (C2 shape, workgroup 0: key block 0, 128 query tiles) =="
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TM" | sort -k3n -k5n | awk '{k=$3" "$5; if (c[k]++ < 2) print}'
for rep in 1 2 3; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[base] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new 64K ] $(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
echo "[base 64K] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"

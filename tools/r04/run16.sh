#!/bin/bash
# round 4, GPU call 16: after the pruning of the 8-wave kernels (dQ kernel re-generated) and the 64-bit cursors:
# native suite incl. backward, a head whose rows span 3 GiB, A/B timing.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | head -30
echo "== rows 2 MiB apart: 1500 rows span 3 GiB =="
USP_KBENCH_ROWSTRIDE=1048576 timeout 600 $K bwd 1 1500 1500 2 1 128 1 0 1 0 | cut -c1-170
USP_KBENCH_ROWSTRIDE=1048576 timeout 600 $K bwd 1 1300 1500 1 1 128 0 0 1 0 | cut -c1-170
echo "== timing =="
for rep in 1 2; do
  echo "$(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  echo "$(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "$(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME)"
  echo "[USP_BWD_WAVES=8] $(USP_BWD_WAVES=8 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
done

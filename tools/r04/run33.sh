#!/bin/bash
# round 4, GPU call 33: resident-operand loads of an item merged under one wait (dK/dV: 2 -> 1 round trips, dQ: 4 -> 2):
# native suite, kbench timing against the previous library, python product path.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; cd $R
timeout 1500 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
for rep in 1 2 3; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[prev] $(LD_LIBRARY_PATH=$R/abl/prev3 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new python] $(python tools/prof_product.py c2 40 2>/dev/null | tail -1)"

#!/bin/bash
# round 4, GPU call 28: where a causal forward item's time goes (s_memtime stamps, f_tm build), C2 shape.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
LD_LIBRARY_PATH=$R/abl/f_tm timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TF" | sort -k3n -k5n -k7n | awk '{k=$3" "$5" "$7; if (c[k]++ < 1) print}' | head -90
echo "[base causal] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
echo "[base full  ] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 100 | grep TIME | cut -c60-150)"

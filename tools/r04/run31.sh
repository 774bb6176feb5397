#!/bin/bash
# round 4, GPU call 31: per-item anatomy of the two backward kernels (s_memtime; b_tm / q_tm builds), C2 shape.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TI" | sort -k3n -k5n -k7n | awk '{k=$3" "$5" "$7; if (c[k]++ < 1) print}' | head -50
LD_LIBRARY_PATH=$R/abl/q_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TQ" | sort -k3n -k5n -k7n | awk '{k=$3" "$5" "$7; if (c[k]++ < 1) print}' | head -50

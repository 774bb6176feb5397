# round 5: per-item anatomy of the forward and the dQ kernel (s_memtime builds abl/f_tm, abl/q_tm) at the 64K rank-block shape
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
LD_LIBRARY_PATH=$R/abl/f_tm timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TF" | awk '{k=$3" "$7; if (c[k]++ < 1) print}' | sort -k3n -k7n | head -24
export USP_KBENCH_FLAGS=32        # USP_BWD_SKIP_DKDV: the dQ launch alone
LD_LIBRARY_PATH=$R/abl/q_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TQ" | awk '{k=$3" "$7; if (c[k]++ < 1) print}' | sort -k3n -k7n | head -24
timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 5 2>&1 | grep TIME
unset USP_KBENCH_FLAGS
timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 5 2>&1 | grep TIME

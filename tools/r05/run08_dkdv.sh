# round 5: dK/dV kernel variants -- correctness (native suite, NaN tails) + timing at C2, the 8-GPU rank block and the N = 1 workload
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 600 $K suite bwd 2>&1 | grep -E "FAIL|SUITE"
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
for i in 1 2; do
timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 2>&1 | grep TIME
timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 2>&1 | grep TIME
timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 2>&1 | grep TIME
done
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TM" | awk '{k=$3" "$5; if (c[k]++ < 1) print}' | sort -k3n -k5n | head -8
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TI" | awk '{k=$3" "$7; if (c[k]++ < 1) print}' | sort -k3n -k7n | head -8

# round 5: per-tile anatomy of the dK/dV kernel's two roles (s_memtime build abl/b_tm), GQA rank-block shape
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
mkdir -p $R/gpurun_out/r05
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TM" | awk '{k=$3" "$5; if (c[k]++ < 2) print}' | sort -k3n -k5n | head -40

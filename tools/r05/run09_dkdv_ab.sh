# round 5: same-box A/B of the dK/dV kernel: abl/b_old (round 4's) against the in-tree library, alternating
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
for i in 1 2 3; do
for lib in $R/abl/b_old $R/long-context-attention_amd; do
echo "== $(basename $lib)"
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 30 2>&1 | grep TIME
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 15 2>&1 | grep TIME
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 2>&1 | grep TIME
done; done

# DEV TOOL: what do half-size (head-group) and interleavable launches cost?  kbench timings of the C5 / C3 per-step blocks.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for fl in 0 1; do
  echo "== USP_LAUNCH_INTERLEAVE=$fl"
  for hq_hkv in "16 2" "8 1"; do
    set -- $hq_hkv
    USP_KBENCH_FLAGS=$fl $K fwd 1 16384 16384 $1 $2 128 1 0 0 20 | grep TIME   # C5 step 0
    USP_KBENCH_FLAGS=$fl $K fwd 1 16384 8192 $1 $2 128 0 0 0 20 | grep TIME    # C5 step <= r
    USP_KBENCH_FLAGS=$fl $K fwd 1 8192 16384 $1 $2 128 0 0 0 20 | grep TIME    # C5 step > r
    USP_KBENCH_FLAGS=$fl $K bwd 1 16384 16384 $1 $2 128 1 0 0 5 | grep TIME
    USP_KBENCH_FLAGS=$fl $K bwd 1 16384 8192 $1 $2 128 0 0 0 5 | grep TIME
  done
  for h in 8 4 2; do USP_KBENCH_FLAGS=$fl $K fwd 1 16384 16384 $h $h 128 1 0 0 30 | grep TIME; done   # C3 groups
done

#!/bin/bash
# Round 6's GPU calls, one function per call (bodies as they ran; outputs merged back under gpurun_out/r06/, the ones that are
# evidence copied to profiles/ -- profiles/r06_INDEX.md).   usage:  gpurun -- 'bash tools/r06/gpu_calls.sh run01_gqa_loop'

# the GQA loop inside the dK/dV work item (ABI v7 dkdv_heads): native suite, then heads-per-item sweeps, dK/dV launch alone
run01_gqa_loop() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
mkdir -p $R/gpurun_out/r06; cd $R
( timeout 900 $K suite bwd 2>&1 | grep -E "FAIL|SUITE|TIME  bwd" ) | tee gpurun_out/r06/01_suite.log | tail -30
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch (+ its reduce) alone
for rep in 1 2; do
for h in 1 2 4 8; do
  USP_KBENCH_BWD_HEADS=$h timeout 300 $K bwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME
done
done | tee gpurun_out/r06/01_heads_64k.log
for h in 1 2 4 8; do
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 16384 16384 8 1 128 1 0 0 10 2>&1 | grep TIME
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 2>&1 | grep TIME
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 8192 16384 8 1 128 0 0 0 10 2>&1 | grep TIME
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 32768 32768 32 4 128 1 0 0 5 2>&1 | grep TIME
done | tee gpurun_out/r06/01_heads_rank.log
unset USP_KBENCH_FLAGS
timeout 300 $K bwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME | tee gpurun_out/r06/01_bwd_64k_default.log
}

"$@"

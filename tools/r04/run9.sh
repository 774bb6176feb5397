#!/bin/bash
# round 4, GPU call 9: ablations with REAL tiles in LDS (no-DMA build keeps the prologue's loads).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in m0base m0AB6 nodma2 nodma nolds2 noexp2 nodma2_nolds; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done

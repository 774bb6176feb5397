#!/bin/bash
# round 4, GPU call 6: is the per-piece M0 write what an LDS-DMA piece costs?  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in base dmaA3 probe6 probe6A3 probe7 probe7A3 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done

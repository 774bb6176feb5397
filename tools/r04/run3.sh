#!/bin/bash
# round 4, GPU call 3: what does an LDS-DMA piece cost beside LDS fragment reads / in VALU-only gaps?  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in base dmaAB7 nolds nolds_nodma nolds_dmaA4 nolds_dmaA2 nolds_dmaAB8 nolds_dmaB3 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done

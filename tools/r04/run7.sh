#!/bin/bash
# round 4, GPU call 7: register staging with the LDS writes spread over phase B.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in stgA3B3 stgA2B3 stgA3B2 stgHB3; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 1 3000 5000 9 3 128 0 0 1 0 | cut -c1-150)"
done
for rep in 1 2 3; do
  for v in base probe7 stgA3B3 stgA2B3 stgA3B2 stgHB3 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done

#!/bin/bash
# round 4, GPU call 4: do the four lockstep waves of a workgroup queue behind each other at the CU's texture addresser?  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in stag41 stag51 stag61 stag72; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
done
for rep in 1 2 3; do
  for v in base dmaA4 stag41 stag51 stag61 stag72 nolds nolds_stag41 nolds_stag61 nolds_nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done

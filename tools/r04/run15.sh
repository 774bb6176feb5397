#!/bin/bash
# round 4, GPU call 15: dkdv64 ablations (P hand-off, exp) and element-stream windows.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
$K bwd 2 8192 8192 16 16 128 1 0 0 30 > /dev/null
for v in b_e1856 b_e1650; do echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K bwd 1 2048 2048 4 2 128 1 0 1 0 | cut -c1-150)"; done
for rep in 1 2 3; do for v in b_base b_nopx b_noexp b_e1856 b_e1650; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
done; done

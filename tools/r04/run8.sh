#!/bin/bash
# round 4, GPU call 8: one M0 write per tile (pieces addressed by the immediate offset).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in m0base m0A3; do
  for shape in "2 2048 2048 16 16 128 1 0" "1 3000 5000 9 3 128 0 0" "1 333 200 2 2 128 1 0" "1 200 333 3 1 128 1 0" "1 4096 4096 20 4 128 1 1"; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd $shape 1 0 | cut -c1-150)"
  done
done
for rep in 1 2 3; do
  for v in base m0base m0A3 m0A2 m0A4 m0AB6 m0A1 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  done
done

#!/bin/bash
# round 4, GPU call 18: dkdv64's LDS-DMA: does its cost move with the phase that issues it (latency exposed at the drain)
# or with the number of pieces (issue / LDS write bandwidth)?  Timing-only builds (b_half reads stale dO).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
run() { LD_LIBRARY_PATH=$R/abl/$1 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150; }
for rep in 1 2 3; do
  for v in b_base b_nodma b_half b_ph1 b_ph2 b_ph3; do echo "[$v] $(run $v)"; done
done

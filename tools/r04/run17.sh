#!/bin/bash
# round 4, GPU call 17: where do dkdv64's exposed waits come from?  Timing-only A/B builds that keep REAL tiles in LDS
# (results of the variants are wrong by construction; only kernel time is read).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
run() { LD_LIBRARY_PATH=$R/abl/$1 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c1-150; }
for rep in 1 2 3; do
  for v in b_base b_nodma b_prebar b_nobar b_any b_any_prebar; do echo "[$v] $(run $v)"; done
done

#!/bin/bash
# round 4, GPU call 2: fwd64 variants (DMA placement, schedule knobs, ablations).  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 1 0 0 200 > /dev/null      # warm the clocks
echo "== correctness of the placements (oracle) =="
for v in base dmaA3 dmaA2 dmaAB4 dmaB2 dmaA4 nea46 nea50 pf3 pf4 lead6 max12; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
done
echo "== timing =="
for rep in 1 2; do
  for v in base dmaA3 dmaA2 dmaAB4 dmaB2 dmaA4 nea46 nea50 pf3 pf4 lead6 max12 nodma noexp nolds nobar noexplds none; do
    for shape in "2 8192 8192 16 16 128 1" "2 8192 8192 16 16 128 0"; do
      echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd $shape 0 0 60 | grep TIME)"
    done
  done
done

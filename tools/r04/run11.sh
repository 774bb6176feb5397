#!/bin/bash
# round 4, GPU call 11: fwd64 after the register-pressure fix.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
for shape in "2 512 512 4 4 128 1 0" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "2 2048 2048 16 16 128 1 0" "1 3000 5000 9 3 128 0 0" "1 5000 3000 8 8 128 1 1"; do
  timeout 300 $K fwd $shape 1 0 | cut -c1-160 || echo "RC=$? for $shape"
done
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_FWD_WAVES=$w
  echo "[waves $w] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 50 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 30 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME)"
done; done

#!/bin/bash
# round 4, GPU call 32: whole-row-piece 16-bit epilogue stores in the two 64-row backward kernels: parity (python path, 16-bit
# outputs) and product-path timing against the previous library (swapped into the box's scratch copy).  DEV script.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu -k "64row or backward or bwd or golden or seq64k" 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
for rep in 1 2; do
  echo "[new ] $(python tools/prof_product.py c2 40 | tail -1)"
  cp long-context-attention_amd/libusp_hip.so /tmp/new.so; cp abl/prev3/libusp_hip.so long-context-attention_amd/libusp_hip.so
  echo "[prev] $(python tools/prof_product.py c2 40 | tail -1)"
  cp /tmp/new.so long-context-attention_amd/libusp_hip.so
done
echo "[new 64K] $(python tools/prof_product.py c5 3 | tail -1)"

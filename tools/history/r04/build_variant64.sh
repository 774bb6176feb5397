#!/bin/bash
# usage: tools/r04/build_variant64.sh <fwd|dkdv|dq> <name> "<extra hipcc flags>"  -> abl/<name>/libusp_hip.so
# DEV TOOL: a variant of libusp_hip.so that differs only in ONE of the three one-wave-per-SIMD kernel sources (knobs, timing
# builds -DUSP_F64_TIMING / -DUSP_B64_TIMING / -DUSP_Q64_TIMING, A/B ablations); the other objects are the in-tree ones (run
# `make -C long-context-attention_amd/csrc` first).  Use with LD_LIBRARY_PATH=abl/<name>.  (Rounds 4's three per-kernel scripts.)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C="$R/long-context-attention_amd/csrc"
case "$1" in
  fwd) SRC=usp_flash_fwd64 ;; dkdv) SRC=usp_flash_bwd64 ;; dq) SRC=usp_flash_bwd_dq64 ;;
  *) echo "usage: $0 <fwd|dkdv|dq> <name> [flags]"; exit 64 ;;
esac
mkdir -p "$R/abl/$2"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I"$R/include" $3 -c "$C/$SRC.hip" -o "$R/abl/$2/v.o"
OBJS=""
for o in usp_flash_fwd usp_flash_fwd64 usp_flash_bwd usp_flash_bwd64 usp_flash_bwd_dq64 usp_elementwise; do
  if [ "$o" = "$SRC" ]; then OBJS="$OBJS $R/abl/$2/v.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/abl/$2/libusp_hip.so" $OBJS
rm -f "$R/abl/$2/v.o"

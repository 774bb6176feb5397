#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  -> abl/<name>/libusp_hip.so
# Builds a complete variant of libusp_hip.so (ablations / A-B experiments); run the harness against
# it with LD_LIBRARY_PATH=abl/<name> (kbench uses RUNPATH, so LD_LIBRARY_PATH takes precedence).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C="$R/long-context-attention_amd/csrc"
mkdir -p "$R/abl/$1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I"$R/include" $2 -shared \
  "$C/usp_flash_fwd.hip" "$C/usp_flash_fwd64.hip" "$C/usp_flash_bwd.hip" "$C/usp_flash_bwd64.hip" "$C/usp_elementwise.hip" -o "$R/abl/$1/libusp_hip.so"

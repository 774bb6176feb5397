#!/bin/bash
# usage: tools/r04/build_b64_variant.sh <name> "<extra hipcc flags>"  -> abl/<name>/libusp_hip.so
# DEV TOOL: a variant of libusp_hip.so that differs only in usp_flash_bwd64.hip; the other objects are the in-tree ones.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C="$R/long-context-attention_amd/csrc"
mkdir -p "$R/abl/$1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I"$R/include" $2 -c "$C/usp_flash_bwd64.hip" -o "$R/abl/$1/b64.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/abl/$1/libusp_hip.so" "$C/usp_flash_fwd.o" "$C/usp_flash_fwd64.o" "$C/usp_flash_bwd.o" "$R/abl/$1/b64.o" "$C/usp_flash_bwd_dq64.o" "$C/usp_elementwise.o"
rm -f "$R/abl/$1/b64.o"

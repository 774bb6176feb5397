#!/bin/bash
# usage: tools/r05/census.sh <file.hip> <kernel-name-substring> [extra hipcc flags]
# DEV TOOL (no GPU): compile one kernel source to .s and print, per MFMA loop, the instruction count and what does not belong
# into a pipelined loop (accumulator-file copies, scratch traffic) + the spill counts of every instantiation.
R=$(cd "$(dirname "$0")/../.." && pwd)
S=/tmp/census_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I$R/include --cuda-device-only -S "$1" $3 -o $S 2>/dev/null || { echo "compile failed"; exit 1; }
python $R/tools/s_loop_mix.py $S "$2" 2>&1 | grep -A2 "^loop" | grep -v "^--" | awk '/^loop/ {printf "%s ", $0} /suspicious/ {print $0} /^    [a-z]/ && !/suspicious/ {}' | sed 's/    suspicious: Counter//' | head -8
grep "vgpr_spill_count\|private_segment_fixed_size" $S | awk '{printf "%s ", $2} END {print ""}'
rm -f $S

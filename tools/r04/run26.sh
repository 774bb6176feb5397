#!/bin/bash
# round 4, GPU call 26: per-kernel durations of the backward (kernel trace) + dq64 knob variants.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- $K bwd 2 8192 8192 16 16 128 1 0 0 20 > /tmp/rp.log 2>&1
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), '%9.1f us' % (float(r['AverageNs'])/1e3))
"
for v in q_pf2 q_pf4 q_pf6 q_y16 q_y28; do
  [ -d $R/abl/$v ] && echo "[$v] $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[base ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"

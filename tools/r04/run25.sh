#!/bin/bash
# round 4, GPU call 25: dq64 against the 8-wave dQ kernel on one box (USP_BWD_DQ_WAVES=8 forces the latter only), and the
# per-kernel durations from a kernel trace.  DEV script.
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  echo "[dq64   ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[dq 8w  ] $(USP_BWD_DQ_WAVES=8 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- $K bwd 2 8192 8192 16 16 128 1 0 0 20 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/p1/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY

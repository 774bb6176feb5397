# DEV TOOL: kernel-time breakdown of one emulated rank iteration (tools/rank_emulation.py), pipelined vs one packed exchange
export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
for mode in auto 0; do
  rocprofv3 --kernel-trace --stats -d /tmp/emu_$mode -o x -- python $R/tools/rank_emulation.py --gpus ${1:-8} --iters 5 --env USP_PIPELINE_ULYSSES=$mode > /tmp/emu_$mode.log 2>&1
  grep "per iteration" /tmp/emu_$mode.log | cut -c1-140
  python3 - <<PY
import sqlite3,glob
db=glob.glob('/tmp/emu_$mode/**/*_results.db',recursive=True)[0]
c=sqlite3.connect(db)
rows=c.execute("select name,total_calls,total_duration,average from top_kernels").fetchall()
tot=sum(r[2] for r in rows)
print("mode $mode: total kernel time %.1f ms over 8 iterations (3 warm + 5)" % (tot/1e3))
for n,calls,t,avg in rows[:14]:
    print("   %-70s calls %5d  total %9.1f us  avg %8.1f" % (n.replace('void ','')[:70], calls, t, avg))
PY
done

export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
for v in base $(ls $R/abl); do
  if [ $v = base ]; then LP=""; else LP=$R/abl/$v; fi
  LD_LIBRARY_PATH=$LP rocprofv3 --kernel-trace --stats -d /tmp/abl_$v -o x -- $R/long-context-attention_amd/kbench bwd 2 8192 8192 16 16 128 1 0 0 3 > /tmp/abl_$v.log 2>&1
  python3 - <<PY
import sqlite3,glob
db=glob.glob('/tmp/abl_$v/**/*_results.db',recursive=True)[0]
c=sqlite3.connect(db)
r={n:a for n,a in c.execute("select name,average from top_kernels")}
m1=[a for n,a in r.items() if 'bwd_kernel' in n and 'true, 1>' in n]
m0=[a for n,a in r.items() if 'bwd_kernel' in n and 'true, 0>' in n]
print("ABL %-10s dKdV %8.1f us   dQ %8.1f us" % ("$v", m1[0] if m1 else -1, m0[0] if m0 else -1))
PY
done

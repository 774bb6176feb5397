# usage: tools/abl_bwd.sh [variant ...]   (variants = directories under abl/, built by build_variant.sh;
# "base" = the in-tree library; the single-role dK/dV kernel is a variant: build_variant.sh legacy "-DUSP_BWD_LEGACY")
# Prints the rocprof average duration of every backward kernel at C2 (B2 S8192 H16 D128 bf16 causal).
export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- base
for v in "$@"; do
  LP=""; EV=""
  case $v in
    base) ;;
    *) LP=$R/abl/$v ;;
  esac
  LD_LIBRARY_PATH=$LP rocprofv3 --kernel-trace --stats -d /tmp/abl_$v -o x -- \
    $R/long-context-attention_amd/kbench bwd 2 8192 8192 16 16 128 1 0 0 3 > /tmp/abl_$v.log 2>&1
  grep -E "TF/s|FAIL" /tmp/abl_$v.log | tail -2
  python3 - <<PY
import sqlite3,glob
db=glob.glob('/tmp/abl_$v/**/*_results.db',recursive=True)[0]
c=sqlite3.connect(db)
for n,a in c.execute("select name,average from top_kernels"):
    if 'bwd' in n: print("ABL %-10s %9.1f us  %s" % ("$v", a/1000 if a>1e5 else a, n[:70]))
PY
done

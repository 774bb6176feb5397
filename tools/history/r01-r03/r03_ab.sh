# usage: bash tools/r03/r03_ab.sh variant [variant ...]   -- native suite (bwd) + rocprof kernel times at C2 for each variant under abl/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
K=./long-context-attention_amd/kbench
for v in "$@"; do
  LP=""; [ $v != base ] && LP=$R/abl/$v
  LD_LIBRARY_PATH=$LP $K suite bwd 2>&1 | grep -E "SUITE|FAIL" | head -5 | sed "s/^/[$v] /"
done
bash tools/abl_bwd.sh "$@" 2>&1 | grep -E "ABL|FAIL"

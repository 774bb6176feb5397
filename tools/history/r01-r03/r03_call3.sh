R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
K=./long-context-attention_amd/kbench
$K suite bwd 2>&1 | grep -E "SUITE|FAIL" | head -20 | tee $OUT/8_suite_cuts.log
for sh in "1 16384 16384 2 1" "1 16384 16384 2 2" "1 16384 16384 4 4" "1 16384 16384 4 1" "1 16384 16384 8 1" "1 32768 32768 4 1"; do
  for cuts in "0,0" "2,2" "4,2" "4,4" "2,1" "8,4"; do
    USP_KBENCH_BWD_SPLITS=$cuts $K bwd $sh 128 1 0 0 5 2>&1 | grep -E "TIME|FAIL|failed"
  done
done | tee $OUT/8_bwd_cuts.log

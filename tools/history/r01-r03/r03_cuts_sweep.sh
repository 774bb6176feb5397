K=./long-context-attention_amd/kbench
for sh in "1 16384 16384 2 1" "1 16384 16384 4 1"; do
  for cuts in "4,2" "3,2" "6,2" "5,2" "8,2" "4,3" "3,3" "6,3" "2,2" "3,1" "2,1"; do
    USP_KBENCH_BWD_SPLITS=$cuts $K bwd $sh 128 1 0 0 8 2>&1 | grep -E "TIME|FAIL|failed"
  done
done

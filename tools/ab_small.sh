# DEV TOOL: few-heads launches (one head spans several XCDs): tools/ab_small.sh <variant> [<variant> ...]
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
$K fwd 2 8192 8192 16 16 128 1 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2; do
  for v in "$@"; do
    for fl in 0 1; do
      for shape in "1 16384 16384 4 4 128 1" "1 16384 16384 2 2 128 1" "1 32768 32768 4 1 128 1" "1 16384 16384 8 8 128 1" "1 8192 8192 4 4 128 0"; do
        echo "$v flags=$fl: $(USP_KBENCH_FLAGS=$fl LD_LIBRARY_PATH=$R/abl/$v $K fwd $shape 0 0 50 | grep TIME)"
      done
      for shape in "1 16384 16384 4 4 128 1" "1 32768 32768 4 1 128 1"; do
        echo "$v flags=$fl: $(USP_KBENCH_FLAGS=$fl LD_LIBRARY_PATH=$R/abl/$v $K bwd $shape 0 0 10 | grep TIME)"
      done
    done
  done
done

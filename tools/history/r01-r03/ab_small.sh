# DEV TOOL: launches whose heads do not fill whole XCDs: tools/ab_small.sh <variant> [<variant> ...]
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
$K fwd 2 8192 8192 16 16 128 1 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2; do
  for v in "$@"; do
    for shape in "1 32768 32768 4 1 128 1" "1 16384 16384 6 6 128 1" "1 8192 8192 12 12 128 1" "1 8192 8192 20 4 128 1" "1 16384 16384 3 3 128 1" "1 8192 8192 16 16 128 1"; do
      echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v $K fwd $shape 0 0 50 | grep TIME)"
      echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v $K bwd $shape 0 0 10 | grep TIME)"
    done
  done
done

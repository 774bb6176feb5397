# DEV TOOL: A/B of forward-kernel variants built by tools/build_variant.sh: tools/ab_fwd.sh <variant> [<variant> ...]
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
$K fwd 2 8192 8192 16 16 128 1 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in "$@"; do
    for shape in "2 8192 8192 16 16 128 1" "2 8192 8192 16 16 128 0" "1 16384 16384 16 2 128 1" "4 4096 4096 16 16 64 1"; do
      echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v $K fwd $shape 0 0 100 | grep TIME)"
    done
  done
done

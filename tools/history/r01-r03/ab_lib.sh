# usage: tools/ab_lib.sh variantA variantB -- <kbench args>   ("base" = in-tree lib); two interleaved rounds
R=$GRAFT_REPO_ROOT; A=$1; B=$2; shift 3
for round in 1 2; do for v in $A $B; do
  LP=""; [ $v != base ] && LP=$R/abl/$v
  LD_LIBRARY_PATH=$LP $R/long-context-attention_amd/kbench "$@" 2>&1 | grep -E "^TIME" | sed "s/^/[$round $v] /"
done; done

# usage: tools/ab_fwd.sh [variant ...]  -- kbench fwd timings at C2 (causal + full) for in-tree ("base") and abl/<variant> libs,
# interleaved twice so box / clock drift shows up as a difference between the two rounds.
R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- base
for round in 1 2; do
  for v in "$@"; do
    LP=""; [ $v != base ] && LP=$R/abl/$v
    for causal in 1 0; do
      LD_LIBRARY_PATH=$LP $R/long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 $causal 0 0 50 2>&1 | grep -E "TIME|TF" | tail -1 | sed "s/^/[$round $v] /"
    done
  done
done

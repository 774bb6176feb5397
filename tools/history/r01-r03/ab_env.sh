# usage: tools/ab_env.sh VAR val1 val2 -- <kbench args...>   (runs each value twice, interleaved)
R=$GRAFT_REPO_ROOT; VAR=$1; A=$2; B=$3; shift 4
for round in 1 2; do for v in $A $B; do
  env $VAR=$v $R/long-context-attention_amd/kbench "$@" 2>&1 | grep -E "^TIME" | sed "s/^/[$round $VAR=$v] /"
done; done

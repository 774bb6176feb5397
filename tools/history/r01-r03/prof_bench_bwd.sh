export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_benchbwd; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/b -o b -- python $R/bench.py --bwd 1 --no-cpu-baseline --no-parity --steps 20 --warmup 3 > $OUT/log.txt 2>&1

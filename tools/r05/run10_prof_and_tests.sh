cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
( time bash tools/prof_round.sh r05 ) > gpurun_out/r05/10_prof_round.log 2>&1
tail -5 gpurun_out/r05/10_prof_round.log
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > gpurun_out/r05/10_pytest_gpu.log 2>&1
tail -12 gpurun_out/r05/10_pytest_gpu.log

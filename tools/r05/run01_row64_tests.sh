cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time timeout 1500 python -m pytest tests/test_gpu_row64.py tests/test_gpu_mutation.py -q -x 2>&1 | tail -40 ) > gpurun_out/r05/01_row64.log 2>&1
tail -30 gpurun_out/r05/01_row64.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05/02_bench.log 2>&1
grep -E "^\{" gpurun_out/r05/02_bench.log > gpurun_out/r05/02_bench_line.json
tail -5 gpurun_out/r05/02_bench.log | cut -c1-3000
( time timeout 900 python -m pytest tests/test_gpu_rccl_order.py -q -x -k "bench_" 2>&1 | tail -15 ) > gpurun_out/r05/02_grid.log 2>&1
tail -12 gpurun_out/r05/02_grid.log

# round 5: (1) MFMA-only ceiling + clock / power telemetry beside the product kernels, one box, one process;
#          (2) one rank of the bench workloads of N = 2, 4, 8 with the wire replaced by local copies (compute-only ceilings)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 300 python tools/r05/ceiling.py > gpurun_out/r05/06_ceiling.txt 2>&1
tail -30 gpurun_out/r05/06_ceiling.txt
for n in 8 4 2; do
  timeout 200 python tools/rank_emulation.py --gpus $n --iters 4 2>&1 | grep -A1 "^configs" 
  timeout 200 python tools/rank_emulation.py --gpus $n --iters 4 --env USP_PIPELINE_ULYSSES=0 2>&1 | grep -A1 "^configs" | tail -1
done > gpurun_out/r05/06_rank_emulation.txt 2>&1
cat gpurun_out/r05/06_rank_emulation.txt

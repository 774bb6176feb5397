# Development smoke of bench.py's N > 1 path on a 1-GPU box (gloo transport, all ranks on cuda:0).
R=$GRAFT_REPO_ROOT; cd $R
for n in "$@"; do
  USP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port $((29800 + n)) bench.py --gpus $n --steps 2 --warmup 1 2>&1 \
    | grep -v "socket.cpp\|amdgpu.ids\|Gloo\|OMP_NUM_THREADS\|\*\*\*\*" | tail -6
  echo "== N=$n rc=$?"
done

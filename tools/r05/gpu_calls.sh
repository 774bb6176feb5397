#!/bin/bash
# Round 5's GPU calls, one function per call (bodies as they ran; outputs merged back under gpurun_out/r05/, the ones that are
# evidence copied to profiles/ -- profiles/r05_INDEX.md).   usage:  gpurun -- 'bash tools/r05/gpu_calls.sh run10_prof_and_tests'

run01_row64_tests() {
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time timeout 1500 python -m pytest tests/test_gpu_row64.py tests/test_gpu_mutation.py -q -x 2>&1 | tail -40 ) > gpurun_out/r05/01_row64.log 2>&1
tail -30 gpurun_out/r05/01_row64.log
}

run02_bench() {
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05/02_bench.log 2>&1
grep -E "^\{" gpurun_out/r05/02_bench.log > gpurun_out/r05/02_bench_line.json
tail -5 gpurun_out/r05/02_bench.log | cut -c1-3000
( time timeout 900 python -m pytest tests/test_gpu_rccl_order.py -q -x -k "bench_" 2>&1 | tail -15 ) > gpurun_out/r05/02_grid.log 2>&1
tail -12 gpurun_out/r05/02_grid.log
}

# round 5: per-tile anatomy of the dK/dV kernel's two roles (s_memtime build abl/b_tm), GQA rank-block shape
run04_dkdv_anatomy() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
mkdir -p $R/gpurun_out/r05
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TM" | awk '{k=$3" "$5; if (c[k]++ < 2) print}' | sort -k3n -k5n | head -40
}

# round 5: per-item anatomy of the forward and the dQ kernel (s_memtime builds abl/f_tm, abl/q_tm) at the 64K rank-block shape
run05_fwd_dq_anatomy() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
LD_LIBRARY_PATH=$R/abl/f_tm timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TF" | awk '{k=$3" "$7; if (c[k]++ < 1) print}' | sort -k3n -k7n | head -24
export USP_KBENCH_FLAGS=32        # USP_BWD_SKIP_DKDV: the dQ launch alone
LD_LIBRARY_PATH=$R/abl/q_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TQ" | awk '{k=$3" "$7; if (c[k]++ < 1) print}' | sort -k3n -k7n | head -24
timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 5 2>&1 | grep TIME
unset USP_KBENCH_FLAGS
timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 5 2>&1 | grep TIME
}

# round 5: (1) MFMA-only ceiling + clock / power telemetry beside the product kernels, one box, one process;
#          (2) one rank of the bench workloads of N = 2, 4, 8 with the wire replaced by local copies (compute-only ceilings)
run06_ceiling_emulation() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 300 python tools/r05/ceiling.py > gpurun_out/r05/06_ceiling.txt 2>&1
tail -30 gpurun_out/r05/06_ceiling.txt
for n in 8 4 2; do
  timeout 200 python tools/rank_emulation.py --gpus $n --iters 4 2>&1 | grep -A1 "^configs" 
  timeout 200 python tools/rank_emulation.py --gpus $n --iters 4 --env USP_PIPELINE_ULYSSES=0 2>&1 | grep -A1 "^configs" | tail -1
done > gpurun_out/r05/06_rank_emulation.txt 2>&1
cat gpurun_out/r05/06_rank_emulation.txt
}

# round 5: dK/dV kernel with the statistics two tiles ahead (store at the head of the stream) + fixed-register resident fragments
run07_dkdv_early_stats() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 600 $K suite bwd 2>&1 | grep -E "FAIL|SUITE|TIME  bwd"
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
for i in 1 2; do
timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 2>&1 | grep TIME
timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 2>&1 | grep TIME
done
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TM" | awk '{k=$3" "$5; if (c[k]++ < 1) print}' | sort -k3n -k5n | head -12
}

# round 5: dK/dV kernel variants -- correctness (native suite, NaN tails) + timing at C2, the 8-GPU rank block and the N = 1 workload
run08_dkdv() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 600 $K suite bwd 2>&1 | grep -E "FAIL|SUITE"
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
for i in 1 2; do
timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 2>&1 | grep TIME
timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 2>&1 | grep TIME
timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 2>&1 | grep TIME
done
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 1 2>&1 | grep "^TM" | awk '{k=$3" "$5; if (c[k]++ < 1) print}' | sort -k3n -k5n | head -8
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TI" | awk '{k=$3" "$7; if (c[k]++ < 1) print}' | sort -k3n -k7n | head -8
}

# round 5: same-box A/B of the dK/dV kernel: abl/b_old (round 4's) against the in-tree library, alternating
run09_dkdv_ab() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch alone
for i in 1 2 3; do
for lib in $R/abl/b_old $R/long-context-attention_amd; do
echo "== $(basename $lib)"
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 30 2>&1 | grep TIME
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 15 2>&1 | grep TIME
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 2>&1 | grep TIME
done; done
}

run10_prof_and_tests() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
( time bash tools/prof_round.sh r05 ) > gpurun_out/r05/10_prof_round.log 2>&1
tail -5 gpurun_out/r05/10_prof_round.log
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > gpurun_out/r05/10_pytest_gpu.log 2>&1
tail -12 gpurun_out/r05/10_pytest_gpu.log
}

# round 5: the dQ kernel's key cuts (ksplit > 1 served by flash_bwd_dq64_kernel): native suite (every backward shape x five cut pairs,
# NaN tails), few-head timing with and without cuts, the new pytest cases
run11_dq64_cuts() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
cd $R; mkdir -p gpurun_out/r05
timeout 900 $K suite bwd 2>&1 | grep -E "FAIL|SUITE|TIME  bwd"
for cuts in "0,0" "4,2" "2,1"; do
  echo "== cuts $cuts"
  USP_KBENCH_BWD_SPLITS=$cuts timeout 120 $K bwd 1 16384 16384 2 1 128 1 0 0 5 2>&1 | grep TIME
  USP_KBENCH_BWD_SPLITS=$cuts timeout 120 $K bwd 1 16384 16384 4 4 128 1 0 0 5 2>&1 | grep TIME
done
timeout 900 python -m pytest tests/test_gpu_row64.py -q -x -k "cuts or refuses or dispatch" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "cuts" 2>&1 | tail -3
}

# round 5: the dQ kernel with the barrier between the dP and the dQ blocks + a K ring of three: native suite, then a same-box A/B
# of the dQ launch alone against abl/q_base (the kernel without it), alternating
run12_dq64_midbarrier() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 900 $K suite bwd 2>&1 | grep -E "FAIL|SUITE"
export USP_KBENCH_FLAGS=32        # USP_BWD_SKIP_DKDV: the dQ launch alone
for i in 1 2 3; do
for lib in $R/abl/q_base $R/long-context-attention_amd; do
echo "== $(basename $lib)"
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 30 2>&1 | grep TIME
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 15 2>&1 | grep TIME
LD_LIBRARY_PATH=$lib timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 2>&1 | grep TIME
done; done
}

# round 5: the self-chunk start of the first head group on the 2-GPU grid: two processes sharing the GPU (HIP kernels), the N = 2
# bench workload at full size on the virtual grid through real RCCL, and the compute-only cost of the split (rank emulation)
run13_self_chunk() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_multiproc.py -q -x -k "self_chunk" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_rccl_order.py -q -x -k "bench_2gpu" -rP 2>&1 | grep -E "passed|failed|max abs errors|Error" | tail -6
for env in USP_SELF_CHUNK=0 USP_SELF_CHUNK=1; do
  timeout 200 python tools/rank_emulation.py --gpus 2 --iters 4 --env $env 2>&1 | grep -A1 "^configs" | tail -1
done
}

# round 5, final tree: the whole GPU suite, larger seeded sweeps (forced 4x64 forward 600 seeds, dense 300, packed 100, backward
# family 200), and the driver's N = 1 command twice
run14_final() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r05/14_pytest_gpu.log 2>&1
tail -6 gpurun_out/r05/14_pytest_gpu.log
( time USP_FUZZ_ROW64_FWD=600 USP_FUZZ_DENSE=300 USP_FUZZ_PACKED=100 USP_FUZZ_ROW64=200 timeout 2400 python -m pytest tests/test_gpu_row64.py tests/test_gpu_fuzz.py -q -x -n 4 2>&1 | tail -6 ) > gpurun_out/r05/14_fuzz.log 2>&1
tail -5 gpurun_out/r05/14_fuzz.log
for i in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05/14_bench_$i.err | grep -E "^\{" > gpurun_out/r05/14_bench_$i.json
  python -c "import json; d=json.load(open('gpurun_out/r05/14_bench_$i.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['step']['fwd_ms'], r['step']['dkdv_ms'], r['step']['dq_ms'], r['layer_step_ms_by_kernel_family'], r['mfma_ceiling']['sustained_ceiling_TFLOPs'], (r['traffic'] or {}).get('read_MB'))"
done
}

# round 5: where one rank's compute-only iteration of the 8-GPU config goes (kernel trace of tools/rank_emulation.py)
run15_rank_trace() {
export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
for mode in auto 0; do
  rocprofv3 --kernel-trace --stats -d /tmp/emu_$mode -o x -- python $R/tools/rank_emulation.py --gpus 8 --iters 5 --env USP_PIPELINE_ULYSSES=$mode > /tmp/emu_$mode.log 2>&1
  grep "per iteration" /tmp/emu_$mode.log | cut -c1-140
  python3 - <<PY
import sqlite3,glob
db=glob.glob('/tmp/emu_$mode/**/*_results.db',recursive=True)[0]
c=sqlite3.connect(db)
rows=c.execute("select name,total_calls,total_duration,average from top_kernels").fetchall()
tot=sum(r[2] for r in rows)
print("mode $mode: total kernel time %.1f ms over 8 iterations (3 warm + 5) = %.2f ms per iteration" % (tot/1e3, tot/8e3))
for n,calls,t,avg in rows[:16]:
    print("   %-70s calls %5d  total %9.1f us  avg %8.1f  (%.2f ms / iteration)" % (n.replace('void ','')[:70], calls, t, avg, t/8e3))
PY
done
}

# round 5: 256-item interleavable forward launches (the q[c:] half-row steps of a head group at the 8-GPU config) beside resident
# copy workgroups: the 4-wave 128-row kernel (round 2's heuristic, taken against the 8-wave kernel) against the 4 x 64 kernel
run16_fwd_small_interleave() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for shape in "8192 16384 8 1" "8192 49152 8 1" "16384 8192 8 1"; do
  for w in 4 64 8; do
    echo "== shape (Sq Sk Hq Hkv) $shape   USP_FWD_WAVES=$w"
    USP_OVL_SHAPE="$shape" USP_FWD_WAVES=$w timeout 120 $K overlap 4 16 8 2>&1 | grep "OVERLAP USP_LAUNCH"
  done
done
}

# round 5: the K split through the 4 x 64 forward (split instantiation of flash_fwd64_kernel + split_merge_kernel): parity first,
# then the same few-head causal launches timed with the 8-wave split kernel (USP_FWD_WAVES=8) and with the library's choice
run17_fwd64_ksplit() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05; K=$GRAFT_REPO_ROOT/long-context-attention_amd/kbench
( time USP_FUZZ_ROW64_FWD=300 timeout 1500 python -m pytest tests/test_gpu_row64.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_mutation.py -q -x -n 4 2>&1 | tail -8 ) 2>&1 | tail -12
timeout 300 $K suite 2>&1 | grep -c "ok$\|OK" ; timeout 300 $K suite 2>&1 | grep -i "fail\|bad [1-9]" | head
for shape in "1 16384 16384 2 2" "1 16384 16384 4 4" "1 16384 16384 4 1" "1 32768 32768 2 2" "1 8192 8192 6 6" "1 65536 65536 1 1"; do
  for w in 8 64 0; do
    echo "== B Sq Sk Hq Hkv $shape  USP_FWD_WAVES=$w (8: the 8-wave split kernel; 64: the 4 x 64 one wherever n x items >= 256; 0: the library's choice)"
    USP_FWD_WAVES=$w timeout 200 $K ksplit $shape 128 1 0 0 10 2>&1 | grep "^TIME"
  done
done
}

# round 5: the self-chunk start beside the zigzag ring (ulysses 2 x ring 4): the full-size virtual grid through RCCL with the split
# spied on, then what the split costs in compute on one rank (tools/rank_emulation.py, wire = local copies)
run18_self_chunk_ring() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
( time timeout 1500 python -m pytest tests/test_gpu_rccl_order.py -q -x -k "self_chunk or configs4_8gpu_u2r4" 2>&1 | tail -5 ) 2>&1 | tail -9
for e in USP_SELF_CHUNK=0 USP_SELF_CHUNK=1 USP_SELF_CHUNK=0 USP_SELF_CHUNK=1; do
  timeout 600 python tools/rank_emulation.py --gpus 8 --iters 10 --env $e 2>&1 | grep "per iteration" | cut -c1-200
done
for r in 1 4 5; do
  timeout 600 python tools/rank_emulation.py --gpus 8 --iters 10 --rank $r --env USP_SELF_CHUNK=1 2>&1 | grep "per iteration" | cut -c1-200
done
}

# round 5: why a middle ring rank of the 8-GPU grid takes longer than ring rank 0 (kernel traces of tools/rank_emulation.py;
# 3 warm + 5 timed iterations traced)
run19_rank_compare() {
export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
for rank in 0 4 6; do
  rocprofv3 --kernel-trace --stats -d /tmp/emu_r$rank -o x -- python $R/tools/rank_emulation.py --gpus 8 --iters 5 --rank $rank > /tmp/emu_r$rank.log 2>&1
  grep "per iteration" /tmp/emu_r$rank.log | cut -c1-140
  python3 $R/tools/r05/trace_top.py /tmp/emu_r$rank 8 "rank $rank" 14
done
}

case "$1" in
  run01_row64_tests|run02_bench|run04_dkdv_anatomy|run05_fwd_dq_anatomy|run06_ceiling_emulation|run07_dkdv_early_stats|run08_dkdv|run09_dkdv_ab|run10_prof_and_tests|run11_dq64_cuts|run12_dq64_midbarrier|run13_self_chunk|run14_final|run15_rank_trace|run16_fwd_small_interleave|run17_fwd64_ksplit|run18_self_chunk_ring|run19_rank_compare) "$1" ;;
  *) echo "usage: $0 {run01_row64_tests|run02_bench|run04_dkdv_anatomy|run05_fwd_dq_anatomy|run06_ceiling_emulation|run07_dkdv_early_stats|run08_dkdv|run09_dkdv_ab|run10_prof_and_tests|run11_dq64_cuts|run12_dq64_midbarrier|run13_self_chunk|run14_final|run15_rank_trace|run16_fwd_small_interleave|run17_fwd64_ksplit|run18_self_chunk_ring|run19_rank_compare}"; exit 64 ;;
esac

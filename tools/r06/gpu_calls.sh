#!/bin/bash
# Round 6's GPU calls, one function per call (bodies as they ran; outputs merged back under gpurun_out/r06/, the ones that are
# evidence copied to profiles/ -- profiles/r06_INDEX.md).   usage:  gpurun -- 'bash tools/r06/gpu_calls.sh run01_gqa_loop'

# the GQA loop inside the dK/dV work item (ABI v7 dkdv_heads): native suite, then heads-per-item sweeps, dK/dV launch alone
run01_gqa_loop() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
mkdir -p $R/gpurun_out/r06; cd $R
( timeout 900 $K suite bwd 2>&1 | grep -E "FAIL|SUITE|TIME  bwd" ) | tee gpurun_out/r06/01_suite.log | tail -30
export USP_KBENCH_FLAGS=16        # USP_BWD_SKIP_DQ: the dK/dV launch (+ its reduce) alone
for rep in 1 2; do
for h in 1 2 4 8; do
  USP_KBENCH_BWD_HEADS=$h timeout 300 $K bwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME
done
done | tee gpurun_out/r06/01_heads_64k.log
for h in 1 2 4 8; do
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 16384 16384 8 1 128 1 0 0 10 2>&1 | grep TIME
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 2>&1 | grep TIME
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 8192 16384 8 1 128 0 0 0 10 2>&1 | grep TIME
  USP_KBENCH_BWD_HEADS=$h timeout 120 $K bwd 1 32768 32768 32 4 128 1 0 0 5 2>&1 | grep TIME
done | tee gpurun_out/r06/01_heads_rank.log
unset USP_KBENCH_FLAGS
timeout 300 $K bwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME | tee gpurun_out/r06/01_bwd_64k_default.log
}

# the driver's command on the tree with the GQA loop + the whole GPU suite with per-file durations
run02_bench_tests() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06/02_bench.log 2>&1
grep -E "^\{" gpurun_out/r06/02_bench.log > gpurun_out/r06/02_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/02_bench_line.json"))
r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d.get("frac_of_mfma_roofline"))
print("step", {k: r["step"][k] for k in ("fwd_ms", "delta_ms", "dkdv_ms", "dq_ms", "bwd_frac", "frac")})
print("kinds", r["kernels_launched"], "ceiling", r["mfma_ceiling"]["sustained_ceiling_TFLOPs"])
print("family", r["layer_step_ms_by_kernel_family"]["row64"], r["layer_step_ms_by_kernel_family"]["wave32"], "c2", r["c2"]["bwd_ms"], r["c2"]["fwd_kernel_ms"])
print("parity", r["sampled_parity"]["max_err_over_tolerance"])
PY
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -60 ) > gpurun_out/r06/02_pytest_gpu.log 2>&1
tail -45 gpurun_out/r06/02_pytest_gpu.log
}

# tails + self-chunk defaults on the GPU: virtual-grid RCCL tests, multi-process tests; then one rank of the 8-GPU grid on one GPU
# (wire = local copies) with the round-5 schedule, with the self-chunk start only, and with the defaults (self-chunk + tails)
run03_tails() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests/test_gpu_rccl_order.py tests/test_gpu_multiproc.py -q -x -rP 2>&1 | grep -E "virtual grid at full size|passed|failed|Error|error|seconds per|s  tests" | tail -40 ) > gpurun_out/r06/03_tails_tests.log 2>&1
tail -30 gpurun_out/r06/03_tails_tests.log
for rank in 0 5; do
for rep in 1 2; do
  for e in "USP_SELF_CHUNK=0 USP_TAILS=0" "USP_SELF_CHUNK=1 USP_TAILS=0" "USP_SELF_CHUNK=1 USP_TAILS=4" "USP_SELF_CHUNK=1 USP_TAILS=2"; do
    echo "== rank $rank  $e"
    env $e timeout 200 python tools/rank_emulation.py --gpus 8 --rank $rank --iters 6 2>&1 | grep -A1 "^configs" | tail -1
  done
done
done > gpurun_out/r06/03_rank_emulation.txt 2>&1
cat gpurun_out/r06/03_rank_emulation.txt
}

# the schedule variants on one rank of the 8-GPU grid (compute-only), the launch list of the default, and the bench order A/B
run04_rank_and_order() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT
for rank in 0 5; do
for rep in 1 2; do
  for e in "USP_SELF_CHUNK=0 USP_TAILS=0" "USP_SELF_CHUNK=1 USP_TAILS=0" "USP_SELF_CHUNK=1 USP_TAILS=4" "USP_SELF_CHUNK=1 USP_TAILS=2" "USP_SELF_CHUNK=all USP_TAILS=4"; do
    echo "== rank $rank  $e"
    env $e timeout 200 python tools/rank_emulation.py --gpus 8 --rank $rank --iters 6 2>&1 | grep -A1 "^configs" | tail -1 | cut -c1-150
  done
done
done > gpurun_out/r06/04_rank_emulation.txt 2>&1
cat gpurun_out/r06/04_rank_emulation.txt
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/emu_r6 -o x -- python $R/tools/rank_emulation.py --gpus 8 --rank 0 --iters 3 > /tmp/emu_r6.log 2>&1; python $R/tools/r06/rank_launches.py /tmp/emu_r6 6 ) > gpurun_out/r06/04_rank_launches.txt 2>&1
tail -60 gpurun_out/r06/04_rank_launches.txt
for rep in 1 2; do
for order in headline_first kernels_first; do
  USP_BENCH_ORDER=$order python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -E "^\{" > gpurun_out/r06/04_bench_${order}_$rep.json
  python - gpurun_out/r06/04_bench_${order}_$rep.json $order <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms_per_step", d["ms_per_step"], "frac", d["frac_of_mfma_roofline"], "| kernels alone:", r["step"]["fwd_ms"], r["step"]["dkdv_ms"], r["step"]["dq_ms"],
      "| 2-step family row64", r["layer_step_ms_by_kernel_family"]["row64"], "| ceiling", r["mfma_ceiling"]["sustained_ceiling_TFLOPs"], "| kinds", r["kernels_launched"]["bwd"])
PY
done
done | tee gpurun_out/r06/04_bench_order.txt
}

# the query heads of a KV group side by side in the item walk (usp_group_item): time and HBM fetch of the forward and the dQ kernel
# at the N = 1 workload's shape, against the head-major walk (USP_ITEM_GROUP=0); native harness, alternating
run05_item_group() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; mkdir -p $R/gpurun_out/r06; cd $R
( timeout 900 $K suite bwd 2>&1 | grep -E "FAIL|SUITE" ) | tail -3
for rep in 1 2 3; do
for g in 1 0; do
  echo "USP_ITEM_GROUP=$g fwd: $(USP_ITEM_GROUP=$g timeout 200 $K fwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME)"
  echo "USP_ITEM_GROUP=$g dq : $(USP_ITEM_GROUP=$g USP_KBENCH_FLAGS=32 timeout 200 $K bwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME)"
done
done | tee gpurun_out/r06/05_item_group_time.txt
for g in 1 0; do
  echo "USP_ITEM_GROUP=$g fwd C2: $(USP_ITEM_GROUP=$g timeout 200 $K fwd 2 8192 8192 16 16 128 1 0 0 20 2>&1 | grep TIME)"
  echo "USP_ITEM_GROUP=$g fwd 16K GQA 16/2: $(USP_ITEM_GROUP=$g timeout 200 $K fwd 1 16384 16384 16 2 128 1 0 0 10 2>&1 | grep TIME)"
done | tee -a gpurun_out/r06/05_item_group_time.txt
export TMPDIR=/tmp; cd /tmp
for g in 1 0; do
  USP_ITEM_GROUP=$g rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f$g -o x -- $K fwd 1 65536 65536 32 4 128 1 0 0 1 > /dev/null 2>&1
  USP_ITEM_GROUP=$g USP_KBENCH_FLAGS=32 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_q$g -o x -- $K bwd 1 65536 65536 32 4 128 1 0 0 1 > /dev/null 2>&1
  for d in f q; do
  python3 - /tmp/pmc_$d$g $g <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name = 'FETCH_SIZE' group by name").fetchall()
for name, v, n in rows:
    if "flash_" in name:
        print(f"USP_ITEM_GROUP={sys.argv[2]}  {name.split('(')[0][:60]:60s} dispatches {n}  FETCH_SIZE {v / n:14.1f} KiB/launch -> HBM read {v / n * 2 * 1024 / 1e6:9.1f} MB (x2 gfx950 correction)")
PY
  done
done 2>&1 | tee $R/gpurun_out/r06/05_item_group_fetch.txt
}

# the whole GPU suite as the driver runs it (serial, -x), with per-file durations; then smoke
run06_suite() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > gpurun_out/r06/06_pytest_gpu.log 2>&1
tail -32 gpurun_out/r06/06_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
}

# the round's rocprofv3 record of the product path (kernel trace + PMC passes: tools/prof_round.sh) -> profiles/r06_rocprof_summary.txt
run07_prof() {
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r06 2>&1 | tail -45
cp gpurun_out/prof_r06/bench_line.json gpurun_out/prof_r06/bench_under_rocprof.json 2>/dev/null
}

# dK/dV launch: HBM fetch and time by query heads per work item (the in-item head loop lets the 32 concurrent workgroups of an XCD drift
# apart head by head: does their window still fit the L2?)
run08_gsub_fetch() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; mkdir -p $R/gpurun_out/r06; cd $R
export USP_KBENCH_FLAGS=16 TMPDIR=/tmp
for rep in 1 2; do for h in 1 2 4; do
  echo "heads/item $h: $(USP_KBENCH_BWD_HEADS=$h timeout 300 $K bwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME | cut -c1-110)"
done; done | tee gpurun_out/r06/08_gsub_time.txt
cd /tmp
for h in 1 2 4; do
  USP_KBENCH_BWD_HEADS=$h rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_g$h -o x -- $K bwd 1 65536 65536 32 4 128 1 0 0 1 > /dev/null 2>&1
  python3 - /tmp/pmc_g$h $h <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name = 'FETCH_SIZE' group by name").fetchall()
for name, v, n in rows:
    if "dkdv" in name or "reduce" in name:
        print(f"heads/item {sys.argv[2]}  {name.split('(')[0][:50]:50s} dispatches {n}  HBM read {v / n * 2 * 1024 / 1e6:9.1f} MB per launch (FETCH_SIZE x2)")
PY
done 2>&1 | tee $R/gpurun_out/r06/08_gsub_fetch.txt
}

# final schedule code: the RCCL virtual-grid tests again, then compute-only iterations of one rank of every grid with and without the
# round-6 schedule pieces
run09_schedules() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
( time timeout 1200 python -m pytest tests/test_gpu_rccl_order.py -q -x 2>&1 | tail -12 ) > gpurun_out/r06/09_rccl_order.log 2>&1
tail -10 gpurun_out/r06/09_rccl_order.log
for rep in 1 2; do
  for e in "USP_SELF_CHUNK=0 USP_TAILS=0" "USP_SELF_CHUNK=1 USP_TAILS=0" "USP_SELF_CHUNK=1 USP_TAILS=4" "USP_SELF_CHUNK=1 USP_TAILS=2"; do
    echo "== N=2 rank 0  $e"; env $e timeout 300 python tools/rank_emulation.py --gpus 2 --rank 0 --iters 3 2>&1 | grep -A1 "^configs" | tail -1 | cut -c1-150
  done
  for e in "USP_BWD_SPLIT_STEPS=0" "USP_BWD_SPLIT_STEPS=1"; do
    echo "== N=4 rank 0  $e"; env $e timeout 300 python tools/rank_emulation.py --gpus 4 --rank 0 --iters 4 2>&1 | grep -A1 "^configs" | tail -1 | cut -c1-150
    echo "== N=4 rank 2  $e"; env $e timeout 300 python tools/rank_emulation.py --gpus 4 --rank 2 --iters 4 2>&1 | grep -A1 "^configs" | tail -1 | cut -c1-150
  done
  for e in "USP_SELF_CHUNK=0 USP_TAILS=0 USP_BWD_SPLIT_STEPS=0" "USP_SELF_CHUNK=1 USP_TAILS=4"; do
    echo "== N=8 rank 0  $e"; env $e timeout 300 python tools/rank_emulation.py --gpus 8 --rank 0 --iters 6 2>&1 | grep -A1 "^configs" | tail -1 | cut -c1-150
    echo "== N=8 rank 5  $e"; env $e timeout 300 python tools/rank_emulation.py --gpus 8 --rank 5 --iters 6 2>&1 | grep -A1 "^configs" | tail -1 | cut -c1-150
  done
done > gpurun_out/r06/09_rank_emulation_all.txt 2>&1
cat gpurun_out/r06/09_rank_emulation_all.txt
}

# the final tree: the driver's command twice, the GPU suite as the driver runs it, smoke, then the seeded sweeps at 10x their default size
run10_final() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for i in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep -E "^\{" > gpurun_out/r06/10_bench_driver_cmd_$i.json
  python - gpurun_out/r06/10_bench_driver_cmd_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["frac_of_mfma_roofline"], "| kernels alone:", r["step"]["fwd_ms"], r["step"]["dkdv_ms"], r["step"]["dq_ms"],
      "| bwd_frac", r["step"]["bwd_frac"], "| roofline.frac", r["frac"], "| ceiling", r["mfma_ceiling"]["sustained_ceiling_TFLOPs"], "| traffic", (r.get("traffic") or {}).get("read_MB"),
      "| kinds", r["kernels_launched"], "| c2", r["c2"]["fwd_kernel_ms"], r["c2"]["bwd_ms"], "| parity", r["sampled_parity"]["max_err_over_tolerance"])
PY
done
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -24 ) > gpurun_out/r06/10_pytest_gpu.log 2>&1
tail -22 gpurun_out/r06/10_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time USP_FUZZ_ROW64_FWD=400 USP_FUZZ_DENSE=300 USP_FUZZ_PACKED=100 USP_FUZZ_ROW64=200 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_row64.py -q -x -k "fuzz or sweep or seed" 2>&1 | tail -12 ) > gpurun_out/r06/10_fuzz.log 2>&1
tail -10 gpurun_out/r06/10_fuzz.log
}

# development smoke of bench.py --gpus 8 and --gpus 2 (N processes sharing ONE GPU over gloo: the staged modes, the parity of every
# rank's shard, the overlap probe, with the real kernels) -- NOT a measurement; then the driver's command once more on this box
run11_smoke_multi() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
( time bash tools/smoke_multi.sh 8 ) > gpurun_out/r06/11_smoke_n8.log 2>&1; tail -4 gpurun_out/r06/11_smoke_n8.log | cut -c1-1500
( time bash tools/smoke_multi.sh 2 ) > gpurun_out/r06/11_smoke_n2.log 2>&1; tail -4 gpurun_out/r06/11_smoke_n2.log | cut -c1-1500
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep -E "^\{" > gpurun_out/r06/11_bench_driver_cmd_3.json
python -c "
import json; d=json.load(open('gpurun_out/r06/11_bench_driver_cmd_3.json')); print('ms_per_step', d['ms_per_step'], 'frac', d['frac_of_mfma_roofline'], 'ceiling', d['roofline']['mfma_ceiling']['sustained_ceiling_TFLOPs'])"
}

# launch lists of ring rank 0 and ring rank 2 of the 8-GPU grid (default schedule), same box, back to back
run12_rank_compare() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rank in 0 5 0 5; do
  ( cd /tmp && rm -rf /tmp/emu_rk && rocprofv3 --kernel-trace -d /tmp/emu_rk -o x -- python $R/tools/rank_emulation.py --gpus 8 --rank $rank --iters 3 > /tmp/emu_rk.log 2>&1; echo "######## rank $rank: $(grep 'per iteration' /tmp/emu_rk.log | cut -c1-110)"; python $R/tools/r06/rank_launches.py /tmp/emu_rk 6 )
done > gpurun_out/r06/12_rank_compare.txt 2>&1
grep -c . gpurun_out/r06/12_rank_compare.txt
}

# the two non-causal ring-step shapes of the 8-GPU grid in isolation: q[c:] x all keys (ring ranks' steps s > r) against all rows x front keys
# (steps s <= r), dK/dV and dQ launches alone, by heads per item
run13_step_shapes() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; cd $R; mkdir -p gpurun_out/r06
for rep in 1 2; do
for shape in "8192 16384" "16384 8192"; do
  for h in 0 1 2 4; do
    echo "dkdv rows x keys $shape heads/item $h: $(USP_KBENCH_FLAGS=16 USP_KBENCH_BWD_HEADS=$h timeout 100 $K bwd 1 $shape 8 1 128 0 0 0 20 2>&1 | grep TIME | awk '{print $(NF-5), $(NF-4)}')"
  done
  echo "dq   rows x keys $shape: $(USP_KBENCH_FLAGS=32 timeout 100 $K bwd 1 $shape 8 1 128 0 0 0 20 2>&1 | grep TIME | awk '{print $(NF-5), $(NF-4)}')"
  echo "fwd  rows x keys $shape: $(timeout 100 $K fwd 1 $shape 8 1 128 0 0 0 20 2>&1 | grep TIME | awk '{print $(NF-5), $(NF-4)}')"
done
done | tee gpurun_out/r06/13_step_shapes.txt
}

# the driver's command alone (every gpurun call lands on another box of the pool: the spread of the headline)
run14_bench_only() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep -E "^\{" > gpurun_out/r06/14_bench_$1.json
python -c "
import json; d=json.load(open('gpurun_out/r06/14_bench_$1.json')); r=d['roofline']; print('ms_per_step', d['ms_per_step'], 'frac', d['frac_of_mfma_roofline'], 'ceiling', r['mfma_ceiling']['sustained_ceiling_TFLOPs'], 'kernels', r['step']['fwd_ms'], r['step']['dkdv_ms'], r['step']['dq_ms'])"
}

run15_tails_gpu_test() {
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
( time timeout 900 python -m pytest tests/test_gpu_multiproc.py -q -x -k "tails or self_chunk" 2>&1 | tail -12 ) 2>&1 | tee gpurun_out/r06/15_tails_gpu.log | tail -14
}

# (NOT adopted: -0.45 % at the metric's shape, +1.2 % at C2, -0.7 % on a 16K full launch -- profiles/r06_fwd_lds0_experiment.txt)
# forward 4 x 64: LDS addresses from the integer 0 instead of the symbol (12 v_add of 0 less per two tiles) against the library before
# (abl/fwd_old), alternating; native suite first
run16_fwd_lds0() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; cd $R; mkdir -p gpurun_out/r06
( timeout 900 $K suite 2>&1 | grep -E "FAIL|SUITE" ) | tail -3
for rep in 1 2 3 4; do
  echo "new 64K: $(timeout 200 $K fwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME | awk '{print $(NF-6)}')  old 64K: $(LD_LIBRARY_PATH=$R/abl/fwd_old timeout 200 $K fwd 1 65536 65536 32 4 128 1 0 0 3 2>&1 | grep TIME | awk '{print $(NF-6)}')"
  echo "new C2 : $(timeout 200 $K fwd 2 8192 8192 16 16 128 1 0 0 30 2>&1 | grep TIME | awk '{print $(NF-6)}')  old C2 : $(LD_LIBRARY_PATH=$R/abl/fwd_old timeout 200 $K fwd 2 8192 8192 16 16 128 1 0 0 30 2>&1 | grep TIME | awk '{print $(NF-6)}')"
  echo "new 16K full 8/1: $(timeout 200 $K fwd 1 8192 16384 8 1 128 0 0 0 20 2>&1 | grep TIME | awk '{print $(NF-6)}')  old: $(LD_LIBRARY_PATH=$R/abl/fwd_old timeout 200 $K fwd 1 8192 16384 8 1 128 0 0 0 20 2>&1 | grep TIME | awk '{print $(NF-6)}')"
done | tee gpurun_out/r06/16_fwd_lds0.txt
}

# what D = 64 launches run at (8-wave family; the reference's test default is D = 64, BASELINE configs[0] is B1 S1024 H8 D64)
run17_d64() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; cd $R; mkdir -p gpurun_out/r06
for shape in "1 1024 1024 8 8 64" "2 8192 8192 16 16 64" "1 16384 16384 32 32 64" "2 8192 8192 16 16 128"; do
  echo "fwd $shape: $(timeout 100 $K fwd $shape 1 0 0 20 2>&1 | grep TIME | awk '{print $(NF-6), "ms", $(NF-4), "TFLOP/s"}')"
  echo "bwd $shape: $(timeout 100 $K bwd $shape 1 0 0 10 2>&1 | grep TIME | awk '{print $(NF-6), "ms", $(NF-4), "TFLOP/s"}')"
done | tee gpurun_out/r06/17_d64.txt
}

# microbenchmark: the forward's per-tile instruction budget on one wave per SIMD, on two role-split waves per SIMD and on two symmetric
# waves per SIMD (tools/r06/ubench_roles.hip; built here: hipcc --offload-arch=gfx950 -O2 ... -o gpurun_tools/ubench_roles)
run18_ubench_roles() {
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06
timeout 120 $R/gpurun_tools/ubench_roles 2>&1 | tee gpurun_out/r06/18_ubench_roles.txt
}

"$@"

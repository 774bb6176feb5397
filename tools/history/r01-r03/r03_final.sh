# Round 3, closing call: the whole GPU suite (with the printed parity figures), smoke, the driver's bench command twice, the
# native suite, and one rank of configs[3] / configs[2] / configs[4] on one GPU with the wire as local copies.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -q --timeout 1500 -rP > $OUT/20_gpu_suite_final.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $OUT/20_gpu_suite_final.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/20_smoke.log 2>&1; echo "smoke rc=$?"
for i in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/20_bench_$i.json 2> $OUT/20_bench_$i.err; echo "bench $i rc=$?"
  python - <<PY
import json
d=json.loads(open("$OUT/20_bench_$i.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("bench $i: value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "bwd_ms", r["fwd_bwd"]["bwd_ms"], "64k", r["seq64k_single_gpu"].get("achieved"), r["seq64k_single_gpu"].get("sampled_parity",{}).get("max_abs_err"))
PY
done
./long-context-attention_amd/kbench suite bwd > $OUT/20_kbench_suite.log 2>&1; grep -E "SUITE|TIME" $OUT/20_kbench_suite.log | tail -12
for n in 8 4 2; do python tools/rank_emulation.py --gpus $n --iters 10 > $OUT/20_emu_$n.log 2>&1; tail -n 1 $OUT/20_emu_$n.log | cut -c1-220; done

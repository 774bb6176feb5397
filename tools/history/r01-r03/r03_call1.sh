# usage (on the GPU box, from the repo root): bash tools/r03/r03_call1.sh
# Round 3, first GPU contact: (1) the whole GPU suite with everything round 2 left staged now ON (K split through the
# binding = default policy, direct dK/dV return, operand normalisation) plus the new RCCL virtual-grid test; (2) the
# driver's bench command three times on this box (the N=1 step must land within 3 % of the kernel); (3) the dK/dV
# role ablations (native harness under rocprofv3).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q -x --timeout 1500 > $OUT/1_gpu_suite.log 2>&1; echo "1 suite rc=$?"; tail -n 4 $OUT/1_gpu_suite.log
for i in 1 2 3; do
  extra="--no-cpu-baseline"; [ $i = 1 ] && extra=""
  python bench.py --gpus 1 --steps 20 --warmup 5 $extra > $OUT/2_bench_$i.json 2> $OUT/2_bench_$i.err; echo "2 bench $i rc=$?"
  python - <<PY
import json
d=json.loads(open("$OUT/2_bench_$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("bench $i: value", d["value"], "ms/step", d["ms_per_step"], "dev-ev ms", d["ms_per_step_device_events"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "bwd", r["fwd_bwd"]["bwd_ms"], "64k", r["seq64k_single_gpu"].get("achieved"))
PY
done
bash tools/abl_bwd.sh base a_noexp b_noelem nopx nobar ab_none > $OUT/3_abl_dkdv.log 2>&1; grep -E "ABL|TF/s|FAIL" $OUT/3_abl_dkdv.log

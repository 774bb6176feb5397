# Round 3, call 2: the dK/dV kernel with the one-descriptor staging (native suite + timings), then the RCCL ordering file
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
K=./long-context-attention_amd/kbench
{ $K suite bwd | grep -E "SUITE|FAIL|TIME" | tail -30; } > $OUT/4_kbench_suite_bwd.log 2>&1; tail -5 $OUT/4_kbench_suite_bwd.log
for i in 1 2; do $K bwd 2 8192 8192 16 16 128 1 0 0 10 2>&1 | grep -E "TIME|TF/s|FAIL"; done | tee $OUT/4_bwd_c2.log
$K bwd 1 16384 16384 16 2 128 1 0 0 5 2>&1 | grep -E "TIME|TF/s|FAIL" | tee -a $OUT/4_bwd_c2.log
bash tools/abl_bwd.sh base > $OUT/4_abl_base.log 2>&1; grep ABL $OUT/4_abl_base.log
python -m pytest tests/test_gpu_rccl_order.py -q -x --timeout 1500 > $OUT/5_rccl_order.log 2>&1; echo "rccl rc=$?"; tail -n 5 $OUT/5_rccl_order.log

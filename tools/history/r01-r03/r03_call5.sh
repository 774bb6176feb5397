R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -q -x --timeout 1500 > $OUT/10_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -n 4 $OUT/10_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/10_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/10_smoke.log

R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_parity.py -q -x --timeout 1500 -k "window or cuts or k_split" > $OUT/9_window_cuts.log 2>&1; echo "window/cuts rc=$?"; tail -n 3 $OUT/9_window_cuts.log
bash tools/abl_bwd.sh base > $OUT/9_abl_base.log 2>&1; grep ABL $OUT/9_abl_base.log
./long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 1 0 0 20 2>&1 | grep TIME
for e in NONE=1 USP_PIPELINE_ULYSSES=0 USP_DKDV_LAST_HOP=fp32; do
  python tools/rank_emulation.py --gpus 8 --iters 10 --env $e > $OUT/9_emu_c5_$e.log 2>&1; echo "emu $e rc=$?"; tail -n 2 $OUT/9_emu_c5_$e.log
done

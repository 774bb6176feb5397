# usage (on the GPU box, from the repo root): bash tools/first_contact_r03.sh
# First GPU contact of what round 2 built after its GPU budget was spent (all of it host-side, gloo-green):
#   1. the whole GPU suite on the new defaults (wave-major zigzag forward, wave 0 posted in front of step 0);
#   2. the staged tests (USP_DKDV_RETURN=direct through the HIP kernels: bit-identical to the relay);
#   3. the row-range waves of the zigzag fetch on the small fixtures (USP_ZZ_PIECES=2 and 3);
#   4. the N=1 bench line (restructured tail of bench.py: deadline guard, gc.freeze);
#   2b/2c/6. the forward K split (ABI v4) through the Python binding, and what it buys the 2-GPU config's head groups;
#   5. what the extra merge epilogues of the row-range waves cost in kernel time at configs[3] (link model: 16/32 us each).
# Everything lands in gpurun_out/first_contact/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/first_contact
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q                                            > $OUT/1_gpu_suite.log 2>&1; echo "1 suite: $?"
USP_TEST_STAGED=1 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_parity.py -k "direct_dkdv or k_split or expanded_gradient" -x -q > $OUT/2_staged.log 2>&1; echo "2 staged: $?"
USP_FWD_KSPLIT=auto python -m pytest tests -m gpu -x -q -k "not fuzz"          > $OUT/2b_suite_ksplit_auto.log 2>&1; echo "2b suite with the K split policy on: $?"
./long-context-attention_amd/kbench suite                                      > $OUT/2c_kbench_suite.log 2>&1; echo "2c native suite (incl. ksplit): $?"
for w in 2 3; do
  USP_ZZ_PIECES=$w python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_rccl_order.py -x -q -k "u1r4 or u2r4 or rccl" > $OUT/3_pieces_$w.log 2>&1; echo "3 pieces $w: $?"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline                       > $OUT/4_bench_n1.log 2>&1; echo "4 bench: $?"
for w in 1 2 4; do
  python tools/rank_emulation.py --gpus 4 --iters 20 --env USP_ZZ_PIECES=$w   > $OUT/5_c4_pieces_$w.log 2>&1; echo "5 emulation pieces $w: $?"
done
for e in USP_FWD_KSPLIT=0 USP_FWD_KSPLIT=auto; do                                # the 2-GPU config's rank: head groups of 4 heads
  python tools/rank_emulation.py --gpus 2 --iters 20 --env $e                  > $OUT/6_c3_$e.log 2>&1; echo "6 emulation C3 $e: $?"
done
tail -n 3 $OUT/*.log

# usage (on the GPU box, from the repo root): bash tools/prof_round.sh [tag]
# rocprofv3 of the PRODUCT path (python, yunchang_amd._C: 16-bit epilogues, delta launch -- tools/prof_product.py):
#   * kernel trace (--stats) of the default bench.py run (the driver's N=1 command: B1 S65536 H32/Hkv4 fwd+bwd layer step), of
#     the forward + backward kernels at that shape ("w64k") and at C2 ("c2"), and of the layer-level fwd+bwd step;
#   * PMC passes of both shapes: SQ / FETCH / WRITE+GRBM in separate passes, as the MI355X guide prescribes;
# summary -> gpurun_out/prof_<tag>/summary.txt  (copy to profiles/<tag>_rocprof_summary.txt: bench.py's roofline.traffic reads
# the flash_bwd_dkdv64_kernel lines of the FIRST derived block = the N = 1 workload's shape)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/w64k $OUT/c2set
cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P="python $R/tools/prof_product.py"
rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/w64k/c5 -o c5 -- $P c5 3 > $OUT/c5.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/c2set/c2 -o c2 -- $P c2 40 > $OUT/c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/layer -o layer -- $P layer 20 > $OUT/layer.log 2>&1
# counter passes: few dispatches (per-dispatch GRBM_GUI_ACTIVE windows of many back-to-back launches overlap and overcount)
for set in "w64k c5 2" "c2set c2 4"; do
  set -- $set
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/$1/pmc_sq -o pmc -- $P $2 $3 > $OUT/$1_pmc_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$1/pmc_fetch -o pmc -- $P $2 $3 > $OUT/$1_pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/$1/pmc_write -o pmc -- $P $2 $3 > $OUT/$1_pmc_write.log 2>&1
done
export USP_KERNEL_SRC_SHA16=$(cd $R && python -c "import bench; print(bench.kernel_source_sha16())")
python $R/tools/prof_summary.py $OUT/w64k $OUT/summary_w64k.txt 1 65536 32 128 > /dev/null
python $R/tools/prof_summary.py $OUT/c2set $OUT/summary_c2.txt 2 8192 16 128 > /dev/null
{ echo "######## the N = 1 workload's shape: B1 S65536 H32/Hkv4 D128 causal (one launch per kernel = the whole step's share)"; cat $OUT/summary_w64k.txt;
  python $R/tools/kernel_isa.py | head -1;      # machine-code identity of the profiled roofline kernel (flash_bwd_dkdv64_kernel<bf16, causal>)
  echo; echo "######## BASELINE configs[1] (C2): B2 S8192 H16 D128 causal"; grep -v "^kernel_src_sha16" $OUT/summary_c2.txt;
  echo; echo "######## kernel traces of the driver's command (bench) and of the layer step"; } > $OUT/summary.txt
python $R/tools/prof_summary.py $OUT/bench $OUT/summary_bench.txt > /dev/null; grep -v "^kernel_src_sha16\|DERIVED\|shape B=" $OUT/summary_bench.txt | head -14 >> $OUT/summary.txt
python $R/tools/prof_summary.py $OUT/layer $OUT/summary_layer.txt > /dev/null; grep -v "^kernel_src_sha16\|DERIVED\|shape B=" $OUT/summary_layer.txt | head -10 >> $OUT/summary.txt
grep -E "^\{" $OUT/bench_stdout.log > $OUT/bench_line.json
grep -h "fwd\|layer" $OUT/c2.log $OUT/c5.log $OUT/layer.log | grep -v Warning >> $OUT/summary.txt
rm -rf $OUT/bench $OUT/w64k $OUT/c2set $OUT/layer
head -30 $OUT/summary.txt
du -sh $OUT

# usage (on the GPU box, from the repo root): bash tools/prof_round.sh [tag]
# rocprofv3 of the PRODUCT path (python, yunchang_amd._C: 16-bit epilogues, delta launch -- tools/prof_product.py):
#   * kernel trace (--stats) of the C2 forward + backward kernels, of the same at the metric's 64K shape, of the layer-level
#     fwd+bwd step, and of the default bench.py run (the driver's N=1 command);
#   * PMC passes of the C2 kernels: SQ / FETCH / WRITE+GRBM in separate passes, as the MI355X guide prescribes;
# summary -> gpurun_out/prof_<tag>/summary.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P="python $R/tools/prof_product.py"
rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/c2 -o c2 -- $P c2 40 > $OUT/c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/c5_64k -o c5 -- $P c5 3 > $OUT/c5.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/layer -o layer -- $P layer 20 > $OUT/layer.log 2>&1
# counter passes: few dispatches (per-dispatch GRBM_GUI_ACTIVE windows of many back-to-back launches overlap and overcount)
rocprofv3 --kernel-trace --pmc $SQ -d $OUT/pmc_sq -o pmc -- $P c2 4 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $P c2 4 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_write -o pmc -- $P c2 4 > $OUT/pmc_write.log 2>&1
export USP_KERNEL_SRC_SHA16=$(cd $R && python -c "import bench; print(bench.kernel_source_sha16())")
python $R/tools/prof_summary.py $OUT $OUT/summary.txt > /dev/null
python $R/tools/kernel_isa.py | head -1 >> $OUT/summary.txt      # machine-code identity of the profiled forward kernel
grep -E "^\{" $OUT/bench_stdout.log > $OUT/bench_line.json
grep -h "fwd\|layer" $OUT/c2.log $OUT/c5.log $OUT/layer.log | grep -v Warning >> $OUT/summary.txt
rm -rf $OUT/bench $OUT/c2 $OUT/c5_64k $OUT/layer $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
head -14 $OUT/summary.txt
du -sh $OUT

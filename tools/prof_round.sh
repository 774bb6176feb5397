# usage (on the GPU box, from the repo root): bash tools/prof_round.sh [tag]
# rocprofv3 kernel-trace + PMC passes (SQ / FETCH / WRITE+GRBM in separate passes, as the MI355X guide prescribes) of the
# three flash kernels at the C2 shape, plus the kernel trace of the default bench.py run; summary -> gpurun_out/prof_<tag>/summary.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
KBF="$R/long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 1 0 0 60"
KBB="$R/long-context-attention_amd/kbench bwd 2 8192 8192 16 16 128 1 0 0 12"
# counter passes: few dispatches (per-dispatch GRBM_GUI_ACTIVE windows of many back-to-back launches overlap and overcount)
KBF_P="$R/long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 1 0 0 5"
KBB_P="$R/long-context-attention_amd/kbench bwd 2 8192 8192 16 16 128 1 0 0 3"
rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/fwd -o fwd -- $KBF > $OUT/fwd.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/bwd -o bwd -- $KBB > $OUT/bwd.log 2>&1
for app in fwd bwd; do
  if [ $app = fwd ]; then KB=$KBF_P; else KB=$KBB_P; fi
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/pmc_sq_$app -o pmc -- $KB > $OUT/pmc_sq_$app.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$app -o pmc -- $KB > $OUT/pmc_fetch_$app.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_write_$app -o pmc -- $KB > $OUT/pmc_write_$app.log 2>&1
done
export USP_KERNEL_SRC_SHA16=$(cd $R && python -c "import bench; print(bench.kernel_source_sha16())")
python $R/tools/prof_summary.py $OUT $OUT/summary.txt > /dev/null
python $R/tools/kernel_isa.py | head -1 >> $OUT/summary.txt      # machine-code identity of the profiled forward kernel
grep -E "^\{" $OUT/bench_stdout.log > $OUT/bench_line.json
head -12 $OUT/summary.txt
du -sh $OUT

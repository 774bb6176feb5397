set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
KB="$R/long-context-attention_amd/kbench fwd 2 8192 8192 16 16 128 1 0 0 5"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o pmc -- $KB > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $KB > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_write -o pmc -- $KB > $OUT/pmc_write.log 2>&1
KB2="$R/long-context-attention_amd/kbench bwd 2 8192 8192 16 16 128 1 0 0 3"
rocprofv3 --kernel-trace --stats -d $OUT/bwd -o bwd -- $KB2 > $OUT/bwd.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq_bwd -o pmc -- $KB2 > $OUT/pmc_sq_bwd.log 2>&1
find $OUT -type f | head -50
du -sh $OUT

#!/bin/bash
# Round 4's GPU calls, one function per call (they were 31 one-shot scripts tools/r04/runN.sh; collapsed in round 5, bodies verbatim).
#   usage on the GPU box:  bash tools/r04/experiments.sh run17 > gpurun_out/r04_run17.log 2>&1
# Which call produced which file under profiles/: profiles/r04_INDEX.md.  They need the variant libraries under abl/ that
# tools/r04/build_variant64.sh builds (names in the bodies); DEV scripts, kept as the record of what was measured how.

# round 4, GPU call 1: the 64-rows-per-wave forward (usp_flash_fwd64.hip) -- correctness through kbench + the C oracle,
# then A/B timing against the 8-wave kernel.   bash tools/r04/run1.sh > gpurun_out/r04_run1.log 2>&1
run1() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
echo "== forced USP_FWD_WAVES=64: correctness =="
export USP_FWD_WAVES=64
for shape in "1 256 256 1 1 128 0 0" "1 256 256 1 1 128 1 0" "2 512 512 4 4 128 1 0" "1 384 640 4 2 128 0 0" \
             "1 320 320 4 1 128 1 1" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 1 1 1 1 128 1 0" \
             "1 512 256 2 2 128 0 0" "1 256 512 2 2 128 0 0" "1 2048 2048 2 1 128 1 0" "2 2048 2048 16 16 128 1 0" \
             "1 4096 4096 20 4 128 1 1" "1 3000 5000 9 3 128 0 0" "1 5000 3000 8 8 128 1 0"; do
  timeout 300 $K fwd $shape 1 0 || echo "RC=$? for $shape"
done
timeout 120 $K fwdmerge 1 256 512 2 2 128 0 || echo "RC=$?"
timeout 120 $K fwdmerge 1 64 192 2 1 128 0 || echo "RC=$?"
timeout 120 $K fwdmerge 2 2048 4096 8 2 128 1 || echo "RC=$?"
echo "== C2 full check, new kernel =="
timeout 600 $K fwd 2 8192 8192 16 16 128 1 0 1 0 || echo "RC=$?"
echo "== timing A/B =="
for rep in 1 2; do for w in 8 64; do
  export USP_FWD_WAVES=$w
  echo "[waves $w]"; timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME
  timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 50 | grep TIME
  timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 30 | grep TIME
  timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME
done; done
unset USP_FWD_WAVES
echo "== suite (default policy) =="
timeout 900 $K suite 2>&1 | grep -v "^CHECK.*ok$" | tail -40
}

# round 4, GPU call 2: fwd64 variants (DMA placement, schedule knobs, ablations).  DEV script.
run2() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 1 0 0 200 > /dev/null      # warm the clocks
echo "== correctness of the placements (oracle) =="
for v in base dmaA3 dmaA2 dmaAB4 dmaB2 dmaA4 nea46 nea50 pf3 pf4 lead6 max12; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
done
echo "== timing =="
for rep in 1 2; do
  for v in base dmaA3 dmaA2 dmaAB4 dmaB2 dmaA4 nea46 nea50 pf3 pf4 lead6 max12 nodma noexp nolds nobar noexplds none; do
    for shape in "2 8192 8192 16 16 128 1" "2 8192 8192 16 16 128 0"; do
      echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd $shape 0 0 60 | grep TIME)"
    done
  done
done
}

# round 4, GPU call 3: what does an LDS-DMA piece cost beside LDS fragment reads / in VALU-only gaps?  DEV script.
run3() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in base dmaAB7 nolds nolds_nodma nolds_dmaA4 nolds_dmaA2 nolds_dmaAB8 nolds_dmaB3 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done
}

# round 4, GPU call 4: do the four lockstep waves of a workgroup queue behind each other at the CU's texture addresser?  DEV script.
run4() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in stag41 stag51 stag61 stag72; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
done
for rep in 1 2 3; do
  for v in base dmaA4 stag41 stag51 stag61 stag72 nolds nolds_stag41 nolds_stag61 nolds_nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done
}

# round 4, GPU call 5: LDS-DMA mechanism probes (plain loads, register staging, half the pieces, 4-byte pieces).  DEV script.
run5() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in probe2 probe2A3; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 1 3000 5000 9 3 128 0 0 1 0 | cut -c1-150)"
done
for rep in 1 2 3; do
  for v in base dmaA3 probe1 probe1A3 probe2 probe2A3 probe3 probe5 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done
}

# round 4, GPU call 6: is the per-piece M0 write what an LDS-DMA piece costs?  DEV script.
run6() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in base dmaA3 probe6 probe6A3 probe7 probe7A3 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done
}

# round 4, GPU call 7: register staging with the LDS writes spread over phase B.  DEV script.
run7() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in stgA3B3 stgA2B3 stgA3B2 stgHB3; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 2048 2048 16 16 128 1 0 1 0 | cut -c1-150)"
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 1 3000 5000 9 3 128 0 0 1 0 | cut -c1-150)"
done
for rep in 1 2 3; do
  for v in base probe7 stgA3B3 stgA2B3 stgA3B2 stgHB3 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done
}

# round 4, GPU call 8: one M0 write per tile (pieces addressed by the immediate offset).  DEV script.
run8() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for v in m0base m0A3; do
  for shape in "2 2048 2048 16 16 128 1 0" "1 3000 5000 9 3 128 0 0" "1 333 200 2 2 128 1 0" "1 200 333 3 1 128 1 0" "1 4096 4096 20 4 128 1 1"; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd $shape 1 0 | cut -c1-150)"
  done
done
for rep in 1 2 3; do
  for v in base m0base m0A3 m0A2 m0A4 m0AB6 m0A1 nodma; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  done
done
}

# round 4, GPU call 9: ablations with REAL tiles in LDS (no-DMA build keeps the prologue's loads).  DEV script.
run9() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
$K fwd 2 8192 8192 16 16 128 0 0 0 200 > /dev/null      # warm the clocks
for rep in 1 2 3; do
  for v in m0base m0AB6 nodma2 nodma nolds2 noexp2 nodma2_nolds; do
    echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 60 | grep TIME)"
  done
done
}

# round 4, GPU call 10: fwd64 with every tile through the pipelined code (MODE 1 diagonal, MODE 2 last tile).  DEV script.
run10() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
for shape in "1 256 256 1 1 128 0 0" "1 256 256 1 1 128 1 0" "2 512 512 4 4 128 1 0" "1 384 640 4 2 128 0 0" \
             "1 320 320 4 1 128 1 1" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 1 1 1 1 128 1 0" \
             "1 512 256 2 2 128 0 0" "1 256 512 2 2 128 0 0" "1 2048 2048 2 1 128 1 0" "2 2048 2048 16 16 128 1 0" \
             "1 4096 4096 20 4 128 1 1" "1 3000 5000 9 3 128 0 0" "1 5000 3000 8 8 128 1 0" "1 3000 5000 6 2 128 1 0" "1 65 191 2 1 128 0 1" "2 77 77 2 2 128 1 0"; do
  timeout 300 $K fwd $shape 1 0 | cut -c1-160 || echo "RC=$? for $shape"
done
timeout 120 $K fwdmerge 1 256 512 2 2 128 0 | cut -c1-160
timeout 120 $K fwdmerge 1 64 192 2 1 128 0 | cut -c1-160
timeout 120 $K fwdmerge 2 2048 4096 8 2 128 1 | cut -c1-160
timeout 600 $K fwd 2 8192 8192 16 16 128 1 0 1 0 | cut -c1-160
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_FWD_WAVES=$w
  echo "[waves $w] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 50 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 30 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME)"
done; done
}

# round 4, GPU call 11: fwd64 after the register-pressure fix.  DEV script.
run11() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
export USP_FWD_WAVES=64
for shape in "2 512 512 4 4 128 1 0" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "2 2048 2048 16 16 128 1 0" "1 3000 5000 9 3 128 0 0" "1 5000 3000 8 8 128 1 1"; do
  timeout 300 $K fwd $shape 1 0 | cut -c1-160 || echo "RC=$? for $shape"
done
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_FWD_WAVES=$w
  echo "[waves $w] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 50 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 1 16384 16384 16 2 128 1 0 0 30 | grep TIME)"
  echo "[waves $w] $(timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME)"
done; done
}

# round 4, GPU call 12: the one-wave-per-SIMD dK/dV kernel (usp_flash_bwd64.hip): kbench parity + A/B timing.  DEV script.
run12() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for shape in "1 256 256 1 1 128 0 0" "1 256 256 2 2 128 1 0" "2 512 512 4 2 128 1 0" "1 384 640 4 2 128 0 0" \
             "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 1 1 1 1 128 1 0" "2 77 77 2 2 128 1 0" \
             "1 2048 2048 4 2 128 1 0" "1 3000 5000 6 2 128 1 0" "1 5000 3000 4 4 128 1 0" "1 1000 1300 3 3 128 0 0"; do
  timeout 300 $K bwd $shape 1 0 | cut -c1-170 || echo "RC=$? for $shape"
done
USP_KBENCH_BWD_SPLITS=2,3 timeout 300 $K bwd 1 2048 2048 4 2 128 1 0 1 0 | cut -c1-170
USP_KBENCH_BWD_SPLITS=1,4 timeout 300 $K bwd 1 1000 1300 3 3 128 0 0 1 0 | cut -c1-170
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_BWD_WAVES=$w
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 1 16384 16384 16 2 128 1 0 0 10 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME)"
done; done
unset USP_BWD_WAVES
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04b/trace -o t -- $K bwd 2 8192 8192 16 16 128 1 0 0 12 > /dev/null 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_r04b $R/gpurun_out/prof_r04b/summary.txt > /dev/null; rm -rf $R/gpurun_out/prof_r04b/trace
grep -A8 "calls" $R/gpurun_out/prof_r04b/summary.txt | head -14
}

# round 4, GPU call 14: dkdv64 without the per-tile memory round trip of the statistics wave.  DEV script.
run13() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for shape in "2 512 512 4 2 128 1 0" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 2048 2048 4 2 128 1 0" "1 1000 1300 3 3 128 0 0"; do
  timeout 300 $K bwd $shape 1 0 | cut -c1-170 || echo "RC=$? for $shape"
done
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_BWD_WAVES=$w
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME)"
done; done
unset USP_BWD_WAVES
export TMPDIR=/tmp; cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
KB="$K bwd 2 8192 8192 16 16 128 1 0 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04d/trace -o t -- $KB 12 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $R/gpurun_out/prof_r04d/pmc_sq -o pmc -- $KB 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_r04d/pmc_grbm -o pmc -- $KB 3 > /dev/null 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_r04d $R/gpurun_out/prof_r04d/summary.txt > /dev/null; rm -rf $R/gpurun_out/prof_r04d/trace $R/gpurun_out/prof_r04d/pmc_sq $R/gpurun_out/prof_r04d/pmc_grbm
}

# round 4, GPU call 14: dkdv64 without the per-tile memory round trip of the statistics wave.  DEV script.
run14() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
for shape in "2 512 512 4 2 128 1 0" "1 200 333 3 1 128 1 0" "1 333 200 2 2 128 1 0" "1 2048 2048 4 2 128 1 0" "1 1000 1300 3 3 128 0 0"; do
  timeout 300 $K bwd $shape 1 0 | cut -c1-170 || echo "RC=$? for $shape"
done
echo "== timing =="
for rep in 1 2 3; do for w in 8 64; do
  export USP_BWD_WAVES=$w
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "[dkdv waves $w] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME)"
done; done
unset USP_BWD_WAVES
export TMPDIR=/tmp; cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
KB="$K bwd 2 8192 8192 16 16 128 1 0 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04d/trace -o t -- $KB 12 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $R/gpurun_out/prof_r04d/pmc_sq -o pmc -- $KB 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_r04d/pmc_grbm -o pmc -- $KB 3 > /dev/null 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_r04d $R/gpurun_out/prof_r04d/summary.txt > /dev/null; rm -rf $R/gpurun_out/prof_r04d/trace $R/gpurun_out/prof_r04d/pmc_sq $R/gpurun_out/prof_r04d/pmc_grbm
}

# round 4, GPU call 15: dkdv64 ablations (P hand-off, exp) and element-stream windows.  DEV script.
run15() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
$K bwd 2 8192 8192 16 16 128 1 0 0 30 > /dev/null
for v in b_e1856 b_e1650; do echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K bwd 1 2048 2048 4 2 128 1 0 1 0 | cut -c1-150)"; done
for rep in 1 2 3; do for v in b_base b_nopx b_noexp b_e1856 b_e1650; do
  echo "$v: $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
done; done
}

# round 4, GPU call 16: after the pruning of the 8-wave kernels (dQ kernel re-generated) and the 64-bit cursors:
# native suite incl. backward, a head whose rows span 3 GiB, A/B timing.  DEV script.
run16() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | head -30
echo "== rows 2 MiB apart: 1500 rows span 3 GiB =="
USP_KBENCH_ROWSTRIDE=1048576 timeout 600 $K bwd 1 1500 1500 2 1 128 1 0 1 0 | cut -c1-170
USP_KBENCH_ROWSTRIDE=1048576 timeout 600 $K bwd 1 1300 1500 1 1 128 0 0 1 0 | cut -c1-170
echo "== timing =="
for rep in 1 2; do
  echo "$(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME)"
  echo "$(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
  echo "$(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME)"
  echo "[USP_BWD_WAVES=8] $(USP_BWD_WAVES=8 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME)"
done
}

# round 4, GPU call 17: where do dkdv64's exposed waits come from?  Timing-only A/B builds that keep REAL tiles in LDS
# (results of the variants are wrong by construction; only kernel time is read).  DEV script.
run17() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
run() { LD_LIBRARY_PATH=$R/abl/$1 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c1-150; }
for rep in 1 2 3; do
  for v in b_base b_nodma b_prebar b_nobar b_any b_any_prebar; do echo "[$v] $(run $v)"; done
done
}

# round 4, GPU call 18: dkdv64's LDS-DMA: does its cost move with the phase that issues it (latency exposed at the drain)
# or with the number of pieces (issue / LDS write bandwidth)?  Timing-only builds (b_half reads stale dO).  DEV script.
run18() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
run() { LD_LIBRARY_PATH=$R/abl/$1 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150; }
for rep in 1 2 3; do
  for v in b_base b_nodma b_half b_ph1 b_ph2 b_ph3; do echo "[$v] $(run $v)"; done
done
}

# round 4, GPU calls 19, 20: dkdv64 with the chains started from the row constants (C operand), K pre-scaled; then cheaper
# descriptors, role B reading its first chain ahead of the barrier, the NaN tail behind every kbench tensor.  DEV script.
run19() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | head -30
for rep in 1 2 3; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[base] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new 64K ] $(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
echo "[base 64K] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
timeout 300 $K bwd 2 8192 8192 16 16 128 1 0 1 0 | cut -c1-200
}

# round 4, GPU call 21: (1) the suite with the NaN tail behind every kbench tensor (the kbench of call 20 was stale);
# (2) dkdv64's iteration anatomy from s_memtime stamps around the DMA drain and the barrier (b_tm build).  DEV script.
run21() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
echo "== anatomy (C2 shape) =="
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TM" | sort -k3n -k5n | awk 'NR<=64'
for rep in 1 2; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[base] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
}

# round 4, GPU call 22: dkdv64 after the issue-slot work (packing lag, one LDS wait per phase): suite, timing, anatomy.
run22() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
echo "== anatomy (C2 shape, workgroup 0: key block 0, 128 query tiles) =="
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TM" | sort -k3n -k5n | awk '{k=$3" "$5; if (c[k]++ < 2) print}'
for rep in 1 2 3; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[base] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new 64K ] $(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
echo "[base 64K] $(LD_LIBRARY_PATH=$R/abl/b_base timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
}

# round 4, GPU call 24: the one-wave-per-SIMD dQ kernel (usp_flash_bwd_dq64.hip), first run: suite, timing against the
# 8-wave dQ kernel (USP_BWD_WAVES=8 forces both 8-wave backward kernels).  DEV script.
run24() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1500 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -40
for rep in 1 2 3; do
  echo "[new    ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[8-wave ] $(USP_BWD_WAVES=8 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new 64K ] $(timeout 120 $K bwd 1 65536 65536 32 4 128 1 0 0 2 | grep TIME | cut -c60-150)"
echo "[new full] $(timeout 120 $K bwd 2 8192 8192 16 16 128 0 0 0 10 | grep TIME | cut -c60-150)"
timeout 300 $K bwd 2 8192 8192 16 16 128 1 0 1 0 | cut -c1-200
timeout 300 $K bwd 1 4096 4096 8 2 128 0 0 1 0 | cut -c1-200
}

# round 4, GPU call 25: dq64 against the 8-wave dQ kernel on one box (USP_BWD_DQ_WAVES=8 forces the latter only), and the
# per-kernel durations from a kernel trace.  DEV script.
run25() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  echo "[dq64   ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[dq 8w  ] $(USP_BWD_DQ_WAVES=8 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- $K bwd 2 8192 8192 16 16 128 1 0 0 20 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/p1/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
}

# round 4, GPU call 26: per-kernel durations of the backward (kernel trace) + dq64 knob variants.  DEV script.
run26() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- $K bwd 2 8192 8192 16 16 128 1 0 0 20 > /tmp/rp.log 2>&1
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), '%9.1f us' % (float(r['AverageNs'])/1e3))
"
for v in q_pf2 q_pf4 q_pf6 q_y16 q_y28; do
  [ -d $R/abl/$v ] && echo "[$v] $(LD_LIBRARY_PATH=$R/abl/$v timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[base ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
}

# round 4, GPU call 27: backward kernels at the C2 shape through kbench: kernel trace + two PMC passes -> summary.  DEV script.
run27() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; OUT=$R/gpurun_out/prof_bwd27
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="$K bwd 2 8192 8192 16 16 128 1 0 0 10"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc2 -o p2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc3 -o p3 -- $CMD > $OUT/pmc3.log 2>&1
python3 $R/tools/prof_summary.py $OUT $OUT/summary.txt 2 8192 16 128 > /dev/null 2>&1
head -60 $OUT/summary.txt
find $OUT -name "*.db" -size +20M -delete
}

# round 4, GPU call 28: where a causal forward item's time goes (s_memtime stamps, f_tm build), C2 shape.  DEV script.
run28() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
LD_LIBRARY_PATH=$R/abl/f_tm timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TF" | sort -k3n -k5n -k7n | awk '{k=$3" "$5" "$7; if (c[k]++ < 1) print}' | head -90
echo "[base causal] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
echo "[base full  ] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 100 | grep TIME | cut -c60-150)"
}

# round 4, GPU call 29: forward 4x64 with the first-tile reference path and row-counted descriptors: suite, timing vs previous.
run29() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1200 $K suite fwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
for rep in 1 2 3; do
  echo "[new  causal] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
  echo "[prev causal] $(LD_LIBRARY_PATH=$R/abl/f_prev timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
done
echo "[new  full] $(timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 100 | grep TIME | cut -c60-150)"
echo "[prev full] $(LD_LIBRARY_PATH=$R/abl/f_prev timeout 120 $K fwd 2 8192 8192 16 16 128 0 0 0 100 | grep TIME | cut -c60-150)"
echo "[new  64K] $(timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME | cut -c60-150)"
echo "[prev 64K] $(LD_LIBRARY_PATH=$R/abl/f_prev timeout 120 $K fwd 1 65536 65536 32 4 128 1 0 0 3 | grep TIME | cut -c60-150)"
}

# round 4, GPU call 30: prologues reordered (first tiles' DMA in front of the resident-operand loads) in the three 64-row
# kernels: suite, timing against the previous library.  DEV script.
run30() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
timeout 1500 $K suite 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
for rep in 1 2 3; do
  echo "[new  fwd] $(timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
  echo "[prev fwd] $(LD_LIBRARY_PATH=$R/abl/prev2 timeout 120 $K fwd 2 8192 8192 16 16 128 1 0 0 100 | grep TIME | cut -c60-150)"
  echo "[new  bwd] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[prev bwd] $(LD_LIBRARY_PATH=$R/abl/prev2 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
}

# round 4, GPU call 31: per-item anatomy of the two backward kernels (s_memtime; b_tm / q_tm builds), C2 shape.  DEV script.
run31() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench
LD_LIBRARY_PATH=$R/abl/b_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TI" | sort -k3n -k5n -k7n | awk '{k=$3" "$5" "$7; if (c[k]++ < 1) print}' | head -50
LD_LIBRARY_PATH=$R/abl/q_tm timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 1 2>&1 | grep "^TQ" | sort -k3n -k5n -k7n | awk '{k=$3" "$5" "$7; if (c[k]++ < 1) print}' | head -50
}

# round 4, GPU call 32: whole-row-piece 16-bit epilogue stores in the two 64-row backward kernels: parity (python path, 16-bit
# outputs) and product-path timing against the previous library (swapped into the box's scratch copy).  DEV script.
run32() {
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu -k "64row or backward or bwd or golden or seq64k" 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
for rep in 1 2; do
  echo "[new ] $(python tools/prof_product.py c2 40 | tail -1)"
  cp long-context-attention_amd/libusp_hip.so /tmp/new.so; cp abl/prev3/libusp_hip.so long-context-attention_amd/libusp_hip.so
  echo "[prev] $(python tools/prof_product.py c2 40 | tail -1)"
  cp /tmp/new.so long-context-attention_amd/libusp_hip.so
done
echo "[new 64K] $(python tools/prof_product.py c5 3 | tail -1)"
}

# round 4, GPU call 33: resident-operand loads of an item merged under one wait (dK/dV: 2 -> 1 round trips, dQ: 4 -> 2):
# native suite, kbench timing against the previous library, python product path.  DEV script.
run33() {
R=$GRAFT_REPO_ROOT; K=$R/long-context-attention_amd/kbench; cd $R
timeout 1500 $K suite bwd 2>&1 | grep -v "^CHECK.*ok$" | grep -v "^TIME" | head -30
for rep in 1 2 3; do
  echo "[new ] $(timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
  echo "[prev] $(LD_LIBRARY_PATH=$R/abl/prev3 timeout 120 $K bwd 2 8192 8192 16 16 128 1 0 0 20 | grep TIME | cut -c60-150)"
done
echo "[new python] $(python tools/prof_product.py c2 40 2>/dev/null | tail -1)"
}

case "$1" in
  run1|run2|run3|run4|run5|run6|run7|run8|run9|run10|run11|run12|run13|run14|run15|run16|run17|run18|run19|run21|run22|run24|run25|run26|run27|run28|run29|run30|run31|run32|run33) "$1" ;;
  *) echo "usage: $0 {run1|run2|run3|run4|run5|run6|run7|run8|run9|run10|run11|run12|run13|run14|run15|run16|run17|run18|run19|run21|run22|run24|run25|run26|run27|run28|run29|run30|run31|run32|run33}"; exit 64 ;;
esac

# usage (on the GPU box, from the repo root): bash tools/native_r03.sh        -- native harness only: a few seconds, no python
# The questions round 2 left for the first native call of the next round (each line answers one):
K=./long-context-attention_amd/kbench
OUT=gpurun_out/native; mkdir -p $OUT
{
echo "== 1. the whole native suite on the ABI v4 library (forward incl. ksplit groups, backward)"
$K suite bwd | grep -E "SUITE|FAIL|TIME"
echo "== 2. K split beside a transfer (interleavable launches: one workgroup per item) -- the form the 2-GPU config runs"
USP_KBENCH_FLAGS=1 $K ksplit 1 16384 16384 4 4 128 1 0 0 10
USP_KBENCH_FLAGS=1 $K ksplit 1 16384 16384 2 2 128 1 0 0 10
echo "== 3. where the gain ends: 6 and 8 heads (384 / 512 items), one head (64 items), non-causal few-head ring steps"
$K ksplit 1 16384 16384 6 6 128 1 0 0 10
$K ksplit 1 16384 16384 8 8 128 1 0 0 10
$K ksplit 1 16384 16384 1 1 128 1 0 0 10
$K ksplit 1 8192 8192 4 4 128 0 0 0 10
echo "== 4. GQA few-head groups (the 8-GPU config's sub-groups: 4 and 2 query heads on one KV head), S = 16384"
$K ksplit 1 16384 16384 4 1 128 1 0 0 10
$K ksplit 1 16384 16384 2 1 128 1 0 0 10
echo "== 5. what the backward loses on the same few-head shapes (no cut there yet): 8 / 4 / 2 heads"
$K bwd 1 16384 16384 8 8 128 1 0 0 5
$K bwd 1 16384 16384 4 4 128 1 0 0 5
$K bwd 1 16384 16384 2 2 128 1 0 0 5
} > $OUT/native_r03.log 2>&1
cat $OUT/native_r03.log

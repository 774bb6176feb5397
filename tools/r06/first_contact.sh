#!/bin/bash
# First contact with an N-GPU MI355X node (N = 2 | 4 | 8): every stage under its own deadline, in the order that risks the least
# first; one JSON at the end.  Nothing in this package has ever crossed between two devices: this is the command
# to run when a multi-GPU lease exists.      usage:  bash tools/r06/first_contact.sh 8 [outdir]
#   1 link      RCCL send/recv around the ranks + the pair all_to_all_single (tools/r06/first_contact.py --stage link)
#   2 parity    the real layer on the real grid, small size, forward + backward vs exact fp64 attention: safe mode, then the
#               library default (pipelined exchange beside the ring, self-chunk start, row-chunked tails), then + relayed exchange
#   3 bench     bench.py --gpus N with USP_SAFE_COMM=1 (ONE communicator in flight at a time), no overlap probe
#   4 bench     bench.py --gpus N as the driver runs it: safe -> default -> + relay under its own deadlines, overlap probe
# A stage that times out or fails is recorded and the script goes on to the next; the exit code is 0 unless stage 1 failed.
N=${1:-8}
OUT=${2:-gpurun_out/first_contact}
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R"; mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1"
port=29600
stage() {   # stage <name> <seconds> <command...>
  local name=$1 limit=$2; shift 2
  port=$((port + 1))
  local t0=$(date +%s)
  timeout "$limit" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "{\"stage\": \"$name\", \"rc\": $rc, \"seconds\": $(( $(date +%s) - t0 )), \"limit_s\": $limit}" > "$OUT/$name.status.json"
  echo "[first_contact] $name: rc $rc in $(( $(date +%s) - t0 )) s"
  return $rc
}
stage 1_link 300 $RUN --master-port $port tools/r06/first_contact.py --stage link --out "$OUT/1_link.json" || { echo "[first_contact] RCCL does not come up: stop"; exit 1; }
for mode in safe default relay; do
  [ "$mode" = relay ] && [ "$N" != 8 ] && continue
  stage 2_parity_$mode 600 $RUN --master-port $((port + 1)) tools/r06/first_contact.py --stage parity --mode $mode --out "$OUT/2_parity_$mode.json"
done
USP_SAFE_COMM=1 stage 3_bench_safe 900 $RUN --master-port $((port + 1)) bench.py --gpus $N --steps 10 --warmup 3 --no-overlap
grep -E '^\{' "$OUT/3_bench_safe.log" | tail -1 > "$OUT/3_bench_safe.json"
stage 4_bench_staged 1800 $RUN --master-port $((port + 1)) bench.py --gpus $N --steps 20 --warmup 5
grep -E '^\{' "$OUT/4_bench_staged.log" | tail -1 > "$OUT/4_bench_staged.json"
python - "$OUT" "$N" <<'PY'
import glob, json, os, sys
out, n = sys.argv[1], int(sys.argv[2])
def load(p):
    try:
        return json.load(open(p))
    except Exception:
        return None
res = {"n_gpus": n, "stages": [load(p) for p in sorted(glob.glob(os.path.join(out, "*.status.json")))],
       "link": load(os.path.join(out, "1_link.json")),
       "parity": {m: load(os.path.join(out, f"2_parity_{m}.json")) for m in ("safe", "default", "relay")}}
for tag in ("3_bench_safe", "4_bench_staged"):
    line = load(os.path.join(out, tag + ".json"))
    res[tag] = None if line is None else {k: line.get(k) for k in ("value", "unit", "ms_per_step", "ms_per_step_rank_min_max", "frac_of_mfma_roofline",
                                                                 "comm_modes_ms_per_step", "overlap", "parity_max_abs_err_vs_reference_op")} | \
        {"comm_mode": line["config"].get("comm_mode"), "derived": line["config"].get("derived")}
json.dump(res, open(os.path.join(out, "first_contact.json"), "w"), indent=1)
print(json.dumps(res)[:2000])
PY

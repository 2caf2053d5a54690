#!/bin/bash
# The headline launch on N(0,1) operands and on all-zero operands under the same counters (run on the GPU box via gpurun).
# Same instruction stream, same wave-cycle count expected; what differs is the clock the power management grants
# (MI355X_MICROARCH.md, DVFS note) -> separates "stalled" from "power-capped" for THIS kernel.
#   usage: tools/zero_vs_random.sh [outdir] [workload]
set -u
OUT=${1:-gpurun_out/zero_vs_random}
WORKLOAD=${2:-fwd_bf16_d128}
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p "$OUT"
for FILL in normal zero; do
  O="$REPO/$OUT/$FILL"; mkdir -p "$O"
  ARGS="--workload $WORKLOAD --steps 20 --warmup 5 --no-cpu-baseline --fill $FILL"
  python bench.py $ARGS 2>/dev/null | tail -n 1 > "$O/bench_line.json"
  cd /tmp
  rocprofv3 --kernel-trace --stats -d "$O/stats" -o stats -- python "$REPO/bench.py" $ARGS > "$O/stats.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE -d "$O/pmc1" -o pmc1 -- python "$REPO/bench.py" $ARGS > "$O/pmc1.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d "$O/pmc2" -o pmc2 -- python "$REPO/bench.py" $ARGS > "$O/pmc2.log" 2>&1
  cd "$REPO"
  python tools/summarize_prof.py "$OUT/$FILL" "bench.py $ARGS" > "$O/summary.txt" 2>&1
  rm -rf "$O/stats" "$O/pmc1" "$O/pmc2"
done
{
  echo "# headline launch, N(0,1) operands vs all-zero operands (tools/zero_vs_random.sh $WORKLOAD)"
  for FILL in normal zero; do
    echo; echo "##### fill = $FILL"
    python - "$OUT/$FILL/bench_line.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("un-profiled: ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "fp32-intermediates",
      (d["config"].get("fp32_intermediates") or {}).get("ms_per_step"))
PY
    cat "$OUT/$FILL/summary.txt"
  done
} > "$OUT/zero_vs_random.txt"
cat "$OUT/zero_vs_random.txt"

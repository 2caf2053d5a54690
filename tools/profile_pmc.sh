#!/bin/bash
# Kernel-trace stats + PMC passes for the headline forward kernel (run on the GPU box via gpurun).
# Counters go in separate passes (SQ block has 8 slots; FETCH_SIZE and WRITE_SIZE cannot share a pass)
# and never together with sys/hip traces (MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   usage: tools/profile_pmc.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/prof}; shift || true
ARGS=${@:-"--steps 5 --warmup 2 --no-cpu-baseline"}
export TMPDIR=/tmp
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/stats" -o stats -- python "$REPO/bench.py" $ARGS > "$REPO/$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE -d "$REPO/$OUT/pmc1" -o pmc1 -- python "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d "$REPO/$OUT/pmc2" -o pmc2 -- python "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc2.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$REPO/$OUT/pmc3" -o pmc3 -- python "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc3.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$REPO/$OUT/pmc4" -o pmc4 -- python "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc4.log" 2>&1
cd "$REPO"
# workload and kernel variant of the profiled command, for profiles/traffic.json (bench.py reads it back, hash-checked)
WORKLOAD=$(python - <<PY
import shlex
a = shlex.split("$ARGS")
print(a[a.index("--workload") + 1] if "--workload" in a else "fwd_bf16_d128")
PY
)
export MFA_PROFILED_VARIANT=$(python bench.py $ARGS 2>/dev/null | tail -n 1 | python -c "import json,sys; print(json.load(sys.stdin)['config']['kernel_variants'][0])")
python tools/summarize_prof.py "$OUT" "bench.py $ARGS" "$WORKLOAD" > "$OUT/summary.txt" 2>&1
cp "$OUT/summary.txt" "profiles/${ROUND:-r03}_${WORKLOAD}_summary.txt"
cp profiles/traffic.json "$OUT/traffic.json"   # gpurun merges only gpurun_out/ back: copy it into profiles/ after the call
# the raw rocprofv3 databases are large (gpurun copies at most 64 MiB back): keep the summary, the logs and traffic.json
[ -n "${KEEP_RAW:-}" ] || rm -rf "$OUT/stats" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/pmc4"
cat "$OUT/summary.txt"

#!/bin/bash
# round 4, GPU call 2: the product library with the in-place transposed kernels (was the candidate libmfa_hip_tr.so)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_call2
mkdir -p "$OUT"
sha256sum metal_flash_attention_amd/libmfa_hip.so > "$OUT/library.sha256"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > "$OUT/pytest_gpu.txt"; tail -3 "$OUT/pytest_gpu.txt"
cp gpurun_out/variant_coverage.json "$OUT/" 2>/dev/null
for args in "" "--mixed" "--transposed" "--transposed --mixed" "--mixed --fill zero" "--mixed --causal"; do
  name=$(echo "bucket_perf $args" | tr -s ' -' '__')
  timeout 200 python tools/bucket_perf.py $args 2>&1 | grep -v amdgpu.ids > "$OUT/$name.txt"; cat "$OUT/$name.txt"
done
for w in fwd_bf16_d128 fwd_bf16_d256_transposed fwd_bf16_d128_transposed fwdbwd_bf16_d128_transposed fwdbwd_bf16_d128_transposed_ws \
         fwdbwd_bf16_d128_mixed dq_bf16_d128 dkv_bf16_d128; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_$w.json"
done
for w in dq_bf16_d128 dkv_bf16_d128 fwdbwd_bf16_d128_mixed; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline --fill zero 2>/dev/null | tail -1 > "$OUT/bench_${w}_zero.json"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("bench_")[1][:-5], d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("frac"), d["roofline"]["launch_form"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 200 python tools/bwd_block_overhead.py 2>&1 | grep -v amdgpu.ids > "$OUT/bwd_block_overhead.txt"; cat "$OUT/bwd_block_overhead.txt"

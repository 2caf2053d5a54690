mkdir -p gpurun_out/r05_final3; OUT=gpurun_out/r05_final3
sha256sum metal_flash_attention_amd/libmfa_hip.so > $OUT/library.sha256
MFA_VARIANT_COVERAGE=1 timeout 560 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > $OUT/pytest_gpu.txt; tail -2 $OUT/pytest_gpu.txt
FAST=1 ROUND=r05c bash tools/round_evidence.sh > gpurun_out/r05c_evidence.log 2>&1; tail -5 gpurun_out/r05c_evidence.log | cut -c1-300
cp -r gpurun_out/r05c_final/* $OUT/
for w in dq_bf16_d256 dkv_bf16_d256 dq_bf16_d128 dkv_bf16_d128 fwdbwd_bf16_d128_mixed fwd_bf16_d256_mixed fwd_bf16_d64 fwd_bf16_d128_causal fwd_bf16_d128_fp32mid; do
  timeout 100 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$w.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("bench_")[1][:-5], d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"))
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 120 python tools/lib_ab.py --rounds 1 --fills normal --workloads fwd_bf16_d128,fwd_bf16_d64,fwdbwd_bf16_d128_mixed 2>&1 | grep -v amdgpu.ids | tee $OUT/lib_ab_forward.txt

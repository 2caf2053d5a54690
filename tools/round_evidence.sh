#!/bin/bash
# The evidence of a round in ONE gpurun call on the round's LAST product library (what profiles/rNN_final/ holds):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'ROUND=r04 bash tools/round_evidence.sh'
# then copy gpurun_out/${ROUND}_final/ to profiles/${ROUND}_final/ and commit profiles/traffic.json (profile_pmc.sh rewrites it with
# the library's SHA-256: bench.py reports `traffic` only for the library that hash belongs to -- any later change of a product
# translation unit needs this call again).  ~20 GPU-minutes with the full GPU suite; FAST=1 skips the suite and the side workloads.
cd "$(dirname "$0")/.." || exit 1
ROUND=${ROUND:-rXX}
OUT=gpurun_out/${ROUND}_final
mkdir -p "$OUT"
sha256sum metal_flash_attention_amd/libmfa_hip.so > "$OUT/library.sha256"
if [ -z "$FAST" ]; then
  MFA_VARIANT_COVERAGE=1 timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > "$OUT/pytest_gpu.txt"
  tail -3 "$OUT/pytest_gpu.txt"
  cp gpurun_out/variant_coverage.json "$OUT/variant_coverage.json" 2>/dev/null   # (tests/conftest.py: variant name -> tests; copy to tests/golden/)
fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1; tail -1 "$OUT/smoke.txt"
timeout 300 python bench.py 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_default.json"; cut -c1-300 "$OUT/bench_default.json"
# kernel trace + the separate PMC passes of the headline (MI355X_MICROARCH.md's recipe), summary + traffic.json
if [ -z "$FAST" ]; then   # (first: the headline's pass below leaves ITS traffic.json behind)
  ROUND=$ROUND timeout 400 bash tools/profile_pmc.sh "$OUT/prof_f32" --steps 5 --warmup 2 --no-cpu-baseline --workload fwdbwd_f32_d128 > "$OUT/prof_f32.log" 2>&1
fi
# BASELINE config 2 (attn_fwd16_p6): kernel trace + PMC passes
ROUND=$ROUND timeout 400 bash tools/profile_pmc.sh "$OUT/prof_d64" --steps 20 --warmup 5 --no-cpu-baseline --workload fwd_bf16_d64 > "$OUT/prof_d64.log" 2>&1; tail -12 "$OUT/prof_d64.log" | head -12
ROUND=$ROUND timeout 400 bash tools/profile_pmc.sh "$OUT/prof" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof.log" 2>&1; tail -25 "$OUT/prof.log" | head -40
if [ -z "$FAST" ]; then
  for w in fwd_bf16_d128_fp32mid fwd_bf16_d128_causal fwd_bf16_d128_n16k fwd_bf16_d128_n16k_mixed fwd_bf16_d128_transposed fwd_bf16_d256_transposed fwd_bf16_d64 fwd_bf16_d64_fp32mid fwd_bf16_d64_1head \
           fwd_bf16_d256 fwd_bf16_d256_mixed fwdbwd_bf16_d128 fwdbwd_bf16_d128_mixed fwdbwd_bf16_d128_causal fwdbwd_bf16_d128_transposed \
           fwdbwd_bf16_d128_transposed_ws fwdbwd_f16_d128_refmix fwdbwd_f32_d128 dkv_bf16_d128 dq_bf16_d128 fwdbwd_bf16_d256_mixed dq_bf16_d256 dkv_bf16_d256 fwdbwd_bf16_d320_mixed fwdbwd_bf16_d384_mixed; do
    timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_$w.json"
  done
  timeout 200 python bench.py --workload c1_cpu 2>/dev/null | tail -1 > "$OUT/bench_c1_cpu.json"
  timeout 300 python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_2ranks_one_gpu.json"
  python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("bench_")[1][:-5], d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
  timeout 200 python tools/bucket_perf.py --mixed 2>&1 | grep -v amdgpu.ids > "$OUT/bucket_perf_mixed.txt"; cat "$OUT/bucket_perf_mixed.txt"
  timeout 200 python tools/bucket_perf.py 2>&1 | grep -v amdgpu.ids > "$OUT/bucket_perf_fp32mid.txt"
  timeout 200 python tools/bucket_perf.py --mixed --fill zero 2>&1 | grep -v amdgpu.ids > "$OUT/bucket_perf_mixed_fill_zero.txt"; cat "$OUT/bucket_perf_mixed_fill_zero.txt"
  timeout 200 python tools/bucket_perf.py --mixed --causal 2>&1 | grep -v amdgpu.ids > "$OUT/bucket_perf_mixed_causal.txt"
  timeout 200 python tools/bucket_perf.py --mixed --heads 256 64 128 2>&1 | grep -v amdgpu.ids > "$OUT/bucket_perf_mixed_256heads.txt"
  timeout 200 python tools/bucket_perf.py --mixed --causal --heads 256 64 128 2>&1 | grep -v amdgpu.ids >> "$OUT/bucket_perf_mixed_256heads.txt"
  timeout 200 bash tools/zero_vs_random.sh "$OUT/zero_vs_random_d64" fwd_bf16_d64 > /dev/null 2>&1; cp "$OUT/zero_vs_random_d64/zero_vs_random.txt" "$OUT/d64_zero_vs_random.txt"; rm -rf "$OUT/zero_vs_random_d64"
  timeout 200 python tools/bucket_perf.py --mixed --transposed 2>&1 | grep -v amdgpu.ids > "$OUT/bucket_perf_mixed_transposed_no_workspace.txt"
  timeout 300 bash tools/zero_vs_random.sh "$OUT/zero_vs_random" > /dev/null 2>&1; cp "$OUT/zero_vs_random/zero_vs_random.txt" "$OUT/headline_zero_vs_random.txt"; rm -rf "$OUT/zero_vs_random"
  if [ -f metal_flash_attention_amd/libmfa_hip_dev.so ]; then
    for D in 256 160; do timeout 200 python tools/bwd5_prof.py --D $D 2>&1 | grep -v amdgpu.ids > "$OUT/bwd5_prof_d$D.txt"; done
  fi
  # the FP32 path (BASELINE config 3) kernel by kernel, its phase clocks (developer library) and the fp32 matrix-instruction probe
  (timeout 200 python tools/f32_perf.py; timeout 200 python tools/f32_perf.py --causal; timeout 200 python tools/f32_perf.py --fill zero 128) 2>&1 | grep -v amdgpu.ids > "$OUT/f32_perf.txt"; cat "$OUT/f32_perf.txt"
  if [ -f metal_flash_attention_amd/libmfa_hip_dev.so ]; then
    MFA_LIBRARY=metal_flash_attention_amd/libmfa_hip_dev.so MFA_F32_PROF=1 timeout 200 python tools/f32_perf.py 128 2>&1 | grep -v amdgpu.ids > "$OUT/f32_dq_phase_clocks.txt"
    MFA_LIBRARY=metal_flash_attention_amd/libmfa_hip_dev.so MFA_F32_GENERAL=1 timeout 200 python tools/f32_perf.py 2>&1 | grep -v amdgpu.ids > "$OUT/f32_perf_general_kernels.txt"
  fi
  [ -x tools/probe_f32_mfma.out ] && timeout 100 tools/probe_f32_mfma.out > "$OUT/probe_f32_mfma.txt" 2>&1
  timeout 200 python tools/time_single_head.py 2>&1 | grep -v amdgpu.ids > "$OUT/single_head.txt"
  timeout 200 python tools/time_single_head.py --mixed 2>&1 | grep -v amdgpu.ids > "$OUT/single_head_mixed.txt"
  timeout 300 python tools/fuzz_shapes.py 120 1 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_120_seed1.txt"; grep "random problems" "$OUT/fuzz_120_seed1.txt"
  timeout 400 python tools/fuzz_shapes.py 80 7 --workspace 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_workspace_80_seed7.txt"; grep "random problems" "$OUT/fuzz_workspace_80_seed7.txt"
  timeout 300 python tools/fuzz_shapes.py 120 5 --fp32 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_fp32_120_seed5.txt"; grep "random problems" "$OUT/fuzz_fp32_120_seed5.txt"
  timeout 300 python tools/fuzz_shapes.py 120 2 --transposed 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_transposed_120_seed2.txt"; grep "random problems" "$OUT/fuzz_transposed_120_seed2.txt"
  timeout 400 python tools/fuzz_shapes.py 90 3 --transposed --backward 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_transposed_backward_90_seed3.txt"; grep "random problems" "$OUT/fuzz_transposed_backward_90_seed3.txt"
fi

if [ -z "$FAST" ]; then
  # round 6: the 256 < D <= 384 forward on the 16-bit matrix cores, per-batch lengths on the persistent forwards, the vendor GEMM calibration,
  # the product schedule against the round-5 streams in one process (developer library)
  timeout 300 python tools/time_wide.py --backward 2>&1 | grep -v "amdgpu.ids" | grep -v "fp32 inputs  backward" > "$OUT/time_wide.txt"; cat "$OUT/time_wide.txt"
  timeout 300 bash tools/vendor_calib.sh "$OUT/vendor_calib" > /dev/null 2>&1; cp "$OUT/vendor_calib/vendor_calib.txt" "$OUT/vendor_gemm_calibration.txt"; rm -rf "$OUT/vendor_calib"
  if [ -f metal_flash_attention_amd/libmfa_hip_dev.so ]; then
    timeout 300 python tools/time_varlen.py --D 128 2>&1 | grep -v amdgpu.ids > "$OUT/time_varlen_d128.txt"; cat "$OUT/time_varlen_d128.txt"
    timeout 300 python tools/time_varlen.py --D 64 2>&1 | grep -v amdgpu.ids > "$OUT/time_varlen_d64.txt"; cat "$OUT/time_varlen_d64.txt"
    timeout 300 python tools/p4p_streams_ab.py --streams R5_BF16_FOLD_L16 --rounds 7 2>&1 | grep -v amdgpu.ids > "$OUT/p4p_product_vs_round5.txt"; cat "$OUT/p4p_product_vs_round5.txt"
    timeout 300 python tools/p4p_sprof.py --fill normal --stream BF16_FOLD_L16_FL1_SPROF 2>&1 | grep -v amdgpu.ids > "$OUT/p4p_phase_stamps.txt"
    timeout 300 python tools/p4p_sprof.py --fill zero --stream BF16_FOLD_L16_FL1_SPROF 2>&1 | grep -v amdgpu.ids >> "$OUT/p4p_phase_stamps.txt"
  fi
fi

#!/bin/bash
# Runs the GPU evidence a candidate library needs before it becomes the product library (inside one gpurun call):
#   make -C metal_flash_attention_amd/csrc TR_STREAMS=1          # here, on the CPU: ../libmfa_hip_tr.so
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/validate_candidate.sh'
# GPU suite (the staged-kernel tests included) against the candidate, the timing tools of the staged kernels through the developer
# library (they use its A/B knobs), the transposed fuzz with all three kernels.  Output: gpurun_out/candidate/.
cd "$(dirname "$0")/.." || exit 1
LIB=${LIB:-metal_flash_attention_amd/libmfa_hip_tr.so}
OUT=gpurun_out/candidate
mkdir -p "$OUT"
sha256sum "$LIB" > "$OUT/library.sha256"
MFA_LIBRARY=$LIB timeout 1100 python -m pytest tests -q -m gpu 2>&1 | tail -15 > "$OUT/pytest_gpu.txt"; tail -3 "$OUT/pytest_gpu.txt"
MFA_LIBRARY=$LIB timeout 300 python tools/fuzz_shapes.py 150 4 --transposed 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_transposed_150_seed4.txt"; tail -8 "$OUT/fuzz_transposed_150_seed4.txt"
MFA_LIBRARY=$LIB timeout 400 python tools/fuzz_shapes.py 90 5 --transposed --backward 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_transposed_backward_90_seed5.txt"; tail -8 "$OUT/fuzz_transposed_backward_90_seed5.txt"
for w in fwd_bf16_d256_transposed fwd_bf16_d128_transposed fwdbwd_bf16_d128_transposed; do
  MFA_LIBRARY=$LIB timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_$w.json"; cut -c1-200 "$OUT/bench_$w.json"
done
DEVLIB=metal_flash_attention_amd/libmfa_hip_dev.so
if [ -f "$DEVLIB" ]; then
  MFA_LIBRARY=$DEVLIB timeout 120 python tools/time_p5_tr.py 32 2>&1 | grep -v amdgpu.ids > "$OUT/time_p5_tr_32heads.txt"; cat "$OUT/time_p5_tr_32heads.txt"
  MFA_LIBRARY=$DEVLIB timeout 120 python tools/time_bwd_tr.py 2>&1 | grep -v amdgpu.ids > "$OUT/time_bwd_tr.txt"; cat "$OUT/time_bwd_tr.txt"
fi

#!/bin/bash
# round 4, GPU call 4: attn_dq16_p5 (role-split pairs x 64 rows, hand-placed) -- parity + A/B against the 32-row waves
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_call4
mkdir -p "$OUT"
sha256sum metal_flash_attention_amd/libmfa_hip.so > "$OUT/library.sha256"
timeout 300 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "test_backward_16bit_mfma" 2>&1 | tail -8 > "$OUT/pytest_backward16.txt"; tail -3 "$OUT/pytest_backward16.txt"
for args in "160 192 256" "--mixed 160 192 256" "--mixed --fill zero 160 192 256" "--mixed --causal 160 192 256"; do
  name=$(echo "bucket_perf $args" | tr -s ' -' '__')
  timeout 200 python tools/bucket_perf.py $args 2>&1 | grep -v amdgpu.ids > "$OUT/$name.txt"; cat "$OUT/$name.txt"
done
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > "$OUT/pytest_gpu.txt"; tail -5 "$OUT/pytest_gpu.txt"
cp gpurun_out/variant_coverage.json "$OUT/" 2>/dev/null
timeout 300 python tools/fuzz_shapes.py 80 22 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_80_seed22.txt"; grep "random problems" "$OUT/fuzz_80_seed22.txt"

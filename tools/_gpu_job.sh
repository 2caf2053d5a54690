cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_gpu.py -q -m gpu -k "backward_16bit_mfma or causal_bf16 or reference_low_precision or backward_16bit_multi_head" -x > gpurun_out/pytest_rs.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_rs.txt
tail -n 4 gpurun_out/pytest_rs.txt
for impl in rs w4; do echo "== $impl"; timeout 200 bash tools/time_kernels.sh fwdbwd_bf16_d128 MFA_DKV16_IMPL=$impl | grep dkv; done > gpurun_out/time_rs.txt 2>&1
cat gpurun_out/time_rs.txt

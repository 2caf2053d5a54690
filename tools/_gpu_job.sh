cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_torch_binding.py tests/test_c_abi.py -q -m gpu -x > gpurun_out/pytest_torch.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_torch.txt
tail -n 12 gpurun_out/pytest_torch.txt
timeout 600 bash tools/profile_pmc.sh gpurun_out/prof_fwdbwd --steps 3 --warmup 1 --no-cpu-baseline --workload fwdbwd_bf16_d128 > /dev/null 2>&1
rm -rf gpurun_out/prof_fwdbwd/stats gpurun_out/prof_fwdbwd/pmc1 gpurun_out/prof_fwdbwd/pmc2 gpurun_out/prof_fwdbwd/pmc3 gpurun_out/prof_fwdbwd/pmc4
head -12 gpurun_out/prof_fwdbwd/summary.txt
timeout 300 python bench.py --workload fwdbwd_bf16_d128 --no-cpu-baseline > gpurun_out/bench_fwdbwd_bf16.json 2> gpurun_out/bench_fwdbwd_bf16.err; cat gpurun_out/bench_fwdbwd_bf16.json
timeout 300 python bench.py --workload fwdbwd_bf16_d128_causal --no-cpu-baseline > gpurun_out/bench_fwdbwd_bf16_causal.json 2>/dev/null; cat gpurun_out/bench_fwdbwd_bf16_causal.json

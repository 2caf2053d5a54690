cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "square_fp32 or rectangular or causal_fp32 or transpose or multi_head or config3 or block_sparse or variable" > gpurun_out/pytest_f32.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_f32.txt; tail -n 4 gpurun_out/pytest_f32.txt
timeout 300 python bench.py --workload fwdbwd_f32_d128 --no-cpu-baseline > gpurun_out/final/bench_fwdbwd_f32_d128.json 2>/dev/null; cut -c1-200 gpurun_out/final/bench_fwdbwd_f32_d128.json
timeout 600 bash tools/profile_pmc.sh gpurun_out/final/prof_f32 --steps 3 --warmup 1 --no-cpu-baseline --workload fwdbwd_f32_d128 > /dev/null 2>&1
rm -rf gpurun_out/final/prof_f32/stats gpurun_out/final/prof_f32/pmc1 gpurun_out/final/prof_f32/pmc2 gpurun_out/final/prof_f32/pmc3 gpurun_out/final/prof_f32/pmc4; head -12 gpurun_out/final/prof_f32/summary.txt

cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/final/pytest_gpu.txt
tail -n 4 gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final/smoke.txt 2>&1; tail -n 1 gpurun_out/final/smoke.txt
timeout 300 python bench.py --workload fwdbwd_f32_d128 --no-cpu-baseline > gpurun_out/final/bench_fwdbwd_f32_d128.json 2>/dev/null; cut -c1-220 gpurun_out/final/bench_fwdbwd_f32_d128.json
timeout 400 python bench.py > gpurun_out/final/bench_headline.json 2>/dev/null; cut -c1-200 gpurun_out/final/bench_headline.json

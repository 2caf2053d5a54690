cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py -q -m gpu -x > gpurun_out/pytest_gemm.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_gemm.txt
tail -n 12 gpurun_out/pytest_gemm.txt
timeout 300 python tools/bench_gemm.py --sizes 4096 --dtypes bf16 --odd > gpurun_out/bench_gemm3.txt 2>&1; cat gpurun_out/bench_gemm3.txt

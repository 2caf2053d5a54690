cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py -q -m gpu -x > gpurun_out/pytest_gemm.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_gemm.txt
tail -n 25 gpurun_out/pytest_gemm.txt

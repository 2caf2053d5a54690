cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_c_abi.py -q -m gpu > gpurun_out/pytest_cabi.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_cabi.txt; tail -n 6 gpurun_out/pytest_cabi.txt

cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_torch_binding.py -q -m gpu -x > gpurun_out/pytest_torch.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_torch.txt
tail -n 12 gpurun_out/pytest_torch.txt

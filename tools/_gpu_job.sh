cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "block_sparse" > gpurun_out/pytest_bs.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_bs.txt
tail -n 25 gpurun_out/pytest_bs.txt

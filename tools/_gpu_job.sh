cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "block_sparse or backward_16bit_mfma or causal_bf16" > gpurun_out/pytest_bs.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_bs.txt
tail -n 8 gpurun_out/pytest_bs.txt
timeout 300 python tools/time_sparse.py > gpurun_out/time_sparse.txt 2>&1; cat gpurun_out/time_sparse.txt

cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 bash tools/time_kernels.sh fwdbwd_f32_d128 > gpurun_out/time_f32.txt 2>&1
cat gpurun_out/time_f32.txt
timeout 300 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "block_sparse" > gpurun_out/pytest_bs.txt 2>&1; tail -n 3 gpurun_out/pytest_bs.txt

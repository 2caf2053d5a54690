set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/ab_fwd16.py --impls v3:0,v4:0,v4:1,v4:2,v4:4,v4:8,v4:16 --N 4096 --D 128 --heads 256 > gpurun_out/ab_v4_d128.txt 2>&1; echo "rc=$?" >> gpurun_out/ab_v4_d128.txt
timeout 300 python tools/ab_fwd16.py --impls v3:0,v4:0,v4:1,v4:2,v4:8,v4:16 --N 4096 --D 64 --heads 256 > gpurun_out/ab_v4_d64.txt 2>&1; echo "rc=$?" >> gpurun_out/ab_v4_d64.txt
timeout 600 python -m pytest tests/test_attention_gpu.py -q -m gpu -k "role_alternating or forced_rescale" -x > gpurun_out/pytest_v4.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_v4.txt
timeout 300 python -m pytest tests/test_c_abi.py -q -m gpu > gpurun_out/pytest_cabi.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_cabi.txt
tail -5 gpurun_out/ab_v4_d128.txt gpurun_out/pytest_v4.txt gpurun_out/pytest_cabi.txt

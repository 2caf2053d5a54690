cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/ab_fwd16.py --impls v3:0,v3:41,v3:50,v3:51,v3:52,v3:14 --N 4096 --D 128 --heads 256 > gpurun_out/ab_abl_d128.txt 2>&1; echo "rc=$?" >> gpurun_out/ab_abl_d128.txt
timeout 600 python -m pytest tests/test_attention_gpu.py -q -m gpu -k "role_alternating or forced_rescale" -x > gpurun_out/pytest_vd.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_vd.txt
cat gpurun_out/ab_abl_d128.txt; tail -n 5 gpurun_out/pytest_vd.txt

cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "forward or rescale or split or causal or block_sparse or low_precision or variable" > gpurun_out/pytest_fwd.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_fwd.txt
tail -n 6 gpurun_out/pytest_fwd.txt
timeout 300 python tools/ab_fwd16.py --impls v3:0,v3:41,v3:50,v3:51,v3:52 --N 4096 --D 128 --heads 256 > gpurun_out/ab_after_cleanup.txt 2>&1; cat gpurun_out/ab_after_cleanup.txt

cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "variable_sequence or block_sparse" > gpurun_out/pytest_var.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_var.txt
tail -n 4 gpurun_out/pytest_var.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.txt 2>&1; tail -n 4 gpurun_out/smoke.txt

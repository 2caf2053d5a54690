cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/final/pytest_gpu.txt
tail -n 4 gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final/smoke.txt 2>&1; tail -n 2 gpurun_out/final/smoke.txt
timeout 400 python bench.py > gpurun_out/final/bench_headline.json 2> gpurun_out/final/bench_headline.err; cut -c1-400 gpurun_out/final/bench_headline.json
for w in fwd_bf16_d64 fwd_bf16_d64_1head fwd_bf16_d256 fwdbwd_f32_d128 fwdbwd_bf16_d128 fwd_bf16_d128_causal fwdbwd_bf16_d128_causal fwd_bf16_d128_n16k; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/final/bench_$w.json 2>/dev/null; python - <<PY
import json
d=json.load(open("gpurun_out/final/bench_$w.json"))
print("$w", d["value"], d["ms_per_step"], d["mfma_tflops"], d["roofline"]["frac"], d["config"]["kernel_variants"])
PY
done
# multi-process control path on one GPU (gloo control plane; both ranks share GPU 0)
MFA_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_2proc.json 2> gpurun_out/final/bench_2proc.err; cut -c1-300 gpurun_out/final/bench_2proc.json
timeout 600 bash tools/profile_pmc.sh gpurun_out/final/prof_headline --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 bash tools/profile_pmc.sh gpurun_out/final/prof_fwdbwd --steps 3 --warmup 1 --no-cpu-baseline --workload fwdbwd_bf16_d128 > /dev/null 2>&1
timeout 600 bash tools/profile_pmc.sh gpurun_out/final/prof_f32 --steps 3 --warmup 1 --no-cpu-baseline --workload fwdbwd_f32_d128 > /dev/null 2>&1
for d in prof_headline prof_fwdbwd prof_f32; do rm -rf gpurun_out/final/$d/stats gpurun_out/final/$d/pmc1 gpurun_out/final/$d/pmc2 gpurun_out/final/$d/pmc3 gpurun_out/final/$d/pmc4; head -8 gpurun_out/final/$d/summary.txt; done

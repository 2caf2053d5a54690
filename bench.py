#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): GINSTR/s + achieved MFMA TFLOPS vs peak, forward, N=4096, D=128, bf16.
GINSTR = (2D+5) N^2 per head per dispatch (reference: README.md:108-124,
Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:742-756).

A "step" is one forward dispatch over this rank's shard of batch x head (synthetic Q/K/V already
resident in HBM).  The reference benchmarks a single head and raises N until the GPU is full
(SquareAttentionTest.swift:159-165); a single N=4096 head is 16 workgroups on a 256-CU chip, so
the throughput number is taken with a batch x head grid (B=8, H=32 per GPU), the sharding axis of
BASELINE config 5.  Multi-GPU: one process per GPU, heads sharded, NO data-path collective and no RCCL
(attention heads are independent); torch.distributed's gloo backend carries the one barrier and the
max-over-ranks of the elapsed time.  `python bench.py --gpus N` starts its own N ranks when it was not launched by
torch.distributed.run (ranks beyond the device count share devices round-robin, which is how the N > 1 path is
exercised on a 1-GPU box).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fwd_bf16_d128|...]

Workloads: fwd_bf16_d128 = the headline (BASELINE config 4's dtype at the metric's N and D); fwd_bf16_d128_n16k =
BASELINE config 5's per-GPU shard (32 heads of N=16384 per GPU: `--gpus 8 --workload fwd_bf16_d128_n16k` is the
B=8 H=32 job); c1_cpu = BASELINE config 1 (N=128, D=64, fp32, the CPU oracle only: one thread and all cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # /opt/skills/guides/MI355X_MICROARCH.md
# Calibration of the power-limited ceiling, measured (not assumed): the vendor GEMM (hipBLASLt bf16, n = 8192, A B^T) on N(0,1)
# operands on an MI355X of this pool under the same counters as the attention kernel -- TF/s un-profiled, matrix-pipe busy fraction
# and effective clock from the PMC pass (tools/vendor_calib.sh -> profiles/r06_vendor_gemm_calibration.txt).  A body that is almost
# nothing but matrix instructions sustains 0.64 of the spec peak on random data on this chip; reported next to the spec fraction.
# (Round 5 printed a "sustained peak" of 1560 TF here that the vendor GEMM itself exceeds: removed.)
VENDOR_GEMM_CALIBRATION = {"bf16": {"tflops_random": 1594.1, "tflops_zero": 2259.9, "mfma_busy_random": 0.865, "clock_ghz_random": 1.707,
                                    "source": "profiles/r06_vendor_gemm_calibration.txt"}}
VENDOR_GEMM_CALIBRATION["f16"] = VENDOR_GEMM_CALIBRATION["bf16"]

# HBM bytes per launch come from rocprofv3 PMC passes (it cannot run inside this process): tools/profile_pmc.sh writes
# profiles/traffic.json with the SHA-256 of the libmfa_hip.so it profiled; the bench line carries a number only when
# that hash is the hash of the library loaded NOW and the kernel variant matches -- otherwise null (never stale).
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")
EADDRINUSE_EXIT = 98   # a self-spawned rank could not join the rendezvous: spawn_ranks starts the group again on another port


def measured_traffic(workload, variant):
    import hashlib
    try:
        with open(TRAFFIC_JSON) as f:
            table = json.load(f)
        from metal_flash_attention_amd import _abi
        with open(_abi.library_path(), "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()
    except Exception:  # noqa: BLE001
        return None, None, None
    for row in table.get("entries", []):
        if row.get("workload") == workload and row.get("variant") == variant and row.get("lib_sha256") == digest:
            return row.get("bytes_per_launch"), row.get("source"), row
    return None, None, None


WORKLOADS = {
    # name: (kernel types, N, D, dtype, batch, heads)
    # headline: the reference's MIXED-PRECISION mode (lowPrecisionInputs + lowPrecisionIntermediates, the mode its own
    # headline numbers are quoted in, README.md:15 / BASELINE.md) with BF16 storage
    "fwd_bf16_d128": dict(N=4096, D=128, dtype="bf16", batch=8, heads=32, types=("forward",), low_mid=True),
    # same shape with the attention matrix kept in FP32 registers (lowPrecisionIntermediates = false): the scale is applied
    # in fp32 per score instead of being folded into Q
    "fwd_bf16_d128_fp32mid": dict(N=4096, D=128, dtype="bf16", batch=8, heads=32, types=("forward",)),
    # config 2 (batched / as written): like the headline in the reference's mixed-precision mode (round 5: attn_fwd16_p6 serves it);
    # _fp32mid = lowPrecisionInputs only (the eight-wave kernel attn_fwd16_v3)
    "fwd_bf16_d64": dict(N=4096, D=64, dtype="bf16", batch=8, heads=32, types=("forward",), low_mid=True),
    "fwd_bf16_d64_fp32mid": dict(N=4096, D=64, dtype="bf16", batch=8, heads=32, types=("forward",)),
    "fwd_bf16_d64_1head": dict(N=4096, D=64, dtype="bf16", batch=1, heads=1, types=("forward",), low_mid=True),
    "fwd_bf16_d256": dict(N=8192, D=256, dtype="bf16", batch=2, heads=16, types=("forward",)),   # config 4, batched
    "fwd_bf16_d256_mixed": dict(N=8192, D=256, dtype="bf16", batch=2, heads=16, types=("forward",), low_mid=True),
    "fwdbwd_f32_d128": dict(N=4096, D=128, dtype="f32", batch=2, heads=16,
                            types=("forward", "backwardQuery", "backwardKeyValue")),             # config 3, batched
    "fwdbwd_bf16_d128": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16,
                             types=("forward", "backwardQuery", "backwardKeyValue")),
    # one backward kernel timed alone (forward and backwardQuery run once, untimed, so that L and D hold real values)
    "dkv_bf16_d128": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True, timed=("backwardKeyValue",),
                          types=("forward", "backwardQuery", "backwardKeyValue")),
    "dkv_bf16_d128_fp32mid": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, timed=("backwardKeyValue",),
                                  types=("forward", "backwardQuery", "backwardKeyValue")),
    "dkv_bf16_d128_causal": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True, causal=True,
                                 timed=("backwardKeyValue",), types=("forward", "backwardQuery", "backwardKeyValue")),
    "dq_bf16_d128": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True, timed=("backwardQuery",),
                         types=("forward", "backwardQuery", "backwardKeyValue")),
    "dq_bf16_d128_fp32mid": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, timed=("backwardQuery",),
                                 types=("forward", "backwardQuery", "backwardKeyValue")),
    "dq_bf16_d128_causal": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True, causal=True,
                                timed=("backwardQuery",), types=("forward", "backwardQuery", "backwardKeyValue")),
    # the reference's own low-precision mix (+Precisions.swift:13-17): FP16 Q, K, V with BF16 dO, FP16 L, BF16 D
    "fwdbwd_f16_d128_refmix": dict(N=4096, D=128, dtype="f16", batch=4, heads=16, low_mid=True,
                                   types=("forward", "backwardQuery", "backwardKeyValue")),
    "fwdbwd_bf16_d128_mixed": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True,
                                   types=("forward", "backwardQuery", "backwardKeyValue")),
    # the headline shape with Q, K, V, O stored transposed ([D][N]): read and written in place by the matrix-core kernel's
    # transposed code object (round 3; round 2: re-layout passes through a workspace)
    "fwd_bf16_d128_transposed": dict(N=4096, D=128, dtype="bf16", batch=8, heads=32, types=("forward",), low_mid=True,
                                     tr=(True, True, True, True)),
    # BASELINE config 4's shape with Q, K, V, O transposed (the hand-placed stream attn_fwd16_p5_tr reads K^T / V^T in place); and
    # all three kernels on transposed operands WITHOUT a workspace (round 4: the in-place backward kernels attn_dq16_p4_tr /
    # attn_dkv16_p4_tr); `_ws` = the same with a caller workspace, i.e. the re-layout path of round 2
    "fwd_bf16_d256_transposed": dict(N=8192, D=256, dtype="bf16", batch=2, heads=16, types=("forward",), low_mid=True,
                                     tr=(True, True, True, True)),
    "fwdbwd_bf16_d128_transposed": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True, tr=(True, True, True, True),
                                        types=("forward", "backwardQuery", "backwardKeyValue")),
    "fwdbwd_bf16_d128_transposed_ws": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, low_mid=True, tr=(True, True, True, True),
                                           relayout_workspace=True, types=("forward", "backwardQuery", "backwardKeyValue")),
    "fwd_bf16_d128_causal": dict(N=4096, D=128, dtype="bf16", batch=8, heads=32, types=("forward",), causal=True),
    "fwdbwd_bf16_d128_causal": dict(N=4096, D=128, dtype="bf16", batch=4, heads=16, causal=True,
                                    types=("forward", "backwardQuery", "backwardKeyValue")),
    # BASELINE config 4's head dimension, all three kernels (round 4: the role-split backward streams attn_dq16_p5 / attn_dkv16_p5)
    "fwdbwd_bf16_d256_mixed": dict(N=4096, D=256, dtype="bf16", batch=4, heads=16, low_mid=True,
                                   types=("forward", "backwardQuery", "backwardKeyValue")),
    "dq_bf16_d256": dict(N=4096, D=256, dtype="bf16", batch=4, heads=16, low_mid=True, timed=("backwardQuery",),
                         types=("forward", "backwardQuery", "backwardKeyValue")),
    "dkv_bf16_d256": dict(N=4096, D=256, dtype="bf16", batch=4, heads=16, low_mid=True, timed=("backwardKeyValue",),
                          types=("forward", "backwardQuery", "backwardKeyValue")),
    # round 6: the head blocks above 256 on the 16-bit matrix cores (attn_fwd16_wide, attn_dq16 with 32-key tiles, attn_dkv16_wide)
    "fwdbwd_bf16_d384_mixed": dict(N=4096, D=384, dtype="bf16", batch=2, heads=16, low_mid=True,
                                   types=("forward", "backwardQuery", "backwardKeyValue")),
    "fwdbwd_bf16_d320_mixed": dict(N=4096, D=320, dtype="bf16", batch=2, heads=16, low_mid=True,
                                   types=("forward", "backwardQuery", "backwardKeyValue")),
    "fwd_bf16_d128_n16k": dict(N=16384, D=128, dtype="bf16", batch=1, heads=32, types=("forward",)),  # config 5 shard
    # the same shard in the reference's mixed-precision mode, like the headline (FP16 L: at N = 16384 |L| is still below 16,
    # the FP16 resolution the reference's own L tolerance of 7e-3 assumes -- tests/test_attention_gpu.py holds it at full size)
    "fwd_bf16_d128_n16k_mixed": dict(N=16384, D=128, dtype="bf16", batch=1, heads=32, types=("forward",), low_mid=True),
    "c1_cpu": dict(N=128, D=64, dtype="f32", batch=1, heads=1, types=("forward",)),                   # config 1, CPU only
}
OPS_PER_N2 = {"forward": lambda D: 2 * D + 5, "backwardQuery": lambda D: 3 * D + 5,
              "backwardKeyValue": lambda D: 4 * D + 5}       # README.md:108-124
FLOPS_PER_N2 = {"forward": lambda D: 4 * D, "backwardQuery": lambda D: 6 * D,
                "backwardKeyValue": lambda D: 8 * D}         # MFMA flops: 2, 3, 4 GEMMs of 2 N^2 D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fwd_bf16_d128", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-seconds", type=float, default=0.25,
                    help="untimed run of the same step before the W warmup steps, until the power management has settled "
                         "(the first ~10 launches after an idle period run ~9 %% slower: profiles/r02_bench_spinup.txt)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--fill", default="normal", choices=("normal", "zero"),
                    help="zero: all-zero Q/K/V/dO -- NOT a valid measurement, only the clock experiment of tools/zero_vs_random.sh "
                         "(MI355X_MICROARCH.md DVFS note: operand toggling sets the power-limited clock); the line says so in `data`")
    args = ap.parse_args()

    if args.workload == "c1_cpu":
        print(json.dumps(c1_cpu_line(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % ndev)   # ranks share a device only when there are fewer devices than ranks
    dist = None
    ctl_device = "cpu"
    if world > 1:
        # control plane only (one barrier + max of one scalar), always gloo: attention heads need no data-path
        # collective and this path must not depend on RCCL (north_star: "no RCCL")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
        except Exception as exc:   # noqa: BLE001 -- the rendezvous port was taken between spawn_ranks' probe and rank 0's listen
            # only an address-in-use condition asks spawn_ranks for a new port (it looks at rank 0's exit code); timeouts and
            # real gloo errors keep their own exception and exit code on every rank
            msg = str(exc).lower()
            if os.environ.get("MFA_BENCH_SPAWNED") and ("eaddrinuse" in msg or "address already in use" in msg):
                raise SystemExit(EADDRINUSE_EXIT)
            raise

    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    from metal_flash_attention_amd.sharding import max_over_ranks, shard_range

    w = WORKLOADS[args.workload]
    N, D, B, H = w["N"], w["D"], w["batch"], w["heads"]
    # weak scaling: the job has B*H heads per GPU; this rank owns a contiguous range of the
    # flattened batch x head axis (no data-path collective, heads are independent)
    unit_begin, unit_end = shard_range(B * H * world, world, rank)
    assert unit_end - unit_begin == B * H
    low = w["dtype"] != "f32"
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = low
    desc.lowPrecisionIntermediates = bool(w.get("low_mid", False))
    desc.lowPrecisionInputType = P.BF16 if w["dtype"] == "bf16" else P.FP16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = tuple(w.get("tr", (False, False, False, False)))
    types = [AttentionKernelType[t] for t in w["types"]]
    kernels = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in types}

    # synthetic inputs: i.i.d. N(0,1) (Network.swift:96-129 distribution), this rank's shard of heads
    tdtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[w["dtype"]]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + unit_begin)
    shape = (B, H, N, D)
    bufs = {}
    for op in (Op.Q, Op.K, Op.V):
        bufs[op] = torch.randn(shape, generator=gen, device="cuda", dtype=torch.float32).to(tdtype)
    mem = desc.memoryPrecisions   # L is FP16 and D BF16 in the reference's mixed-precision mode (+Precisions.swift:82-83)
    tprec = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    bufs[Op.O] = torch.empty(shape, device="cuda", dtype=tprec[mem[Op.O]])
    bufs[Op.L] = torch.empty((B, H, N), device="cuda", dtype=tprec[mem[Op.L]])
    if args.fill == "zero":
        for op in (Op.Q, Op.K, Op.V):
            bufs[op].zero_()
    backward = len(types) > 1
    if backward:
        # the reference stores dO as BF16 in low-precision mode (+Precisions.swift:17)
        bufs[Op.dO] = torch.randn(shape, generator=gen, device="cuda", dtype=torch.float32).to(
            torch.bfloat16 if low else torch.float32)
        if args.fill == "zero":
            bufs[Op.dO].zero_()
        bufs[Op.D] = torch.empty((B, H, N), device="cuda", dtype=tprec[mem[Op.D]])
        for op in (Op.dQ, Op.dK, Op.dV):
            bufs[op] = torch.empty(shape, device="cuda", dtype=torch.float32)
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
    bs = {op: v * H for op, v in hs.items()}
    stream = torch.cuda.current_stream().cuda_stream
    # caller-owned scratch: lets a launch with too few row blocks (single head) run column-parallel
    relayout = bool(w.get("relayout_workspace"))   # transposed operands through the re-layout scratch instead of in place
    ws_bytes = max(kernels[t].workspaceSize(row=N, column=N, heads=H, batches=B) if (t.name == "forward" or relayout) else 0
                   for t in types)
    workspace = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") if ws_bytes else None

    def step():
        for t in timed:
            kernels[t].dispatch(bufs, row=N, column=N, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                                stream=stream, workspace=workspace if (t.name == "forward" or relayout) else None,
                                causal=bool(w.get("causal", False)))

    timed = types
    step()                            # every kernel once: L, D (and O) hold real values for a kernel timed alone
    timed = [t for t in types if t.name in w.get("timed", w["types"])]
    types = timed                     # rates and rooflines below count the timed kernels only

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # cold start: the first K launches after the buffers were filled (one functional launch before them, no spin-up, no warmup)
    torch.cuda.synchronize()
    cev0, cev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cev0.record()
    for _ in range(args.steps):
        step()
    cev1.record()
    torch.cuda.synchronize()
    cold_ms_per_step = cev0.elapsed_time(cev1) / args.steps

    # spin-up: the chip leaves its idle clocks only under load (launches 6-10 after idle: 1.93 ms, steady state: 1.77 ms);
    # the reference's own benchmark takes the best of several multi-dispatch trials for the same reason
    # (SquareAttentionTest.swift:159-212).  Untimed, reported in config.spinup_steps.
    spinup_steps = 0
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_seconds:
        step()
        spinup_steps += 1
        if spinup_steps % 8 == 0:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # HIP events on the stream the kernels are launched on
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    device_ms = ev0.elapsed_time(ev1)
    elapsed = max_over_ranks(elapsed, dist, device=ctl_device)
    if dist is not None:
        dist.barrier()

    heads_total = B * H * world
    # causal workloads do (N+1)/2N of the work; GINSTR and flops count what is actually computed
    work = (N + 1) / (2.0 * N) if w.get("causal") else 1.0
    ops_step = sum(OPS_PER_N2[t.name](D) for t in types) * N * N * heads_total * work
    ginstrs = ops_step * args.steps / elapsed / 1e9
    # roofline of the dominant kernel, per launch on this rank, from the HIP-event time
    flops_launch_rank = sum(FLOPS_PER_N2[t.name](D) for t in types) * N * N * B * H * work
    launch_ms = device_ms / args.steps
    achieved_tflops = flops_launch_rank / (launch_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[w["dtype"]]

    traffic_bytes, traffic_source, traffic_row = measured_traffic(args.workload, kernels[types[0]].variant)
    # derived from the same PMC passes (profiles/traffic.json, tied to the library's SHA-256): HBM GB/s = measured bytes per launch
    # over THIS run's launch time; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs of the
    # profiled launches (a cycle ratio: it does not depend on the clock the box grants)
    hbm_gbs = round(traffic_bytes / (launch_ms * 1e-3) / 1e9, 1) if traffic_bytes else None
    mfma_busy = None
    if traffic_row and traffic_row.get("mfma_busy_cycles") and traffic_row.get("gui_active"):
        mfma_busy = round(traffic_row["mfma_busy_cycles"] / 1024.0 / (traffic_row["gui_active"] / 8.0), 4)

    # The headline is timed in the reference's mixed-precision mode; SURVEY.md 8(d) words the metric "bf16 Q/K/V, fp32 O + L",
    # i.e. lowPrecisionIntermediates = false (scale applied in fp32 per score, L stored in FP32): the same shape, same inputs,
    # same process, timed right behind it and reported in config.fp32_intermediates
    other_mode = None
    if args.workload in ("fwd_bf16_d128", "fwd_bf16_d64"):
        desc2 = AttentionDescriptor()
        desc2.lowPrecisionInputs, desc2.lowPrecisionIntermediates = True, False
        desc2.lowPrecisionInputType = P.BF16
        desc2.matrixDimensions = (N, N, D)
        desc2.transposeState = (False, False, False, False)
        k2 = AttentionKernel(desc2.kernelDescriptor(AttentionKernelType.forward))
        bufs2 = dict(bufs)
        bufs2[Op.O] = torch.empty(shape, device="cuda", dtype=torch.float32)
        bufs2[Op.L] = torch.empty((B, H, N), device="cuda", dtype=torch.float32)

        def step2():
            k2.dispatch(bufs2, row=N, column=N, heads=H, batches=B, headStrides=hs, batchStrides=bs, stream=stream, workspace=workspace)

        t_spin2 = time.perf_counter()          # the same untimed spin-up as the headline mode got (clocks settle under load)
        n2 = 0
        while time.perf_counter() - t_spin2 < args.spinup_seconds:
            step2()
            n2 += 1
            if n2 % 8 == 0:
                torch.cuda.synchronize()
        for _ in range(max(args.warmup, 3)):
            step2()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step2()
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / args.steps
        tf2 = FLOPS_PER_N2["forward"](D) * N * N * B * H / (ms2 * 1e-3) / 1e12
        other_mode = {"precision_mode": "lowPrecisionInputs only: scale applied in fp32 per score, FP32 L (SURVEY.md 8(d) wording)",
                      "variant": k2.variant, "ms_per_step": round(ms2, 4), "tflops": round(tf2, 2), "frac": round(tf2 / peak, 4),
                      "ginstrs_per_gpu": round(OPS_PER_N2["forward"](D) * N * N * B * H / (ms2 * 1e-3) / 1e9, 2),
                      "max_abs_diff_O_vs_mixed_mode": float((bufs2[Op.O] - bufs[Op.O]).abs().max().item())}
    out = {
        "metric": ("GINSTR/s forward attention N=4096 D=128 bf16 (mixed-precision mode: lowPrecisionInputs + "
                   "lowPrecisionIntermediates; the fp32-intermediates mode is config.fp32_intermediates)")
        if args.workload == "fwd_bf16_d128" else f"GINSTR/s {args.workload}",
        "value": round(ginstrs, 2),
        "unit": "GINSTR/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": w["dtype"],
        "data": "synthetic" if args.fill == "normal" else "all-zero operands: clock experiment only, not a measurement of the metric",
        "config": {"workload": f"attention {'+'.join(w.get('timed', w['types']))} N={N} D={D} {w['dtype']} Q/K/V, fp32 O, {mem[Op.L].name} L; "
                               f"B={B} H={H} heads per GPU, batch x head sharded across GPUs, no collectives",
                   "precision_mode": ("mixed: lowPrecisionInputs + lowPrecisionIntermediates (the reference's mixed-precision "
                                      "benchmark mode, README.md:15)" if w.get("low_mid") else
                                      ("lowPrecisionInputs only (attention matrix in FP32 registers)" if low else "FP32")),
                   "kernel_variants": [kernels[t].variant for t in types],
                   "split_kv_workspace_bytes": ws_bytes,
                   "control_plane": "gloo" if world > 1 else "none", "devices_visible": ndev,
                   "spinup_steps": spinup_steps,
                   "cold_start_ms_per_step": round(cold_ms_per_step, 4),
                   # (ranks SHARING a device, i.e. fewer devices than ranks: their launches serialize on it and a per-rank fraction of the
                   # roof means nothing -- reported only with one device per rank)
                   "per_gpu_roofline_frac": round(achieved_tflops / peak, 4) if ndev >= world else None,
                   "ranks_share_a_device": bool(ndev < world)},
        "mfma_tflops": round(achieved_tflops * world, 2),
        "roofline": {"bound": "mfma", "achieved": round(achieved_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved_tflops / peak, 4),
                     "vendor_gemm_same_chip": VENDOR_GEMM_CALIBRATION.get(w["dtype"]),
                     "frac_of_vendor_gemm": (round(achieved_tflops / VENDOR_GEMM_CALIBRATION[w["dtype"]]["tflops_random"], 4)
                                             if w["dtype"] in VENDOR_GEMM_CALIBRATION and args.fill == "normal" else None),
                     "traffic": traffic_bytes, "traffic_unit": "bytes/launch (HBM, PMC)", "traffic_source": traffic_source,
                     "hbm_gbs": hbm_gbs, "hbm_peak_gbs": 8000.0, "mfma_busy": mfma_busy,
                     "algorithmic_bytes": (3 * N * D * (2 if low else 4) + N * D * bufs[Op.O].element_size()
                                           + N * bufs[Op.L].element_size()) * B * H if not backward else None,
                     "kernel": "+".join(kernels[t].variant for t in types), "launch_ms": round(launch_ms, 4),
                     # what the timed launches ran (mfa_attention_kernel_launch_form): e.g. the persistent form attn_fwd16_p4p of the
                     # D <= 128 forward object -- the kernel name rocprofv3 reports for this command
                     "launch_form": [kernels[t].launchForm(bufs, row=N, column=N, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                                                           workspace=workspace if (t.name == "forward" or relayout) else None,
                                                           causal=bool(w.get("causal", False))) for t in types]},
    }

    if other_mode is not None:
        out["config"]["fp32_intermediates"] = other_mode
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(np, torch, w, bufs, Op, args.cpu_seconds, backward)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks of this script (one per GPU; round-robin
    over the visible devices when there are fewer), rendezvous on 127.0.0.1 over gloo.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    # A free port found by bind-then-release can be taken by another process before rank 0 listens on it (busy node):
    # rank 0 then exits with EADDRINUSE_EXIT before any GPU work, and the whole group is started again on a new port.
    code = 1
    for attempt in range(8):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), MFA_BENCH_SPAWNED="1",
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        codes = [p.wait() for p in procs]
        code = max(abs(c) for c in codes)
        if codes[0] != EADDRINUSE_EXIT:   # rank 0 owns the listening socket: only its address-in-use exit means "try another port"
            break
    return code


def c1_cpu_line(args):
    """BASELINE config 1: forward, one head, N=128, D=64, fp32, through the naive CPU reference -- here its C restatement
    (oracle/network.c).  No GPU is touched.  `value` = all host threads; the single-thread rate (the analogue of the
    unparallelised Swift loop) is reported next to it, both checked against the committed fixture."""
    import numpy as np
    from oracle import Network, NetworkDescriptor, max_threads
    w = WORKLOADS["c1_cpu"]
    N, D = w["N"], w["D"]
    ops = OPS_PER_N2["forward"](D) * N * N
    golden = np.load(os.path.join(ROOT, "tests", "golden", "network_golden.npz"))

    def rate(threads):
        net = Network(NetworkDescriptor(N, N, D), seed=10, threads=threads)
        res = net.run(backward=False)
        assert np.array_equal(res["O"], golden["s10_O"]) and np.array_equal(res["L"], golden["s10_L"])
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            net.run(backward=False)
            reps += 1
        return ops * reps / (time.perf_counter() - t0) / 1e9

    one, many = rate(1), rate(0)
    return {"metric": "GINSTR/s c1_cpu", "value": round(many, 3), "unit": "GINSTR/s", "n_gpus": 0, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ops / many / 1e6, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "attention forward N=128 D=64 fp32, one head, CPU oracle only (BASELINE config 1)"},
            "cpu_baseline": {"value": round(many, 3), "unit": "GINSTR/s", "cores": max_threads(), "kind": "port",
                             "single_thread_value": round(one, 3),
                             "sample": "the whole workload, repeated for 1 s per thread count; O and L equal the committed "
                                       "fixture tests/golden/network_golden.npz bit for bit"}}


def cpu_baseline(np, torch, w, bufs, Op, target_seconds, backward):
    """Times the oracle (C restatement of the reference's naive CPU Network) on the host cores for a
    bounded sample of the SAME workload -- whole heads taken from the GPU buffers -- and checks the
    GPU result of those heads against it."""
    from oracle import Network, NetworkDescriptor, max_threads

    N, D = w["N"], w["D"]
    ops_head = (OPS_PER_N2["forward"](D) + (OPS_PER_N2["backwardQuery"](D) + OPS_PER_N2["backwardKeyValue"](D)
                                            if backward else 0)) * N * N * ((N + 1) / (2.0 * N) if w.get("causal") else 1.0)
    threads = max_threads()
    heads_done, cpu_time, max_err, max_err_l = 0, 0.0, 0.0, 0.0
    flat = {op: t.reshape(-1, *t.shape[2:]) for op, t in bufs.items()}
    nheads = flat[Op.Q].shape[0]
    while heads_done < nheads and heads_done < 64:
        net = Network(NetworkDescriptor(N, N, D), seed=0)
        net.Q = flat[Op.Q][heads_done].float().cpu().numpy()
        net.K = flat[Op.K][heads_done].float().cpu().numpy()
        net.V = flat[Op.V][heads_done].float().cpu().numpy()
        if backward:
            net.dO = flat[Op.dO][heads_done].float().cpu().numpy()
        t0 = time.perf_counter()
        ref = net.run(backward=backward, causal=bool(w.get("causal", False)))
        cpu_time += time.perf_counter() - t0
        got = flat[Op.O][heads_done].float().cpu().numpy()
        max_err = max(max_err, float(np.abs(got - ref["O"]).max()))
        # L is stored in base-2 units (m + log2 l, +Caching.swift:373-377); the oracle's is natural-log (Network.swift:181-203)
        got_l = flat[Op.L][heads_done].float().cpu().numpy() / np.float32(1.44269504089)
        max_err_l = max(max_err_l, float(np.abs(got_l - ref["L"]).max()))
        heads_done += 1
        if cpu_time >= target_seconds:
            break
    return {"value": round(ops_head * heads_done / cpu_time / 1e9, 3), "unit": "GINSTR/s", "cores": threads,
            "kind": "port",
            "sample": f"{heads_done} of {nheads} heads of the same workload (same Q/K/V as the GPU run), "
                      f"{cpu_time:.2f} s of oracle time, OpenMP over rows",
            "gpu_vs_oracle_max_abs_err_O": max_err, "gpu_vs_oracle_max_abs_err_L": max_err_l}


if __name__ == "__main__":
    main()

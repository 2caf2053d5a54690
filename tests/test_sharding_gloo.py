"""N > 1 path on CPU: world_size-2 gloo processes shard batch x head exactly like bench.py does on
GPUs (disjoint ranges, no data-path collective) and the per-rank oracle results, put side by side,
equal the single-process result."""
import os
import socket

import numpy as np
import pytest

from metal_flash_attention_amd.sharding import all_ranges, shard_range


def test_shard_ranges_partition_the_heads():
    for total in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 4, 8):
            ranges = all_ranges(total, world)
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (b0, e0), (b1, e1) in zip(ranges, ranges[1:]):
                assert e0 == b1
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, total_heads, out_queue):
    import torch.distributed as dist
    from metal_flash_attention_amd.sharding import max_over_ranks, shard_range
    from oracle import Network, NetworkDescriptor

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    begin, end = shard_range(total_heads, world, rank)
    outs = {}
    for unit in range(begin, end):     # each unit = one head: seed == global head index
        net = Network(NetworkDescriptor(24, 40, 16), seed=unit, threads=1)
        outs[unit] = net.run(backward=False)["O"]
    dist.barrier()
    slowest = max_over_ranks(0.001 * (rank + 1), dist)   # rank 1 is "slower"
    out_queue.put((rank, begin, end, {k: v.tobytes() for k, v in outs.items()}, slowest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import Network, NetworkDescriptor

    total_heads, world = 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_heads, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = {}
    for rank, begin, end, outs, slowest in results:
        assert (begin, end) == shard_range(total_heads, world, rank)
        assert abs(slowest - 0.002) < 1e-12          # max over ranks, identical on every rank
        covered.update(outs)
    assert sorted(covered) == list(range(total_heads))   # every head exactly once, none shared
    for unit in range(total_heads):
        ref = Network(NetworkDescriptor(24, 40, 16), seed=unit, threads=1).run(backward=False)["O"]
        assert covered[unit] == ref.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_rank_shards_of_real_device_buffers_equal_the_unsharded_launch(world):
    """The multi-GPU structure on one device: `world` ranks each launch the HIP kernels on THEIR contiguous range of the
    flattened batch x head axis of the same device buffers (exactly what bench.py does with one rank per GPU); together
    they must reproduce the single launch over all heads bit for bit -- no head skipped, none computed twice."""
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    B, H, N, D = 2, 5, 384, 128
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    q, k, v = (torch.randn((B * H, N, D), generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    stream = torch.cuda.current_stream().cuda_stream

    def run(lo, hi, o, l):
        kernel.dispatch({Op.Q: q[lo:hi], Op.K: k[lo:hi], Op.V: v[lo:hi], Op.O: o[lo:hi], Op.L: l[lo:hi]}, row=N, column=N,
                        heads=hi - lo, headStrides=hs, stream=stream)

    o_all = torch.full((B * H, N, D), float("nan"), device="cuda")
    l_all = torch.full((B * H, N), float("nan"), device="cuda")
    run(0, B * H, o_all, l_all)
    o_sh, l_sh = torch.full_like(o_all, float("nan")), torch.full_like(l_all, float("nan"))
    covered = []
    for rank in range(world):
        lo, hi = shard_range(B * H, world, rank)
        covered += list(range(lo, hi))
        if hi > lo:
            run(lo, hi, o_sh, l_sh)
    torch.cuda.synchronize()
    assert covered == list(range(B * H))
    assert not torch.isnan(o_sh).any() and torch.equal(o_sh, o_all) and torch.equal(l_sh, l_all)

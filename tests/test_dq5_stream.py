"""CPU checks of the hand-placed ROLE-SPLIT instruction stream of attn_dq16_p5 (tools/dq5gen.py: backwardQuery at the
head-dimension buckets 160 / 192 / 256) on the lane-exact model in tools/p4sim.py: the stream that is compiled into libmfa_hip.so
is executed instruction by instruction for one 128-row workgroup (two S-role waves, two P-role waves, P and dS' exchanged through
LDS) over all its key blocks and compared with a float64 backward pass (the formulas of the reference's Network.swift:202-402 in
matrix form).  No GPU, no oracle library needed."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import dq5gen  # noqa: E402
import dq5sim  # noqa: E402

V = dq5gen.VARIANTS


def _check(R, C, rblk=0, causal=False, cfg=None, seed=0, **kw):
    cfg = cfg or V["D256_BF16_FOLD"]
    e, m, wg = dq5sim.check(R=R, C=C, rblk=rblk, causal=causal, cfg=cfg, seed=seed, **kw)
    # dS' enters the last product in the 16-bit type and is formed from the ROUNDED P (the exchange carries packed fragments)
    rel = 3e-3 if cfg.dtype == "f16" else 1.5e-2
    assert e < rel * max(1.0, m), (e, m)
    return wg


@pytest.mark.parametrize("D", [160, 192, 256])
@pytest.mark.parametrize("C", [32, 64, 96, 128, 160, 192, 352])   # 1 .. 6 key blocks (every exit of the loop), 11: the five- and three-stage rings wrap
def test_key_blocks(C, D):
    _check(128, C, cfg=V["D%d_BF16_FOLD" % D])


@pytest.mark.parametrize("R,C,rblk", [(100, 77, 0), (200, 50, 1), (1, 128, 0), (129, 100, 1), (64, 33, 0)])
def test_ragged(R, C, rblk):
    _check(R, C, rblk=rblk, seed=1)
    _check(R, C, rblk=rblk, seed=1, cfg=V["D192_BF16_EXACT"])


@pytest.mark.parametrize("D", [160, 256])
@pytest.mark.parametrize("R,C,rblk", [(128, 128, 0), (300, 392, 1), (256, 256, 1), (200, 456, 0), (384, 384, 2)])
def test_causal(R, C, rblk, D):
    _check(R, C, rblk=rblk, causal=True, seed=2, cfg=V["D%d_BF16_FOLD" % D])


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0), (2, 0, 3, 1)])
def test_ring_and_exchange_discipline(dma_mode, order):
    # DMA data and the two exchanges' writes landing as early / as late as the waits allow, waves of either role running ahead of /
    # behind each other between barriers
    _check(128, 224, dma_mode=dma_mode, order=order, seed=3)
    _check(256, 288, rblk=1, dma_mode=dma_mode, order=order, seed=3, cfg=V["D160_F16_FOLD"], causal=True)


@pytest.mark.parametrize("name", [n for n, c in V.items() if not c.prof and not c.abl])   # (ablations: timing only)
def test_every_compiled_variant(name):
    _check(128, 96, cfg=V[name], seed=4)
    _check(200, 264, cfg=V[name], causal=True, seed=5, rblk=1)


@pytest.mark.parametrize("D,Dr", [(160, 136), (160, 152), (192, 176), (256, 200), (256, 232), (256, 248)])
def test_head_dimensions_inside_a_bucket(D, Dr):
    """chunks beyond the head dimension are zero-filled (LDS-DMA offsets out of range, cached fragments zeroed by the kernel)"""
    _check(100, 128, cfg=V["D%d_BF16_FOLD" % D], Dr=Dr, seed=6)
    _check(64, 100, cfg=V["D%d_F16_EXACT" % D], Dr=Dr, seed=7)


def test_instruction_budget():
    """n key blocks cost n x (2 nks + 4 x share) matrix instructions per wave; every LDS fragment feeds two of them"""
    for name in ("D160_BF16_FOLD", "D192_BF16_FOLD", "D256_BF16_FOLD"):
        cfg = V[name]
        n = 9
        wg = _check(128, 32 * n, cfg=cfg, seed=8)
        for w in wg.waves:
            role = w.id >> 1
            mfma = sum(c for op, c in w.count.items() if op.startswith("v_mfma"))
            assert mfma == n * (2 * cfg.nks + 4 * cfg.share(role)), (name, w.id, mfma)
            assert w.count["ds_read_b64_tr_b16"] == n * 2 * 2 * cfg.share(role)
            assert w.count["ds_read_b128"] == n * cfg.nks + n * 4          # row fragments + the partner's four exchange fragments
            assert w.count["ds_write_b128"] == n * 4


def test_stream_file_is_current(built_library):
    """csrc/attn_dq16_p5_stream.inc is what tools/dq5gen.py generates"""
    import tempfile
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_dq16_p5_stream.inc")
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        dq5gen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run make (tools/gen_streams.py)"

"""CPU checks of the hand-placed instruction stream of attn_dq16_p4 (tools/dq4gen.py) on the lane-exact model in
tools/p4sim.py: the stream that is compiled into libmfa_hip.so is executed instruction by instruction for one 256-row
workgroup over all its key tiles and compared with a float64 backward pass (the formulas of the reference's
Network.swift:202-330 in matrix form).  No GPU, no oracle library needed."""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import dq4gen  # noqa: E402
import dq4sim  # noqa: E402

V = dq4gen.VARIANTS


def _check(R, C, rblk=0, causal=False, cfg=None, seed=0, **kw):
    cfg = cfg or V["BF16_FOLD"]
    e, m, wg = dq4sim.check(R=R, C=C, rblk=rblk, causal=causal, cfg=cfg, seed=seed, **kw)
    rel = 2.5e-3 if cfg.dtype == "f16" else 1.2e-2     # dS enters the last product in the 16-bit type
    assert e < rel * max(1.0, m), (e, m)
    return wg


@pytest.mark.parametrize("C", [64, 128, 192, 256, 320, 576])   # 576 keys = 9 tiles: the four-stage ring wraps twice
def test_tile_counts(C):
    _check(256, C)


@pytest.mark.parametrize("R,C,rblk", [(256, 100, 0), (200, 130, 0), (300, 200, 1), (70, 1, 0)])
def test_ragged(R, C, rblk):
    _check(R, C, rblk=rblk, seed=1)


@pytest.mark.parametrize("R,C,rblk", [(256, 256, 0), (512, 512, 1), (300, 400, 1), (256, 320, 0)])
def test_causal_per_wave_bounds_and_skip_loop(R, C, rblk):
    _check(R, C, rblk=rblk, causal=True, seed=2)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_discipline(dma_mode, order):
    _check(256, 448, dma_mode=dma_mode, order=order, seed=3)
    _check(256, 448, causal=True, dma_mode=dma_mode, order=order, seed=3)


@pytest.mark.parametrize("name", [n for n, c in V.items() if not c.prof])
def test_every_compiled_variant(name):
    _check(256, 192, cfg=V[name], seed=4)
    _check(200, 260, cfg=V[name], causal=True, seed=5)


@pytest.mark.parametrize("name", ["D64_BF16_FOLD", "D64_F16_EXACT"])
@pytest.mark.parametrize("R,C,rblk,causal", [(256, 64, 0, False), (256, 128, 0, False), (256, 576, 0, False), (200, 130, 0, False),
                                             (70, 1, 0, False), (512, 512, 1, True), (300, 400, 1, True), (128, 128, 0, True)])
def test_d64_deferred_second_key_block(name, R, C, rblk, causal):
    """the 64 bucket's stream updates dQ with a tile's second key block beside the NEXT tile (and after the loop for the
    last one): one tile, many tiles (the ring wraps), ragged edges, waves that leave the loop early (causal)"""
    _check(R, C, rblk=rblk, causal=causal, cfg=V[name], seed=6)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_d64_ring_discipline(dma_mode, order):
    _check(256, 448, cfg=V["D64_BF16_FOLD"], dma_mode=dma_mode, order=order, seed=3)
    _check(256, 448, cfg=V["D64_BF16_FOLD"], causal=True, dma_mode=dma_mode, order=order, seed=3)


@pytest.mark.parametrize("name,dma_mode,order", [("BF16_FOLD_TR", "early", (3, 2, 1, 0)), ("BF16_EXACT_TR", "late", (0, 1, 2, 3)),
                                                 ("F16_FOLD_TR", "late", (0, 1, 2, 3)), ("F16_EXACT_TR", "early", (3, 2, 1, 0))])
def test_transposed_key_value_streams(name, dma_mode, order):
    """K and V handed over TRANSPOSED ([128][C], whole tiles): the images keep the source orientation and the two read recipes
    change roles -- K / V row fragments by transposing reads (Q' and dO in their element order), K^T fragments as two 8-byte
    reads in the order dS' holds its keys.  Model-verified streams (no kernel behind them yet): tile counts
    across two ring wraps, ragged row blocks, causal with per-wave bounds and the skip loop, DMA early / late, waves in either
    order."""
    cfg = dq4gen.TR_VARIANTS[name]
    for R, C, rblk, causal in ((256, 64, 0, False), (256, 576, 0, False), (200, 320, 0, False), (512, 512, 1, True), (300, 448, 1, True)):
        wg = _check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=11, dma_mode=dma_mode, order=order)
    assert wg.waves[0].count.get("ds_read_b128", 0) == 0 and wg.waves[0].count["ds_read_b64"] > 0


def test_stream_file_is_current(built_library):
    """csrc/attn_dq16_p4_stream.inc is what tools/dq4gen.py generates"""
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_dq16_p4_stream.inc")
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        dq4gen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run python tools/dq4gen.py"


def test_tile_shape_and_filler_budget():
    """96 matrix instructions per tile (exactly the 6 N^2 D flops of the algorithm: no extra k-steps); at most 7 other
    instructions in any gap except the seam (barrier, waits)"""
    for name in ("BF16_FOLD", "BF16_EXACT"):
        ins = dq4gen.Stream(V[name]).build()
        loop = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("LOOP"))
        end = next(i for i, x in enumerate(ins) if x.op == "s_cbranch_scc1" and x.mod.get("target", "").startswith("LOOP"))
        gaps, cur = [], None
        for x in ins[loop:end]:
            if x.op.startswith("v_mfma"):
                if cur is not None:
                    gaps.append(cur)
                cur = 0
            elif cur is not None and x.op != "label":
                cur += 1
        assert len(gaps) == dq4gen.N_MFMA - 1
        inner = gaps[:88] + gaps[89:]
        assert max(inner) <= 7, (name, max(inner), inner.index(max(inner)))
        assert sum(gaps) / len(gaps) < 3.8, sum(gaps) / len(gaps)


def test_d64_tile_shape():
    """the 64 bucket: 48 matrix instructions per tile in the loop plus 8 after it (the last tile's second key block), one
    barrier per tile, the exp2 / multiply / pack work spread so that no gap holds more than 40 VALU cycles of it"""
    ins = dq4gen.Stream(V["D64_BF16_FOLD"]).build()
    loop = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("LOOP"))
    end = next(i for i, x in enumerate(ins) if x.op == "s_cbranch_scc1" and x.mod.get("target", "").startswith("LOOP"))
    body = [x for x in ins[loop:end] if x.op != "label"]
    assert sum(1 for x in body if x.op.startswith("v_mfma")) == 48
    assert sum(1 for x in body if x.op == "s_barrier") == 1
    skip = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("SKIP"))
    assert sum(1 for x in ins[end:skip] if x.op.startswith("v_mfma")) == 8
    cost = {"v_exp_f32": 16, "v_mul_f32": 4, "v_cvt_pk_bf16_f32": 4, "v_pk_mul_f32": 8}
    gaps, cur = [], None
    for x in body:
        if x.op.startswith("v_mfma"):
            if cur is not None:
                gaps.append(cur)
            cur = 0
        elif cur is not None:
            cur += cost.get(x.op, 0)
    assert max(gaps) <= 40, max(gaps)
    assert sum(gaps) == 64 * 16 + 64 * 4 + 32 * 4 - (cur or 0) or sum(gaps) + (cur or 0) == 64 * 16 + 64 * 4 + 32 * 4

"""CPU checks of the hand-placed instruction stream of attn_dkv16_p4 (tools/dkv4gen.py) on the lane-exact model in
tools/p4sim.py: the stream that is compiled into libmfa_hip.so is executed instruction by instruction for one
256-key workgroup over all its row steps and compared with a float64 backward pass (the formulas of the reference's
Network.swift:202-330 in matrix form).  No GPU, no oracle library needed."""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import dkv4gen  # noqa: E402
import dkv4sim  # noqa: E402

V = dkv4gen.VARIANTS


def _check(R, C, cblk=0, causal=False, cfg=None, seed=0, **kw):
    cfg = cfg or V["BF16_MIXED"]
    ev, ek, mv, mk, wg = dkv4sim.check(R=R, C=C, cblk=cblk, causal=causal, cfg=cfg, seed=seed, **kw)
    # P and dS enter the second products in the 16-bit type (8 / 11 bits of mantissa), as in attn_dkv16_rs
    rel = 2.5e-3 if (cfg.dtype == "f16" and not cfg.mix) else 1.2e-2    # mix streams: P and V enter the dO products as BF16
    assert ev < rel * max(1.0, mv) and ek < rel * max(1.0, mk), (ev, mv, ek, mk)
    return wg


@pytest.mark.parametrize("R", [32, 96, 128, 160, 320])   # 320 rows: the four-stage ring wraps twice
def test_row_steps(R):
    _check(R, 256)


@pytest.mark.parametrize("R,C,cblk", [(77, 200, 0), (50, 300, 1), (1, 256, 0), (100, 257, 1)])
def test_ragged(R, C, cblk):
    _check(R, C, cblk=cblk, seed=1)


@pytest.mark.parametrize("R,C,cblk", [(256, 256, 0), (300, 520, 1), (200, 256, 0), (512, 512, 1)])
def test_causal(R, C, cblk):
    _check(R, C, cblk=cblk, causal=True, seed=2)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_discipline(dma_mode, order):
    # DMA data landing as early / as late as the waits allow, waves running ahead of / behind each other
    _check(224, 256, dma_mode=dma_mode, order=order, seed=3)


@pytest.mark.parametrize("name", [n for n, c in V.items() if not c.prof])
def test_every_compiled_variant(name):
    _check(96, 256, cfg=V[name], seed=4)
    _check(160, 200, cfg=V[name], causal=True, seed=5)


def test_exact_stream_is_closer():
    """the exact streams keep K as stored and scale in fp32: with FP16 inputs and FP32 L, D the result is an order of
    magnitude closer to float64 than with K pre-multiplied in the 16-bit type"""
    e_fold = dkv4sim.check(R=96, C=256, cfg=V["F16_MIXED"], seed=6)
    e_exact = dkv4sim.check(R=96, C=256, cfg=V["F16_F32"], seed=6)
    assert e_exact[1] < 0.5 * e_fold[1]


@pytest.mark.parametrize("name,dma_mode,order", [("BF16_MIXED_TR", "early", (3, 2, 1, 0)), ("BF16_F32_TR", "late", (0, 1, 2, 3)),
                                                 ("F16_MIXED_TR", "late", (0, 1, 2, 3)), ("F16_F32_TR", "early", (3, 2, 1, 0)),
                                                 ("F16_DOBF16_MIXED_TR", "early", (0, 1, 2, 3)), ("F16_DOBF16_F32_TR", "late", (3, 2, 1, 0))])
def test_transposed_query_gradient_streams(name, dma_mode, order):
    """Q and dO handed over TRANSPOSED ([128][R], whole 32-row steps): a step's tile in the source orientation is the same
    [4][32][64 bytes] image with the two read recipes' roles exchanged -- Q / dO row fragments by transposing reads (K' and V in
    their element order), dO^T / Q^T fragments as two 8-byte reads in the order P and dS' hold their rows; with BF16 dO^T next to FP16
    operands the two products that read it run in BF16.  Model-verified streams (developer kernels only): step counts across two ring wraps, ragged key blocks, causal, DMA early / late,
    waves in either order."""
    cfg = dkv4gen.TR_VARIANTS[name]
    for R, C, cblk, causal in ((32, 256, 0, False), (320, 256, 0, False), (96, 200, 0, False), (512, 512, 1, True), (288, 448, 1, True)):
        wg = _check(R, C, cblk=cblk, causal=causal, cfg=cfg, seed=12, dma_mode=dma_mode, order=order)
    assert wg.waves[0].count.get("ds_read_b128", 0) == 32 and wg.waves[0].count["ds_read_b64"] > 0   # (b128: the K' / V hand-over only)


def test_stream_file_is_current(built_library):
    """csrc/attn_dkv16_p4_stream.inc is what tools/dkv4gen.py generates"""
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_dkv16_p4_stream.inc")
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        dkv4gen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run python tools/dkv4gen.py"


def test_step_shape_and_filler_budget():
    """68 matrix instructions per step; at most 7 other instructions in any gap except the seam (barrier, waits)"""
    for name in ("BF16_MIXED", "BF16_F32"):
        ins = dkv4gen.Stream(V[name]).build()
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
        assert len(gaps) == dkv4gen.N_MFMA - 1
        inner = gaps[:60] + gaps[61:]
        assert max(inner) <= 7, (name, max(inner), inner.index(max(inner)))
        assert sum(gaps) / len(gaps) < 4.0, sum(gaps) / len(gaps)


def test_d64_step_shape():
    """the 64 bucket: 36 matrix instructions per step (S 10, dP 10 with the per-row k-step, dV^T 8, dK^T 8)"""
    for name in ("D64_BF16_MIXED", "D64_BF16_F32"):
        ins = dkv4gen.Stream(V[name]).build()
        loop = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("LOOP"))
        end = next(i for i, x in enumerate(ins) if x.op == "s_cbranch_scc1" and x.mod.get("target", "").startswith("LOOP"))
        body = [x for x in ins[loop:end] if x.op != "label"]
        n_mfma = sum(1 for x in body if x.op.startswith("v_mfma"))
        assert n_mfma == 36, (name, n_mfma)
        assert sum(1 for x in body if x.op == "s_barrier") == 1
        assert (len(body) - n_mfma) / n_mfma < 5.5, (name, len(body))   # the same exp2 / multiply / pack work as at D = 128 beside half the matrix instructions

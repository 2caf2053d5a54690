"""CPU checks of the hand-placed instruction stream of attn_fwd16_p5 (tools/f256gen.py; forward, 128 < D <= 256) on the
lane-exact model in tools/p4sim.py: the stream that is compiled into libmfa_hip.so is executed instruction by instruction
for one 256-row block and compared with a float64 attention.  No GPU, no oracle library needed."""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import f256gen  # noqa: E402
import f256sim  # noqa: E402

V = f256gen.VARIANTS


def _check(R, C, rblk=0, causal=False, cfg=None, seed=0, tol_o=4e-3, tol_l=2e-5, **kw):
    cfg = cfg or V["BF16_THR8"]
    if cfg.fold:
        tol_l = max(tol_l, 6e-4 if cfg.dtype == "f16" else 5e-3)   # Q * scale2 rounded to the 16-bit type
    dO, dL, wg = f256sim.check(R=R, C=C, rblk=rblk, causal=causal, cfg=cfg, seed=seed, **kw)
    assert dO < tol_o and dL < tol_l * 12, (dO, dL)
    return wg


@pytest.mark.parametrize("C", [32, 64, 96, 160, 288])      # 288 keys = 9 steps: the four-stage ring wraps twice
def test_step_counts(C):
    _check(256, C)


@pytest.mark.parametrize("R,C,rblk", [(256, 100, 0), (200, 130, 0), (300, 70, 1), (70, 1, 0)])
def test_ragged(R, C, rblk):
    _check(R, C, rblk=rblk, seed=1)


@pytest.mark.parametrize("R,C,rblk", [(256, 256, 0), (512, 512, 1), (300, 400, 1)])
def test_causal_per_wave_bounds_and_skip_loop(R, C, rblk):
    _check(R, C, rblk=rblk, causal=True, seed=2)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_discipline(dma_mode, order):
    _check(256, 224, dma_mode=dma_mode, order=order, seed=3)


@pytest.mark.parametrize("name", ["BF16_THR8", "BF16_FOLD"])
def test_deferred_rescale_spike(name):
    # a score far above the row's others at a late step must rescale O, l after every P^T V^T product of the step before
    wg = _check(256, 192, cfg=V[name], seed=4, spike=(5, 150, 3.0), tol_o=8e-3, tol_l=2e-3)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 256


@pytest.mark.parametrize("name", [n for n, c in V.items() if not c.prof])
def test_every_compiled_variant(name):
    _check(256, 160, cfg=V[name], seed=5)
    _check(256, 320, cfg=V[name], causal=True, seed=6)


@pytest.mark.parametrize("name", ["D256_BF16_THR8_TR", "D256_F16_FOLD_TR", "D192_BF16_FOLD_TR", "D160_F16_THR8_TR"])
def test_transposed_key_value_streams(name):
    """K and V handed over TRANSPOSED ([D][C], whole 32-key steps): a step's tile in the source orientation is the same
    [D/32][32][64 bytes] image with the two read recipes' roles exchanged -- K row fragments by transposing reads (Q' parked in their
    element order), V^T fragments as two 8-byte reads in the order P^T holds its keys.  Model-verified streams (no kernel behind them
    yet): step counts across two ring wraps, ragged row blocks, causal with per-wave bounds, a forced rescale, DMA
    early / late with the waves in either order."""
    cfg = f256gen.TR_VARIANTS[name]
    for R, C, rblk, causal, mode in ((256, 32, 0, False, "late"), (256, 288, 0, False, "early"), (200, 160, 0, False, "late"),
                                     (512, 512, 1, True, "early")):
        wg = _check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=14, dma_mode=mode, order=(3, 2, 1, 0) if mode == "early" else (0, 1, 2, 3))
    _check(256, 160, cfg=cfg, spike=(5, 100, 3.0), seed=15, tol_o=1.2e-2)
    # (b128 reads: only the Q' fragments taken over from LDS at the start)
    assert wg.waves[0].count.get("ds_read_b128", 0) == 2 * cfg.nks and wg.waves[0].count["ds_read_b64"] > 0


@pytest.mark.parametrize("name", ["D256_BF16_THR8_TRK", "D256_F16_FOLD_TRV", "D192_BF16_FOLD_TRK", "D160_F16_THR8_TRV",
                                  "D160_BF16_FOLD_TRK", "D192_F16_THR8_TRV"])
def test_streams_with_one_transposed_operand(name):
    """K^T alone or V^T alone (Cfg.tr is a bit mask like tools/p4gen.py's): the halves of a step's fragment list are independent --
    the transposed operand takes the exchanged read recipe (and, for V^T, four addresses), the other keeps the row-major one.
    Streams of f256gen.MODEL_ONLY_VARIANTS: not in the tracked generated file, `make DEV=1` generates them for the developer kernel."""
    cfg = f256gen.MODEL_ONLY_VARIANTS[name]
    for R, C, rblk, causal, mode in ((256, 32, 0, False, "late"), (256, 288, 0, False, "early"), (200, 160, 0, False, "late"),
                                     (512, 512, 1, True, "early")):
        wg = _check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=16, dma_mode=mode, order=(3, 2, 1, 0) if mode == "early" else (0, 1, 2, 3))
    _check(256, 160, cfg=cfg, spike=(5, 100, 3.0), seed=17, tol_o=1.2e-2)
    count = wg.waves[0].count
    if cfg.kt:     # K^T by transposing reads, V^T fragments of the row-major V too: no 16-byte read beyond the Q' hand-over
        assert count.get("ds_read_b128", 0) == 2 * cfg.nks and "ds_read_b64" not in count
    else:          # K rows 16 bytes at a time, V^T 8 bytes at a time: no transposing read at all
        assert count["ds_read_b128"] > 2 * cfg.nks and count["ds_read_b64"] > 0 and "ds_read_b64_tr_b16" not in count


@pytest.mark.parametrize("name", ["D256_BF16_THR8_TR", "D160_F16_FOLD_TR", "D192_BF16_FOLD_TR"])
def test_transposed_streams_survive_a_ragged_last_step_on_dense_rows(name):
    """Not something the kernel launches today (attn_fwd16_p5_tr takes whole 32-key steps), but a property of the streams worth
    knowing before widening it: with DENSE rows of K^T / V^T (leading dimension = C, C % 8 == 0) a ragged last step needs no extra
    offsets -- the keys beyond C in a row are the next row's finite values (zeros behind the last row): their scores fall to the
    edge mask, their P = 0 multiplies finite V^T values."""
    cfg = f256gen.TR_VARIANTS[name]
    for R, C, causal in ((256, 40, False), (200, 104, False), (300, 328, True)):
        _check(R, C, causal=causal, cfg=cfg, seed=18, tol_o=8e-3)   # (40 keys do not average BF16's roundings out: 4.7e-3 in the folded stream)


def test_stream_file_is_current(built_library):
    """csrc/attn_fwd16_p5_stream.inc is what tools/f256gen.py generates"""
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p5_stream.inc")
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        f256gen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run python tools/f256gen.py"

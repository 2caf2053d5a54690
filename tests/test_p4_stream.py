"""CPU checks of the hand-placed instruction stream of attn_fwd16_p4 (tools/p4gen.py) on the lane-exact model in
tools/p4sim.py: the stream that is compiled into libmfa_hip.so is executed instruction by instruction for one
256-row block and compared with a float64 attention (the formulas of the reference's Network.swift:134-200 in
matrix form).  No GPU, no oracle library needed."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import p4gen  # noqa: E402
import p4sim  # noqa: E402


def _check(R, C, rblk=0, causal=False, cfg=None, dma_mode="late", order=(0, 1, 2, 3), seed=0, spike=None, tol_o=4e-3, tol_l=2e-5, tr_pad=0):
    rng = np.random.default_rng(seed)
    f16 = cfg is not None and cfg.dtype == "f16"
    if cfg is not None and cfg.fold:
        tol_l = max(tol_l, 6e-4 if f16 else 5e-3)   # Q * scale2 rounded to the 16-bit type (11 / 8 bit mantissa)
    q, k, v = (p4sim.rand_bf16(s, rng, f16=f16) for s in ((R, 128), (C, 128), (C, 128)))
    if spike is not None:   # one key row aligned with one query row: forces the deferred rescale at a chosen tile
        qrow, krow, gain = spike
        qf = p4sim.bf16_to_f32(q[qrow].astype(np.uint32))
        k[krow] = p4sim.f32_to_bf16_rne((qf * gain).astype(np.float32)).astype(np.uint16)
    O, L, wg = p4sim.run_block(q, k, v, rblk, cfg=cfg, causal=causal, dma_mode=dma_mode, order=order, tr_pad=tr_pad)
    Oref, Lref = p4sim.reference(q, k, v, causal=causal, f16=f16)
    rows = np.arange(rblk * 256, min(R, rblk * 256 + 256))
    dO = np.abs(O[: len(rows)] - Oref[rows]).max()
    dL = np.abs(L[: len(rows)] - Lref[rows]).max()
    assert dO < tol_o, (dO, dL)
    assert dL < tol_l * max(1.0, np.abs(Lref[rows]).max()), (dO, dL)
    return wg


@pytest.mark.parametrize("C", [64, 128, 192, 256, 320])
def test_tile_counts(C):
    _check(256, C)


@pytest.mark.parametrize("R,C", [(256, 100), (200, 130), (256, 200), (70, 1)])
def test_ragged(R, C):
    _check(R, C, seed=1)


@pytest.mark.parametrize("R,C,rblk", [(256, 256, 0), (512, 512, 1), (300, 400, 1), (256, 320, 0)])
def test_causal(R, C, rblk):
    _check(R, C, rblk=rblk, causal=True, seed=2)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_discipline(dma_mode, order):
    # DMA data landing as early / as late as the waits allow, waves running ahead of / behind each other
    _check(256, 448, dma_mode=dma_mode, order=order, seed=3)


@pytest.mark.parametrize("thr", [0.0, 8.0])
def test_deferred_rescale_spike(thr):
    # cdna_hip_programming.md T13: a score far above the row's others at a late tile must rescale O, l and the
    # pending P exactly once
    cfg = p4gen.Cfg("bf16", thr, 0)
    wg = _check(256, 320, cfg=cfg, spike=(5, 200, 3.0), seed=4, tol_o=1.2e-2)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 128   # the rescale section ran (the first tile needs none: O = 0)
    _check(256, 320, cfg=cfg, spike=(40, 300, 4.0), seed=5, tol_o=1.2e-2)


@pytest.mark.parametrize("name", sorted(n for n, c in p4gen.VARIANTS.items() if not c.abl))   # (ablations: timing only)
def test_every_compiled_variant(name):
    cfg = p4gen.VARIANTS[name]
    if cfg.prof:   # the model has no shader clock (PROF adds stamps only)
        cfg = p4gen.Cfg(cfg.dtype, cfg.thr, cfg.xe, cfg.order_a, cfg.pad, 0, cfg.fold, cfg.xb, cfg.dma, cfg.abl, cfg.xf)
    _check(256, 256, cfg=cfg, seed=6)


FOLD = p4gen.VARIANTS["BF16_FOLD"]


@pytest.mark.parametrize("R,C,rblk,causal", [(256, 64, 0, False), (256, 130, 0, False), (200, 449, 0, False), (512, 512, 1, True),
                                             (300, 400, 1, True), (70, 1, 0, False)])
def test_fold_stream_shapes(R, C, rblk, causal):
    _check(R, C, rblk=rblk, causal=causal, cfg=FOLD, seed=8)


@pytest.mark.parametrize("dma_mode,order", [("early", (3, 2, 1, 0)), ("late", (0, 1, 2, 3))])
def test_fold_stream_ring_and_spike(dma_mode, order):
    wg = _check(256, 448, cfg=FOLD, dma_mode=dma_mode, order=order, spike=(5, 300, 3.0), seed=9, tol_o=1.2e-2)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 128   # the spike (the first tile re-bases nothing: O = 0)


@pytest.mark.parametrize("name", ["BF16_THR8", "BF16_FOLD"])
@pytest.mark.parametrize("dma_mode,order", [("early", (3, 2, 1, 0)), ("late", (0, 1, 2, 3)), ("early", (0, 1, 2, 3)), ("late", (3, 2, 1, 0))])
def test_causal_per_wave_bounds_and_skip_loop(name, dma_mode, order):
    """causal: a wave stops multiplying at the diagonal of ITS rows and then only keeps the barriers and its LDS-DMA share
    going for the others; waves running ahead / behind, DMA landing early / late"""
    cfg = p4gen.VARIANTS[name]
    for R, C, rblk in ((512, 512, 1), (256, 256, 0), (700, 1000, 2), (300, 300, 1)):
        wg = _check(R, C, rblk=rblk, causal=True, cfg=cfg, dma_mode=dma_mode, order=order, seed=13)
    counts = [w.count.get("v_mfma_f32_32x32x16_bf16", 0) for w in wg.waves]
    assert len(set(w.count["s_barrier"] for w in wg.waves)) == 1


@pytest.mark.parametrize("name", ["BF16_THR8_TRK", "F16_FOLD_TRK", "BF16_FOLD_TRV", "F16_THR8_TRV"])   # (the other four: test_every_compiled_variant)
def test_streams_with_one_transposed_operand(name):
    """K^T alone / V^T alone: the same exchange for that operand only (V^T alone: the step's two chunk addresses in scratch
    registers, K keeps its eight); ragged last tiles with NaN behind the sequence, causal, a forced rescale."""
    cfg = p4gen.VARIANTS[name]
    for R, C, rblk, causal, mode in ((256, 456, 0, False, "late"), (512, 520, 1, True, "early"), (200, 64, 0, False, "late")):
        _check(R, C, rblk=rblk, causal=causal, cfg=cfg, dma_mode=mode, order=(3, 2, 1, 0) if mode == "early" else (0, 1, 2, 3), seed=31, tr_pad=8)
    wg = _check(256, 448, cfg=cfg, spike=(5, 300, 3.0), seed=32, tol_o=1.2e-2, tr_pad=16)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 128   # the rescale section ran
    assert (wg.waves[0].count.get("ds_read_b128", 0) == 0) == bool(cfg.kt) and (wg.waves[0].count.get("ds_read_b64", 0) > 0) == bool(cfg.vt)


@pytest.mark.parametrize("name,dma_mode,order", [("BF16_THR8_TR", "early", (3, 2, 1, 0)), ("BF16_THR8_TR", "late", (0, 1, 2, 3)),
                                                 ("BF16_FOLD_TR", "early", (3, 2, 1, 0)), ("F16_THR8_TR", "late", (0, 1, 2, 3)),
                                                 ("F16_FOLD_TR", "early", (3, 2, 1, 0))])
def test_transposed_streams(name, dma_mode, order):
    """K and V handed over TRANSPOSED ([128][C], whole tiles): the images keep the source orientation and the read recipes change
    places (K^T fragments by transposing reads, Q in their element order; V^T 8 bytes at a time from swizzled rows) -- tile counts
    across the ring's wrap, ragged row blocks, causal with per-wave bounds and the skip loop, a forced rescale, DMA landing
    early / late with the waves in either order."""
    cfg = p4gen.VARIANTS[name]
    for R, C, rblk, causal in ((256, 64, 0, False), (256, 448, 0, False), (200, 320, 0, False), (512, 512, 1, True), (300, 448, 1, True),
                               (700, 1024, 2, True)):
        _check(R, C, rblk=rblk, causal=causal, cfg=cfg, dma_mode=dma_mode, order=order, seed=21)
    # a partial last tile, rows padded with NaN: the last V^T tile comes through offsets of its own (zeros from key C on), the
    # scores of K^T's padding are replaced by the edge mask
    for R, C, rblk, causal in ((256, 40, 0, False), (256, 72, 0, False), (256, 456, 0, False), (512, 520, 1, True), (700, 1000, 2, True)):
        _check(R, C, rblk=rblk, causal=causal, cfg=cfg, dma_mode=dma_mode, order=order, seed=23, tr_pad=24)
    wg = _check(256, 448, cfg=cfg, dma_mode=dma_mode, order=order, spike=(5, 300, 3.0), seed=22, tol_o=1.2e-2)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 128   # the rescale section ran
    assert wg.waves[0].count.get("ds_read_b128", 0) == 0 and wg.waves[0].count["ds_read_b64"] > 0


def test_fold_stream_very_negative_scores():
    """every score far below zero: the first tile must still set m to the true maximum (m starts at 0 in FOLD streams)"""
    rng = np.random.default_rng(3)
    q = p4sim.rand_bf16((256, 128), rng)
    k = p4sim.f32_to_bf16_rne((-3.0 * p4sim.bf16_to_f32(q[:192].astype(np.uint32))).astype(np.float32).reshape(-1)).astype(np.uint16).reshape(192, 128)
    v = p4sim.rand_bf16((192, 128), rng)
    O, L, _ = p4sim.run_block(q, k, v, 0, cfg=FOLD)
    Oref, Lref = p4sim.reference(q, k, v)
    assert np.isfinite(O).all() and np.abs(O - Oref).max() < 2e-2 and np.abs(L - Lref).max() < 0.2


def test_rendered_memory_instructions_keep_their_offsets():
    """the model executes the instruction list, the GPU the rendered text: every LDS read's immediate offset must be in it
    (a plain ds_read_b64 once went out without -- every head-dimension block but the first read the wrong rows)"""
    for name in ("BF16_THR8", "BF16_FOLD_TR"):
        for ins in p4gen.Stream(p4gen.VARIANTS[name]).build():
            if ins.op.startswith("ds_read") and ins.mod.get("offset"):
                assert ("offset:%d" % ins.mod["offset"]) in p4gen.render_one(ins), p4gen.render_one(ins)


def test_stream_file_is_current(built_library):
    """csrc/attn_fwd16_p4_stream.inc is what tools/p4gen.py generates"""
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p4_stream.inc")
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        p4gen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run python tools/p4gen.py"


@pytest.mark.parametrize("name,most,mean", [("BF16_FOLD", 11, 6.8), ("BF16_THR8", 10, 7.8)])
def test_filler_budget(name, most, mean):
    """issue slots per gap of the steady-state phases (round-5 schedule, p4gen Cfg.bal): a wave alone on its SIMD issues one
    instruction per ~4 clocks and a matrix instruction holds the pipe for 32, so a gap has eight slots -- the matrix instruction
    takes one, a transcendental two, anything else (counted waits included) one.  The generator deals the fillers under a cap of
    7 (FOLD streams) or 8 (exact-scale streams: 64 more multiply-subtracts per tile); the counted waits and the decision (seven
    slots in one gap of the FOLD streams) come on top in a few gaps"""
    ins = p4gen.Stream(p4gen.VARIANTS[name]).build()
    loop = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("LOOP"))
    end = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("ENDEVEN"))
    gaps, cur = [], None
    for x in ins[loop:end]:
        if x.op.startswith("v_mfma"):
            if cur is not None:
                gaps.append(cur)
            cur = 1
        elif cur is not None and x.op not in ("label",):
            cur += 2 if x.op == "v_exp_f32" else 1
    assert len(gaps) >= 127
    first = gaps[:64]                      # one tile: phase A gaps 0..31 (31 = the seam with the barrier), phase B 32..63
    inner = first[:31] + first[32:63]      # the two phase seams carry waits, barrier, loop control (and the skipped mask code)
    assert max(inner) <= most, max(inner)
    assert sum(inner) / len(inner) < mean, sum(inner) / len(inner)


def test_lds_dma_pieces_lead_phase_b():
    """round 5: the LDS-DMA pieces of K(j+2) and V(j+1) are the FIRST fillers of phase B(j) (two phases of flight to the next
    barrier instead of one and a quarter; profiles/r05_p4p_bal2_early_dma.txt): in every steady-state phase B all eight pieces
    are issued within the first eight matrix-instruction gaps, in front of every K fragment read of that phase"""
    for name in ("BF16_FOLD", "BF16_THR8"):
        ins = p4gen.Stream(p4gen.VARIANTS[name]).build()
        loop = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("LOOP"))
        end = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("ENDEVEN"))
        g, dma_gaps, kread_gaps = -1, [], []
        for x in ins[loop:end]:
            if x.op.startswith("v_mfma"):
                g += 1
            elif x.op == "buffer_load_dwordx4_lds":
                dma_gaps.append(g % 64)
            elif x.op == "ds_read_b128":
                kread_gaps.append(g % 64)
        assert len(dma_gaps) == 16 and all(32 <= d < 40 for d in dma_gaps), dma_gaps
        assert min(kread_gaps) >= max(dma_gaps)


def test_model_executes_buffer_stores():
    """buffer_store_dwordx4 in the lane-exact model (for streams that write their results themselves):
    16 bytes per lane at its own offset, out-of-range lanes dropped by the resource bounds, counted in vmcnt"""
    from p4gen import A, I, SN, V, VN, Stream, render
    st = Stream(p4gen.VARIANTS["BF16_FOLD"])
    for r in range(4):
        st.emit("v_mov_b32", V(40 + r), [I(r + 1)])
    st.emit("v_accvgpr_write_b32", A(7), [I(9)])
    st.emit("buffer_store_dwordx4", None, [V(40, 4), VN("off"), SN("ores", 4)])
    st.emit("s_waitcnt", None, [], vmcnt=0)
    assert any(t.startswith("buffer_store_dwordx4 v[40:43], %[off], %[ores], 0 offen") for t in render(st.ins))
    wg = p4sim.Workgroup(st.ins)
    mem = np.zeros(64 * 16, np.uint8)          # offsets start at byte 8: the last lane of wave 3 falls outside
    for w in wg.waves:
        w.vn["off"] = ((np.arange(64) + 64 * (w.id & 0)) * 16 + (8 if w.id == 3 else 0)).astype(np.uint32)
        w.sn["ores"] = (mem, mem.size if w.id == 3 else 0)   # only wave 3 has a live resource
    wg.run((0, 1, 2, 3))
    words = mem[8:8 + 63 * 16].view(np.uint32).reshape(63, 4)
    assert (words == np.array([1, 2, 3, 4], np.uint32)).all()
    assert not mem[:8].any() and not mem[8 + 63 * 16:].any()
    assert all(not w.vm_q for w in wg.waves)

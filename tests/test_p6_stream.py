"""CPU checks of the D <= 64 PERSISTENT forward stream (tools/p6gen.py -> csrc/attn_fwd16_p6_stream.inc) on the lane-exact model
(tools/p6sim.py over tools/p4psim.py): one workgroup walks several 256-row blocks inside ONE instruction stream -- rings of four
K / V images filled two tiles ahead, row sums in the matrix pipe (L^T += ONES P^T), O^T through a staging area of its own -- and
every row of every block must equal a float64 attention (Network.swift:134-200 in matrix form).  No GPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import p6gen  # noqa: E402
import p6sim  # noqa: E402
import p4psim  # noqa: E402

FOLD = p6gen.VARIANTS["BF16_FOLD_L16"]


def _pairs(H, R):
    """the causal launch's table order: per head the pairs (last - i, i) of row blocks, the long one first"""
    nrb = (R + 255) // 256
    out = []
    for h in range(H):
        for p in range((nrb + 1) // 2):
            out.append((h, nrb - 1 - p))
            if p != nrb - 1 - p:
                out.append((h, p))
    return out


def _check(H, R, C, cfg=FOLD, blocks=None, seed=0, D=64, spike=None, tol_o=None, **kw):
    rng = np.random.default_rng(seed)
    f16 = cfg.dtype == "f16"
    q, k, v = (p4psim.rand_bf16(s, rng, f16=f16) for s in ((H, R, D), (H, C, D), (H, C, D)))
    if spike is not None:   # one key aligned with one query of head 0: forces the deferred rescale at a chosen tile
        qrow, krow, gain = spike
        qf = p4psim.h16_to_f32(q[0, qrow].astype(np.uint32), f16)
        k[0, krow] = p4psim.f32_to_h16((qf * gain).astype(np.float32), f16).astype(np.uint16)
    nrb = (R + 255) // 256
    if blocks is None:
        blocks = _pairs(H, R) if cfg.causal else [(h, rb) for h in range(H) for rb in range(nrb)]
    O, L, wg, raw = p6sim.run_workgroup(q, k, v, blocks, cfg, D=D, **kw)
    tol_o = tol_o or ((3e-2 if not f16 else 4e-3) if cfg.o16 else 6e-3)   # (bf16 P, few keys: the rounding of P does not average out)
    tol_l = (2e-2 if cfg.l16 else 2e-5) + (6e-4 if f16 else 5e-3)
    for h, rb in blocks:
        Oref, Lref = p4psim.reference(q[h], k[h], v[h], causal=bool(cfg.causal), f16=f16)
        rows = slice(rb * 256, min(R, rb * 256 + 256))
        dO, dL = np.abs(O[h, rows] - Oref[rows]).max(), np.abs(L[h, rows] - Lref[rows]).max()
        assert dO < tol_o and dL < tol_l * max(1.0, np.abs(Lref[rows]).max() / 8), (h, rb, dO, dL)
    return wg, raw, (q, k, v)


@pytest.mark.parametrize("C", [1, 64, 128, 192, 256, 449, 512])
def test_tile_counts(C):
    """a block walks a multiple of four tiles (the loop body is four tiles = the ring of four images, every ring position an
    immediate; the LDS-DMA runs two tiles ahead): the surplus tiles are fully masked"""
    _check(2, 256, C)


@pytest.mark.parametrize("H,R,C", [(1, 700, 130), (3, 200, 100), (2, 300, 320)])
def test_ragged_rows_and_keys(H, R, C):
    _check(H, R, C, seed=1)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("stores", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_and_store_discipline_exact(dma_mode, stores, order):
    _check(2, 256, 320, cfg=p6gen.VARIANTS["BF16_EXACT"], dma_mode=dma_mode, stores=stores, order=order, seed=9)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("stores", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_and_store_discipline(dma_mode, stores, order):
    """LDS-DMA data landing as early / as late as the waits allow (vmcnt(4): the pieces of the previous phase B may still fly),
    stores reaching memory at issue / only when a wait retires them, waves ahead of / behind each other -- three blocks"""
    _check(3, 256, 384, dma_mode=dma_mode, stores=stores, order=order, seed=2)


@pytest.mark.parametrize("C", [256, 320, 384, 448])
def test_ring_position_carries_over_blocks(C):
    """tile counts 4 and 8 (key counts that round up by one to three masked tiles among them): every block starts in ring
    image 0, where its predecessor's run-ahead pieces put its first tiles"""
    _check(4, 256, C, dma_mode="late", seed=8)


def test_blocks_out_of_order_and_single_block():
    _check(2, 512, 192, blocks=[(1, 1), (0, 0), (1, 0), (0, 1)], seed=3)
    _check(1, 256, 320, blocks=[(0, 0)], seed=3)


def test_untouched_blocks_keep_their_bytes():
    wg, (om, lm), _ = _check(2, 300, 128, blocks=[(1, 0)], seed=4)
    o = om.view(np.float32).reshape(2, 300, 64)
    assert (om.reshape(2, 300, 256)[0] == 0xCD).all() and (om.reshape(2, 300, 256)[1, 256:] == 0xCD).all()
    assert np.isfinite(o[1, :256]).all()
    l = lm.view(np.uint16).reshape(2, 300)
    assert (l[0] == 0xCDCD).all() and (l[1, 256:] == 0xCDCD).all() and (l[1, :256] != 0xCDCD).any()


@pytest.mark.parametrize("name", ["BF16_FOLD_L16", "BF16_FOLD_L16_VSUM", "BF16_EXACT"])
def test_deferred_rescale_in_a_later_block(name):
    """cdna_hip_programming.md T13 across a block switch: the spike sits in the SECOND block of head 0; with `lsum` the rescale
    multiplies the L^T accumulators together with O^T"""
    cfg = p6gen.VARIANTS[name]
    wg, _, _ = _check(1, 512, 320, cfg=cfg, spike=(300, 200, 3.0), seed=5, tol_o=1.2e-2)
    per_block = 64 + (2 if cfg.lsum else 0)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 2 * per_block + 64


@pytest.mark.parametrize("name", sorted(p6gen.VARIANTS))
def test_every_stream(name):
    cfg = p6gen.VARIANTS[name]
    if cfg.abl:
        pytest.skip("timing-only ablation")
    if cfg.split:
        pytest.skip("split streams: test_column_parallel_pieces")
    _check(2, 256, 320 if cfg.causal else 200, cfg=cfg, seed=6)     # (causal needs C >= R)


CAUSAL_FOLD = p6gen.VARIANTS["BF16_FOLD_L16_CAUSAL"]
CAUSAL_EXACT = p6gen.VARIANTS["BF16_EXACT_CAUSAL"]


@pytest.mark.parametrize("H,R,C", [(1, 512, 512), (2, 768, 768), (1, 300, 400), (1, 700, 1000), (1, 256, 256), (1, 1280, 1280)])
def test_causal_pairs(H, R, C):
    """row r sees key c iff c <= r + C - R: per block the stream computes its tile count (a multiple of four), the first masked
    tile of each wave and the lanes' limits; every wave walks all of the block's tiles (keys beyond its diagonal are masked)"""
    _check(H, R, C, cfg=CAUSAL_FOLD, seed=10, dma_mode="late")


@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_causal_exact_and_wave_orders(order):
    _check(1, 768, 832, cfg=CAUSAL_EXACT, seed=11, order=order)
    _check(2, 512, 512, cfg=CAUSAL_FOLD, seed=12, order=order, stores="early")


def test_causal_blocks_in_any_table_order():
    _check(1, 1024, 1024, cfg=CAUSAL_FOLD, blocks=[(0, 0), (0, 3), (0, 1), (0, 2)], seed=13)


@pytest.mark.parametrize("name", ["BF16_FOLD_SPLIT", "BF16_EXACT_SPLIT", "F16_FOLD_SPLIT"])
@pytest.mark.parametrize("R,C,splits", [(256, 1024, 4), (512, 512, 2), (300, 2048, 2)])
def test_column_parallel_pieces(name, R, C, splits):
    """split streams: a table entry is (row block, piece of the key range); the un-normalised O^T and (m, l) of every piece land in
    the workspace slabs and the merge of attn_fwd_combine (restated in tools/p6sim.py) gives the attention of the whole key range.
    One workgroup walks all pieces of all row blocks here, in an order that mixes them"""
    cfg = p6gen.VARIANTS[name]
    rng = np.random.default_rng(20)
    f16 = cfg.dtype == "f16"
    H, D = 2, 64
    q, k, v = (p4psim.rand_bf16(s, rng, f16=f16) for s in ((H, R, D), (H, C, D), (H, C, D)))
    nrb = (R + 255) // 256
    blocks = [(h, rb, sp) for sp in range(splits) for h in range(H) for rb in range(nrb)]
    O, L, wg, _ = p6sim.run_workgroup(q, k, v, blocks, cfg, D=D, splits=splits, dma_mode="late")
    for h in range(H):
        Oref, Lref = p4psim.reference(q[h], k[h], v[h], causal=False, f16=f16)
        dO, dL = np.abs(O[h] - Oref).max(), np.abs(L[h] - Lref).max()
        assert dO < 6e-3 and dL < 6e-3, (h, dO, dL)


@pytest.mark.parametrize("D", [8, 40, 56])
@pytest.mark.parametrize("name", ["BF16_FOLD_L16", "BF16_FOLD_O16_L16", "BF16_EXACT"])
def test_head_dimensions_below_the_bucket(D, name):
    """D < 64: chunks beyond D are fetched out of range (zeros) and the stores of columns >= D are issued out of range; bytes
    beyond column D stay untouched"""
    cfg = p6gen.VARIANTS[name]
    wg, (om, lm), _ = _check(2, 256, 320, cfg=cfg, D=D, seed=7, ld=64)
    osz = 2 if cfg.o16 else 4
    assert (om.reshape(2, 256, 64 * osz)[:, :, D * osz:] == 0xCD).all()
    assert wg.waves[0].count["buffer_store_dwordx2" if cfg.o16 else "buffer_store_dwordx4"] == 2 * 16


def test_row_sums_run_in_the_matrix_pipe():
    """`lsum`: 40 matrix instructions per steady tile (16 + 16 + 8) and no row-sum addition; the developer stream without it has
    32 and 64 additions"""
    for name, nm, nadd in (("BF16_FOLD_L16", 40, 0), ("BF16_FOLD_L16_VSUM", 32, 64), ("BF16_EXACT", 32, 64 + 2)):   # (+ 2: m + THR per row block)
        body = _loop_body(p6gen.Stream6(p6gen.VARIANTS[name]).build())
        assert sum(1 for x in body if x.op.startswith("v_mfma")) == 4 * nm
        assert sum(1 for x in body if x.op == "v_add_f32") == 4 * nadd


def _loop_body(ins):
    loop = next(i for i, x in enumerate(ins) if x.op == "label" and x.mod["name"].startswith("LOOP"))
    end = next(i for i, x in enumerate(ins) if x.op == "s_branch" and x.mod["target"].startswith("LOOP"))
    return ins[loop:end]


def test_lds_dma_runs_two_tiles_ahead():
    """the pieces lead phase B and the loop's wait is vmcnt(4): what phase B(j-1) requested may still be in flight at barrier j;
    the loop body is four tiles with ONE exit test, and no address arithmetic (every LDS read takes its ring image as an immediate)"""
    body = _loop_body(p6gen.Stream6(FOLD).build())
    waits = [x.mod["vmcnt"] for x in body if x.op == "s_waitcnt" and "vmcnt" in x.mod]
    assert waits == [4, 4, 4, 4]
    g, dma = -1, []
    for x in body:
        if x.op.startswith("v_mfma"):
            g += 1
        elif x.op == "buffer_load_dwordx4_lds":
            dma.append(g % 40)
    assert len(dma) == 16 and all(16 <= d < 20 for d in dma), dma
    assert sum(1 for x in body if x.op == "s_cmp_ge_i32") == 1
    assert not any(x.op in ("v_add_u32", "v_xor_b32") for x in body if x.op != "label")


def test_stream_file_is_current(built_library):
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p6_stream.inc")
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        p6gen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run python tools/p6gen.py"


# ---- round 6: per-batch lengths on the persistent kernel (the causal streams read rows / keys per block-table entry) -----------------
def _check_lengths(cfg, lens, cflag, R, C, seed=0, **kw):
    rng = np.random.default_rng(seed)
    H = len(lens)
    q, k, v = (p4psim.rand_bf16(s_, rng) for s_ in ((H, R, 64), (H, C, 64), (H, C, 64)))
    blocks = [(h, rb) for h in range(H) for rb in range((lens[h][0] + 255) // 256)]   # the compacted table: non-empty row blocks only
    O, L, wg, (om, lm) = p6sim.run_workgroup(q, k, v, blocks, cfg, lengths={h: lens[h] for h in range(H)}, cflag=cflag, **kw)
    for h, (Rb, Cb) in enumerate(lens):
        Oref, Lref = p4psim.reference(q[h, :Rb], k[h, :Cb], v[h, :Cb], causal=bool(cflag))
        dO, dL = np.abs(O[h, :Rb] - Oref).max(), np.abs(L[h, :Rb] - Lref).max()
        assert dO < 8e-3 and dL < (2e-2 + 5e-3) * max(1.0, np.abs(Lref).max() / 8), (h, Rb, Cb, dO, dL)
        assert (om.reshape(H, R, -1)[h, Rb:] == 0xCD).all() and (lm.reshape(H, R, -1)[h, Rb:] == 0xCD).all(), (h, Rb)
    return wg


@pytest.mark.parametrize("cflag", [0, 1])
def test_per_batch_lengths_on_the_persistent_kernel(cflag):
    """each table entry carries the rows and keys of its batch entry; with the causal mask every entry has its own diagonal offset
    (clamped at 0 when it has fewer keys than rows), without it the offset is out of reach and the geometry is the dense one"""
    lens = [(300, 520), (77, 100), (512, 64), (130, 130), (256, 700)]
    if cflag:
        lens.append((300, 200))
    _check_lengths(CAUSAL_FOLD, lens, cflag, 512, 704, seed=51 + cflag)
    _check_lengths(CAUSAL_EXACT, lens[:3], cflag, 512, 704, seed=53 + cflag, dma_mode="early", stores="late", order=(3, 2, 1, 0))

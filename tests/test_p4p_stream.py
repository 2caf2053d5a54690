"""CPU checks of the PERSISTENT forward stream (tools/p4pgen.py -> csrc/attn_fwd16_p4p_stream.inc) on the lane-exact model
(tools/p4psim.py): one workgroup walks several 256-row blocks inside ONE instruction stream -- block table in LDS, the next
block's Q / K / V requested by LDS-DMA under the last two tiles, O / l and L = m + log2 l stored from the registers behind a
`vmcnt(34)` -- and every row of every block must equal a float64 attention (Network.swift:134-200 in matrix form).  No GPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import p4pgen  # noqa: E402
import p4psim  # noqa: E402

FOLD = p4pgen.VARIANTS["BF16_FOLD_L16"]
EXACT = p4pgen.VARIANTS["BF16_EXACT"]


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


def _check(H, R, C, cfg=FOLD, blocks=None, seed=0, D=128, spike=None, tol_o=None, **kw):
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
    O, L, wg, raw = p4psim.run_workgroup(q, k, v, blocks, cfg, D=D, **kw)
    tol_o = tol_o or ((3e-2 if not f16 else 4e-3) if cfg.o16 else 4e-3)
    tol_l = (2e-2 if cfg.l16 else 2e-5) + (6e-4 if f16 else 5e-3) * bool(cfg.fold)
    pad = lambda x: np.concatenate([x, np.zeros(x.shape[:-1] + (128 - D,), x.dtype)], axis=-1) if D < 128 else x
    for h, rb in blocks:
        Oref, Lref = p4psim.reference(q[h], k[h], v[h], causal=bool(cfg.causal), f16=f16)
        rows = slice(rb * 256, min(R, rb * 256 + 256))
        dO, dL = np.abs(O[h, rows] - Oref[rows]).max(), np.abs(L[h, rows] - Lref[rows]).max()
        assert dO < tol_o and dL < tol_l * max(1.0, np.abs(Lref[rows]).max() / 8), (h, rb, dO, dL)
    return wg, raw, (q, k, v)


@pytest.mark.parametrize("C", [1, 64, 128, 192, 449])
def test_tile_counts_even_and_odd(C):
    """an odd tile count walks one extra, fully masked tile; C <= 64 walks two"""
    _check(2, 256, C)


@pytest.mark.parametrize("H,R,C", [(1, 700, 130), (3, 200, 100), (2, 300, 320)])
def test_ragged_rows_and_keys(H, R, C):
    _check(H, R, C, seed=1)


@pytest.mark.parametrize("dma_mode", ["early", "late"])
@pytest.mark.parametrize("stores", ["early", "late"])
@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 2, 1, 0)])
def test_ring_and_store_discipline(dma_mode, stores, order):
    """LDS-DMA data landing as early / as late as the waits allow, stores reaching memory at issue / only when a wait retires
    them, waves running ahead of / behind each other -- three blocks, so two block switches"""
    _check(3, 256, 256, dma_mode=dma_mode, stores=stores, order=order, seed=2)


def test_blocks_out_of_order_and_single_block():
    """the table decides which block comes next (fwd16_decode_block's order is not ascending rows)"""
    _check(2, 512, 192, blocks=[(1, 1), (0, 0), (1, 0), (0, 1)], seed=3)
    _check(1, 256, 320, blocks=[(0, 0)], seed=3)


def test_untouched_blocks_keep_their_bytes():
    """a workgroup writes the rows of ITS blocks only; rows beyond R and L of other blocks stay as they were"""
    wg, (om, lm), _ = _check(2, 300, 128, blocks=[(1, 0)], seed=4)
    o = om.view(np.float32).reshape(2, 300, 128)
    assert (om.reshape(2, 300, 512)[0] == 0xCD).all() and (om.reshape(2, 300, 512)[1, 256:] == 0xCD).all()
    assert np.isfinite(o[1, :256]).all()
    l = lm.view(np.uint16).reshape(2, 300)
    assert (l[0] == 0xCDCD).all() and (l[1, 256:] == 0xCDCD).all() and (l[1, :256] != 0xCDCD).any()


@pytest.mark.parametrize("cfg", [FOLD, EXACT], ids=["fold", "exact"])
def test_deferred_rescale_in_a_later_block(cfg):
    """cdna_hip_programming.md T13 across a block switch: the spike sits in the SECOND block of head 0"""
    wg, _, _ = _check(1, 512, 320, cfg=cfg, spike=(300, 200, 3.0), seed=5, tol_o=1.2e-2)
    assert wg.waves[0].count.get("v_accvgpr_read_b32", 0) >= 2 * 128 + 128   # two epilogues + at least one rescale


@pytest.mark.parametrize("name", sorted(p4pgen.PRODUCT_STREAMS))   # (the PROF stream adds clock stamps only)
def test_every_compiled_stream(name):
    cfg = p4pgen.VARIANTS[name]
    if cfg.split:
        pytest.skip("split streams: test_column_parallel_pieces")
    _check(2, 256, 320 if cfg.causal else 200, cfg=cfg, seed=6)     # (causal needs C >= R)


@pytest.mark.parametrize("D", [72, 96, 120])
@pytest.mark.parametrize("name", ["BF16_FOLD_L16", "BF16_EXACT_O16"])
def test_head_dimensions_below_the_bucket(D, name):
    """D < 128: chunks beyond D are fetched out of range (zeros) and the stores of columns >= D are issued out of range --
    the count of stores per block (what vmcnt(34) relies on) does not change; bytes beyond column D stay untouched"""
    cfg = p4pgen.VARIANTS[name]
    wg, (om, lm), _ = _check(2, 256, 192, cfg=cfg, D=D, seed=7, ld=128)
    osz = 2 if cfg.o16 else 4
    assert (om.reshape(2, 256, 128 * osz)[:, :, D * osz:] == 0xCD).all()
    assert wg.waves[0].count["buffer_store_dwordx2" if cfg.o16 else "buffer_store_dwordx4"] == 2 * 32


# ---- round 6: the branch-free loop (PCfg.fastloop) -- tiles 1 .. fend - 1 of a block run in a copy of the loop without the mask,
# block-switch and pending-rescale tests; a rescale decided there continues in the ordinary copy of the same phase
@pytest.mark.parametrize("C", [256, 320, 384, 448, 512, 576, 640, 700, 1100])
@pytest.mark.parametrize("cfg", [FOLD, EXACT], ids=["fold", "exact"])
def test_branch_free_loop_boundaries(cfg, C):
    """every number of branch-free tile pairs from none (C <= 256) upwards, even and odd tile counts, a ragged last tile: the tiles
    behind the loop -- the last two of a block (block switch) and the masked ones -- run in the ordinary copy"""
    assert cfg.fastloop
    wg, _, _ = _check(2, 256, C, cfg=cfg, seed=C)
    nt = (C + 63) // 64
    nt += nt & 1
    fend = (min(C // 64, nt - 2) - 1) | 1
    pairs = max(0, (fend - 1) // 2)
    # per block: tile 0, the loop tiles and the tail multiply 64 times each
    assert wg.waves[0].count["v_mfma_f32_32x32x16_bf16"] == 2 * 64 * nt
    assert wg.waves[0].count.get("s_cmp_lg_u32", 0) == 2 * pairs      # (the loop's back-edge test: once per tile pair and block)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("cfg", [FOLD, EXACT], ids=["fold", "exact"])
def test_rescale_decided_in_the_branch_free_loop(cfg, tile):
    """a key aligned with a query in tile `tile` of ten forces the deferred rescale there: tiles 1 .. 6 are branch-free tiles of
    either parity (the decision jumps into the ordinary copy, which applies the rescale and returns to the loop head), 7 .. 9 are not"""
    _check(2, 256, 640, cfg=cfg, seed=7, spike=(17, 64 * tile + 5, 3.0), tol_o=1.2e-2)


@pytest.mark.parametrize("dma_mode,stores,order", [("early", "late", (3, 2, 1, 0)), ("late", "early", (0, 1, 2, 3)), ("late", "late", (2, 0, 3, 1))])
def test_branch_free_loop_ring_discipline(dma_mode, stores, order):
    _check(3, 256, 704, cfg=FOLD, dma_mode=dma_mode, stores=stores, order=order, seed=21)
    _check(2, 300, 520, cfg=EXACT, dma_mode=dma_mode, stores=stores, order=order, seed=22)


def test_round5_streams_are_still_generated_for_the_developer_library():
    """the A/B baselines of tools/p4p_streams_ab.py: same results as the product streams (same arithmetic, other loop)"""
    for dev, prod in (("R5_BF16_FOLD_L16", FOLD), ("R5_BF16_EXACT", EXACT)):
        _check(1, 256, 449, cfg=p4pgen.VARIANTS[dev], seed=3)
        assert not p4pgen.VARIANTS[dev].fastloop and prod.fastloop


CAUSAL_FOLD = p4pgen.VARIANTS["BF16_FOLD_L16_CAUSAL"]
CAUSAL_EXACT = p4pgen.VARIANTS["BF16_EXACT_CAUSAL"]


@pytest.mark.parametrize("H,R,C", [(1, 512, 512), (2, 768, 768), (1, 300, 400), (1, 700, 1000), (1, 256, 256), (1, 1280, 1280)])
def test_causal_pairs(H, R, C):
    """causal streams: tile count, per-wave traversal bound, first masked tile and the lanes' limits are computed per block INSIDE
    the stream; blocks arrive in pairs (long, short); odd block counts (768, 1280 rows) have a middle block that is its own pair"""
    wg, _, _ = _check(H, R, C, cfg=CAUSAL_EXACT, seed=10)
    counts = [w.count.get("v_mfma_f32_32x32x16_bf16", 0) for w in wg.waves]
    assert len(set(w.count["s_barrier"] for w in wg.waves)) == 1, "waves disagree on the barriers"
    if R == C and R % 256 == 0:
        assert counts[0] < counts[3], counts      # a wave stops multiplying at the diagonal of ITS rows


@pytest.mark.parametrize("dma_mode,stores,order", [("early", "late", (3, 2, 1, 0)), ("late", "early", (0, 1, 2, 3)), ("late", "late", (2, 0, 3, 1))])
def test_causal_skip_loop_keeps_the_ring_and_the_block_switch_going(dma_mode, stores, order):
    """waves whose rows end early only keep the barriers and their LDS-DMA share -- including the switch to the NEXT block's K / V
    and their part of its Q image -- going; DMA landing early / late, stores retiring early / late, waves in any order"""
    _check(2, 512, 512, cfg=CAUSAL_FOLD, dma_mode=dma_mode, stores=stores, order=order, seed=11)


def test_causal_spike_below_the_diagonal():
    wg, _, _ = _check(1, 512, 512, cfg=CAUSAL_FOLD, spike=(300, 200, 3.0), seed=12, tol_o=1.2e-2)


@pytest.mark.parametrize("cfg", [CAUSAL_FOLD, CAUSAL_EXACT], ids=["fold", "exact"])
def test_causal_branch_free_loop(cfg):
    """causal streams: the loop's end is per block AND per wave (its first masked tile); waves leave it at different tiles and keep
    meeting at the same barriers; rescales decided inside it in a long block"""
    assert cfg.fastloop
    _check(1, 1024, 1024, cfg=cfg, spike=(700, 200, 3.0), seed=12, tol_o=1.2e-2)
    _check(1, 1024, 1024, cfg=cfg, spike=(900, 330, 3.0), seed=13, tol_o=1.2e-2)
    wg, _, _ = _check(1, 768, 1280, cfg=cfg, seed=14, tol_o=8e-3)
    assert len(set(w.count["s_barrier"] for w in wg.waves)) == 1, "waves disagree on the barriers"


def _check_lengths(cfg, lens, cflag, R, C, seed=0, order_blocks=None, **kw):
    """causal ("geometry") streams with per-batch lengths: head h = a batch entry of lens[h] = (rows, keys) inside [R] x [C] arrays; the
    table holds the non-empty row blocks only (the C++ prologue compacts it), every entry its own lengths"""
    rng = np.random.default_rng(seed)
    H = len(lens)
    q, k, v = (p4psim.rand_bf16(s_, rng) for s_ in ((H, R, 128), (H, C, 128), (H, C, 128)))
    blocks = [(h, rb) for h in range(H) for rb in range((lens[h][0] + 255) // 256)]
    if order_blocks:
        blocks = order_blocks(blocks)
    O, L, wg, (om, lm) = p4psim.run_workgroup(q, k, v, blocks, cfg, lengths={h: lens[h] for h in range(H)}, cflag=cflag, **kw)
    for h, (Rb, Cb) in enumerate(lens):
        Oref, Lref = p4psim.reference(q[h, :Rb], k[h, :Cb], v[h, :Cb], causal=bool(cflag))
        dO, dL = np.abs(O[h, :Rb] - Oref).max(), np.abs(L[h, :Rb] - Lref).max()
        assert dO < 8e-3 and dL < (2e-2 + 5e-3) * max(1.0, np.abs(Lref).max() / 8), (h, Rb, Cb, dO, dL)
        # nothing beyond an entry's rows is written
        assert (om.reshape(H, R, 512)[h, Rb:] == 0xCD).all() and (lm.reshape(H, R, 2 if cfg.l16 else 4)[h, Rb:] == 0xCD).all(), (h, Rb)
    return wg


@pytest.mark.parametrize("cflag", [0, 1])
def test_per_batch_lengths_on_the_persistent_kernel(cflag):
    """round 6: the causal streams take rows / keys per table entry -- per-batch lengths run on the persistent kernel, with the causal
    mask (each entry's own diagonal offset, clamped at 0 when it has fewer keys than rows) or without it (the offset out of reach)"""
    lens = [(300, 520), (77, 100), (512, 64), (130, 130), (256, 700)]
    if cflag:
        lens.append((300, 200))    # fewer keys than rows: offset 0
    _check_lengths(CAUSAL_FOLD, lens, cflag, 512, 704, seed=41 + cflag)
    _check_lengths(CAUSAL_EXACT, lens[:3], cflag, 512, 704, seed=43 + cflag, dma_mode="early", stores="late", order=(3, 2, 1, 0))


def test_per_batch_lengths_long_entries_use_the_branch_free_loop():
    wg = _check_lengths(CAUSAL_FOLD, [(256, 1024), (300, 960)], 0, 512, 1024, seed=45, order_blocks=lambda b: b[::-1])
    assert wg.waves[0].count.get("s_cmp_lg_u32", 0) > 0


@pytest.mark.parametrize("C", [64, 128, 256, 449])
def test_merged_block_switch_experiment(C):
    """developer stream (slower on the GPU, kept as a record): the last tile's softmax finish beside the next block's first K Q^T.
    The model found the hazard its first form had -- no barrier between a wave's wait for its own pieces of K'(1) and the other
    waves' reads of them -- before any GPU run."""
    _check(3, 256, C, cfg=p4pgen.VARIANTS["BF16_FOLD_L16_MERGE"], seed=13, dma_mode="late", order=(0, 1, 2, 3))
    _check(2, 512, C, cfg=p4pgen.VARIANTS["BF16_FOLD_L16_MERGE"], seed=14, dma_mode="early", order=(3, 2, 1, 0))


@pytest.mark.parametrize("C", [128, 449])
def test_fused_tail_experiment(C):
    """developer stream (no gain on the GPU, kept as a record): the epilogue's per-block work dealt out between the last tile's
    P V products, which run in (head-dimension block, key step, row block) order"""
    _check(3, 256, C, cfg=p4pgen.VARIANTS["BF16_FOLD_L16_FUSE"], seed=15)
    _check(2, 300, C, cfg=p4pgen.VARIANTS["BF16_FOLD_L16_FUSE"], seed=16, dma_mode="early", stores="early", order=(3, 2, 1, 0))


def test_stream_file_is_current(built_library):
    """csrc/attn_fwd16_p4p_stream.inc is what tools/p4pgen.py generates"""
    path = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p4p_stream.inc")
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tmp:
        p4pgen.write_inc(tmp.name)
        assert open(path).read() == open(tmp.name).read(), "run python tools/p4pgen.py"


def test_stores_per_block_match_the_wait():
    """every block issues exactly NST vector-memory stores per wave: the vmcnt(NST) waits of the next block's first tiles then
    cover every LDS-DMA piece requested before them"""
    wg, _, _ = _check(3, 256, 128, seed=8)
    w = wg.waves[0]
    stores = sum(w.count.get(op, 0) for op in ("buffer_store_dwordx4", "buffer_store_dwordx2", "buffer_store_dword", "buffer_store_short"))
    assert stores == 3 * p4pgen.NST


# ---- round 6: column-parallel pieces on the persistent kernel (split streams) --------------------------------------------------------
@pytest.mark.parametrize("name", ["BF16_FOLD_SPLIT", "BF16_EXACT_SPLIT", "F16_FOLD_SPLIT", "F16_EXACT_SPLIT"])
@pytest.mark.parametrize("R,C,splits", [(256, 1024, 4), (512, 512, 2), (300, 1280, 2), (256, 256, 2)])
def test_column_parallel_pieces(name, R, C, splits):
    """split streams: a table entry is (row block, piece of the key range: whole multiples of two tiles); the un-normalised O^T and (m, l)
    of every piece land in the workspace slabs and the merge of attn_fwd_combine (restated in tools/p4psim.py) gives the attention of
    the whole key range.  One workgroup walks all pieces of all row blocks here, in an order that mixes them"""
    cfg = p4pgen.VARIANTS[name]
    rng = np.random.default_rng(20)
    f16 = cfg.dtype == "f16"
    H, D = 2, 128
    q, k, v = (p4psim.rand_bf16(s_, rng, f16=f16) for s_ in ((H, R, D), (H, C, D), (H, C, D)))
    nrb = (R + 255) // 256
    blocks = [(h, rb, sp) for sp in range(splits) for h in range(H) for rb in range(nrb)]
    O, L, wg, _ = p4psim.run_workgroup(q, k, v, blocks, cfg, D=D, splits=splits, dma_mode="late")
    for h in range(H):
        Oref, Lref = p4psim.reference(q[h], k[h], v[h], causal=False, f16=f16)
        dO, dL = np.abs(O[h] - Oref).max(), np.abs(L[h] - Lref).max()
        assert dO < 6e-3 and dL < 6e-3, (h, dO, dL)


# ---- round 6, developer streams: O = P V with lane = column (PCfg.orow) -- rows stored straight from the registers, no LDS trip.  Measured
# equal to the product streams (profiles/r06_p4p_epilogue.txt: the epilogue is bound by the store path, not by LDS); kept as the record
@pytest.mark.parametrize("name", ["BF16_FOLD_L16_OROW", "BF16_EXACT_OROW", "BF16_FOLD_L16_CAUSAL_OROW"])
def test_row_major_accumulators_give_the_same_bytes(name):
    """same instruction multiset inside the loop, the second products' operands exchanged: every byte of O and L equals the product
    stream's -- also behind a deferred rescale (the row's factor per REGISTER, by ds_bpermute_b32) and with D < 128, ragged rows / keys"""
    cfg = p4pgen.VARIANTS[name]
    base = p4pgen.VARIANTS[name[:-5]]
    assert cfg.orow and not base.orow and name not in p4pgen.PRODUCT_STREAMS
    C = 448
    for kw in (dict(spike=(17, 64 * 3 + 5, 3.0), tol_o=1.2e-2, stores="late", dma_mode="late", order=(3, 2, 1, 0)), dict(D=72, ld=128, R=300)):
        R = kw.pop("R", 256)
        a = _check(2, R, C, cfg=cfg, seed=31, **kw)
        b = _check(2, R, C, cfg=base, seed=31, **kw)
        assert (a[1][0] == b[1][0]).all() and (a[1][1] == b[1][1]).all()
        if "spike" in kw:
            wg = a[0]
    assert wg.waves[0].count["buffer_store_dword"] >= 2 * 128 and not wg.waves[0].count.get("buffer_store_dwordx4")
    assert wg.waves[0].count.get("ds_bpermute_b32", 0) >= 2 * 32 + 32      # two epilogues + at least one rescale

"""CPU restatement of the LDS address maps of the 256 < D <= 384 kernels (attn_fwd16_wide.h, attn_bwd16.h attn_dq16 with 32-key tiles,
attn_dkv16_wide.h): every 16-byte access of sixteen consecutive lanes must land on sixteen different 16-byte slots of the 256-byte
bank row (64 banks x 4 bytes) -- what the counter pass of round 6 found violated (SQ_LDS_BANK_CONFLICT 22-46 % of the LDS cycles,
profiles/r06_fwdbwd_bf16_d384_mixed_summary_before_lds_fix.txt) and the layouts below fixed (0 afterwards).  The formulas are the
kernels'; a change there that brings the conflicts back fails here instead of showing up as LDS cycles."""
import pytest


def slots(addresses):
    return [(a // 16) % 16 for a in addresses]


def conflict_free(addresses):
    s = slots(addresses)
    return len(set(s)) == len(s)


@pytest.mark.parametrize("D", [320, 384])
def test_padded_row_major_images_are_read_without_conflicts(D):
    """attn_fwd16_wide.h: K rows of D * 2 + 16 bytes, lane (q, hi) reads chunk 2 t + hi of row q (kread + 32 t); attn_dq16 (BC = 32): the V
    image, same pitch and read.  Sixteen consecutive rows start on sixteen different slots because the pitch is an ODD number of chunks."""
    rowb = D * 2 + 16
    assert (rowb // 16) % 2 == 1
    for t in range(D // 16):
        for hi in range(2):
            for q0 in (0, 16):
                assert conflict_free([q * rowb + (2 * t + hi) * 16 for q in range(q0, q0 + 16)])
    # the unpadded pitch with the small buckets' XOR of the low chunk bits (what round 6 first shipped) does conflict
    bad = [q * (D * 2) + ((0 ^ (q & 7)) << 4) for q in range(16)]
    assert not conflict_free(bad)


@pytest.mark.parametrize("D", [320, 384])
def test_blocked_images_are_written_without_conflicts(D):
    """[D/32][32 rows][64 bytes] images (attn_dkv16_wide.h: Q and dO; attn_dq16 BC = 32: K; attn_fwd16_wide.h: V): a wave stages eight rows x a
    pair of d-blocks, sixteen lanes = four rows x the four chunks of one d-block; every (row, chunk) of the tile is written exactly once"""
    ndb, cpr, br = D // 32, D // 8, 32
    seen = set()
    for base in range(0, br * cpr, 16):           # groups of sixteen consecutive lanes (id = tid + i * 256)
        addrs = []
        for ident in range(base, base + 16):
            l, unit = ident & 63, ident >> 6
            dbp, rg = unit % (ndb // 2), unit // (ndb // 2)
            srow = 8 * rg + 4 * (l >> 5) + ((l >> 2) & 3)
            sc = 4 * (2 * dbp + ((l >> 4) & 1)) + (l & 3)
            assert 0 <= srow < br and 0 <= sc < cpr
            seen.add((srow, sc))
            addrs.append(((sc >> 2) * br + srow) * 64 + (((sc & 3) ^ ((srow >> 2) & 3)) * 16))
        assert conflict_free(addrs), base
    assert len(seen) == br * cpr
    # the forward kernel's V staging: thread tid owns key vrow, chunks vc0 + 8 i (attn_fwd16_wide.h)
    for i in range(cpr // 8):
        for base in range(0, 256, 16):
            addrs = []
            for tid in range(base, base + 16):
                vrow = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1) + ((tid >> 2) & 3)
                vc0 = 4 * ((tid >> 4) & 1) + (tid & 3)
                addrs.append(((vc0 >> 2) * 32 + vrow) * 64 + (vc0 & 3) * 16 + 4096 * i)
            assert conflict_free(addrs)
    # consecutive chunks of ONE row on consecutive lanes (attn_dkv16_rs.h's map, what these kernels started from): four slots
    old = [((sc >> 2) * br + 5) * 64 + (((sc & 3) ^ 1) * 16) for sc in range(16)]
    assert len(set(slots(old))) == 4


def test_row_fragments_of_the_blocked_images_are_read_without_conflicts():
    """lane (row, hi) reads chunk 2 (t & 1) + hi of d-block t >> 1, swizzled by (row >> 2) & 3 (attn_dkv16_wide.h fr0 / fr1, attn_dq16 kfread)"""
    for t in range(4):
        for hi in range(2):
            for r0 in (0, 16):
                assert conflict_free([((t >> 1) * 32 + r) * 64 + (((2 * (t & 1) + hi) ^ ((r >> 2) & 3)) * 16) for r in range(r0, r0 + 16)])

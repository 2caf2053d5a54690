"""The index maps of the FP32 kernels (csrc/attn_f32.h), restated in numpy and checked against what the matrix instruction needs:
the LDS image an LDS-DMA of the permuted global addresses produces, the two read patterns (which element every lane gets, that
no two lanes of a 16-lane group hit the same 16-byte bank group), the permuted contraction / head indices and the output rows a
lane stores.  CPU only: a change of the swizzle or of a permutation that the GPU parity tests would report as wrong numbers fails
here with the index that is off."""
import numpy as np
import pytest

BT = 32


def crow(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def geo(DP):
    return dict(ROWB=DP * 4, CPR=DP // 4, TILE=BT * DP * 4, NI=DP // 32, NDB=DP // 32, NG=DP // 8, RB2=DP // 32 * 4)


def dma_image(DP):
    """tile[row][float] -> LDS byte image, as Stager::init / issue lay it out: instruction i of wave w writes the 1 KiB at
    (w * NI + i) * 1024, lane l its 16 bytes at + 16 l, fetched from row r = p / CPR, chunk (p % CPR) ^ (r & 15)."""
    g = geo(DP)
    tile = np.arange(BT * DP, dtype=np.int64).reshape(BT, DP)          # element id = row * DP + d
    lds = np.full(g["TILE"] // 4, -1, np.int64)                        # in floats
    for w in range(4):
        for i in range(g["NI"]):
            for lane in range(64):
                p = (w * g["NI"] + i) * 64 + lane
                r, c = p // g["CPR"], (p % g["CPR"]) ^ ((p // g["CPR"]) & 15)
                lds[p * 4:p * 4 + 4] = tile[r, 4 * c:4 * c + 4]
    assert (lds >= 0).all()
    return tile, lds


@pytest.mark.parametrize("DP", [64, 128])
def test_first_pattern_reads_the_permuted_contraction_index(DP):
    """lane (i, hi), group T: ds_read_b128 at first[T & 7] + (T >> 3) * 256 returns X[i][8 T + 4 hi .. + 3] -- the four contraction
    steps t = 4 T .. 4 T + 3 of the lane's k-half, matching the cached fragment f[4 T + j] = Y[row][8 T + 4 hi + j]"""
    g = geo(DP)
    tile, lds = dma_image(DP)
    for T in range(g["NG"]):
        groups = {}
        for lane in range(64):
            i, hi = lane & 31, lane >> 5
            f = i * g["ROWB"] + ((hi ^ (i & 15)) << 4)
            addr = (f ^ ((T & 7) << 5)) + (T >> 3) * 256
            assert addr % 16 == 0 and addr < g["TILE"]
            got = lds[addr // 4:addr // 4 + 4]
            assert (got == tile[i, 8 * T + 4 * hi:8 * T + 4 * hi + 4]).all(), (T, lane)
            groups.setdefault(lane >> 4, []).append((addr >> 4) & 15)
        for lanes, banks in groups.items():   # sixteen lanes, sixteen different 16-byte bank groups of the 256 bytes LDS serves per clock
            assert len(set(banks)) == 16, (T, lanes, banks)


@pytest.mark.parametrize("DP", [64, 128])
def test_second_pattern_reads_rows_of_the_permuted_head_index(DP):
    """lane (i, hi), step t: NDB floats of row crow(t, hi) at column NDB i -- accumulator block db of the lane holds d = NDB i + db"""
    g = geo(DP)
    tile, lds = dma_image(DP)
    for t in range(16):
        ct, cidx, rowoff = (t & 3) | (8 * ((t >> 2) & 1)), (t & 3) + 4 * ((t >> 2) & 1), (t & 3) + 8 * (t >> 2)
        assert ((cidx & 3) | (8 * (cidx >> 2))) == ct
        banks = {}
        for lane in range(64):
            i, hi = lane & 31, lane >> 5
            byte = i * g["RB2"]
            s = 4 * hi * g["ROWB"] + ((((byte >> 4) ^ (4 * hi)) << 4) | (byte & 15))
            addr = (s ^ (ct << 4)) + rowoff * g["ROWB"]
            assert addr + g["RB2"] <= g["TILE"]
            got = lds[addr // 4:addr // 4 + g["NDB"]]
            assert (got == tile[crow(t, hi), g["NDB"] * i:g["NDB"] * i + g["NDB"]]).all(), (t, lane)
            banks.setdefault((lane >> 4) if DP == 128 else (lane >> 5), []).append(addr >> 2)
        for grp, words in banks.items():      # the lanes of a group read consecutive, non-overlapping words of one row
            assert len(set(w >> 2 if DP == 128 else w >> 1 for w in words)) == len(words), (t, grp)


@pytest.mark.parametrize("DP", [64, 128])
def test_the_two_products_compose_to_the_attention_products(DP):
    """S^T[key][row] = sum over (T, j, hi) of K[key][d] Q[row][d] with d = 8 T + 4 hi + j covers every d once; O^T block db, register r
    of lane (row, hi) is O[row][NDB crow(r, hi) + db]: every column once"""
    g = geo(DP)
    ds = sorted(8 * T + 4 * hi + j for T in range(g["NG"]) for hi in range(2) for j in range(4))
    assert ds == list(range(DP))
    cols = sorted(g["NDB"] * crow(r, hi) + db for r in range(16) for hi in range(2) for db in range(g["NDB"]))
    assert cols == list(range(DP))
    # the key a step of a second product contracts on lane half hi is the key whose score sits in register t of that half
    assert sorted(crow(t, hi) for t in range(16) for hi in range(2)) == list(range(32))


def test_slices_of_l_and_d_line_up_with_the_score_registers():
    """dK/dV: the accumulator of S = Q K^T starts at L[row0 + crow(r, hi)]: four ds_read_b128 at (8 g + 4 hi) floats"""
    for hi in range(2):
        rows = [8 * g + 4 * hi + j for g in range(4) for j in range(4)]
        assert rows == [crow(r, hi) for r in range(16)]

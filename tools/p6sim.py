#!/usr/bin/env python3
"""Lane-exact model of the D <= 64 persistent forward stream (tools/p6gen.py) on p4psim's workgroup: `run_workgroup` restates the
C++ prologue of attn_fwd16_p6 (block table, lane constants, scalar inputs)."""
import numpy as np

import p6gen
from p4psim import GlobalMem, PWorkgroup
from p4sim import h16_to_f32, rand_bf16, reference, f32_to_h16  # noqa: F401


def run_workgroup(q, k, v, blocks, cfg, D=64, dma_mode="late", stores="late", order=(0, 1, 2, 3), stream=None, ld=None, splits=1,
                  lengths=None, cflag=None):
    # lengths (causal / "geometry" streams): {head: (rows, keys)} of the head's batch entry; cflag: 1 = causal mask, 0 = lengths only
    """One persistent workgroup over `blocks` = [(head, row block), ...].  q [H][R][D], k / v [H][C][D] uint16 bit patterns.
    Returns O [H][R][D] float32 (the 16-bit patterns as float32 when cfg.o16), L [H][R] float32 (log2 units), the workgroup."""
    f16 = cfg.dtype == "f16"
    H, R, _ = q.shape
    C = k.shape[1]
    ldq = ldk = ldv = ldo = ld or D
    instrs = stream if stream is not None else p6gen.Stream6(cfg).build()
    mem = GlobalMem()

    def padded(x, width):
        out = np.zeros(x.shape[:-1] + (width,), x.dtype)
        out[..., :x.shape[-1]] = x
        return out
    qm, km, vm = (np.ascontiguousarray(padded(x, ldq)).reshape(-1).view(np.uint8).copy() for x in (q, k, v))
    osz = 2 if cfg.o16 else 4
    lsz = 2 if cfg.l16 else 4
    split = bool(getattr(cfg, "split", 0))
    if split:      # blocks = [(head, row block, piece)]: wsO [splits][H][R][D] fp32, wsML [splits][H][R][2]
        assert C % (256 * splits) == 0 and ldo == D
        lsz, piece = 8, C // splits
        om = np.full(splits * H * R * D * 4, 0xCD, np.uint8)
        lm = np.full(splits * H * R * 8, 0xCD, np.uint8)
    else:
        om = np.full(H * R * ldo * osz, 0xCD, np.uint8)
        lm = np.full(H * R * lsz, 0xCD, np.uint8)
    qb, kb, vb, ob, lb = (mem.alloc(x) for x in (qm, km, vm, om, lm))
    wg = PWorkgroup(instrs, mem, dma_mode, stores)
    table = np.zeros((len(blocks), 16), np.uint32)
    for n, blk in enumerate(blocks):
        h, rblk = blk[0], blk[1]
        if split:
            sp = blk[2]
            bases = (qb + h * R * ldq * 2, kb + (h * C + sp * piece) * ldk * 2, vb + (h * C + sp * piece) * ldv * 2,
                     ob + (sp * H + h) * R * D * 4, lb + (sp * H + h) * R * 8)
        else:
            bases = (qb + h * R * ldq * 2, kb + h * C * ldk * 2, vb + h * C * ldv * 2, ob + h * R * ldo * osz, lb + h * R * lsz)
        for i, a in enumerate(bases):
            table[n, 2 * i], table[n, 2 * i + 1] = a & 0xFFFFFFFF, a >> 32
        table[n, 10] = rblk * 256
        table[n, 11], table[n, 12] = (lengths or {}).get(h, (R, C))
    tb = table.reshape(-1).view(np.uint8)
    wg.lds[p6gen.TABLE:p6gen.TABLE + tb.size] = tb
    nt = max(4, ((C + 63) // 64 + 3) // 4 * 4)     # a multiple of the loop body's four tiles
    Ck = C
    if split:
        Ck, nt = piece, piece // 64
    scale2 = float(np.float32(1.44269504089) * np.float32(1.0 / np.sqrt(np.float32(D))))
    lane = np.arange(64)
    qq, hi = lane & 31, lane >> 5
    n16 = lane & 15
    ldq2, ldk2, ldv2 = ldq * 2, ldk * 2, ldv * 2
    OOB = p6gen.OOB
    for w in wg.waves:
        wave = w.id
        kv, qv = [], []
        for par in range(2):
            c = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7)
            kv.append(np.where(c * 8 < D, (lane >> 3) * ldk2 + c * 16, OOB).astype(np.uint32))
            qv.append(np.where(c * 8 < D, (lane >> 3) * ldq2 + c * 16, OOB).astype(np.uint32))
        vc = (wave >> 1) * 4 + (lane & 3)
        vv = np.where(vc * 8 < D, ((lane >> 2) + 32 * (wave & 1)) * ldv2 + vc * 16, OOB).astype(np.uint32)
        w.vn.update({
            "kbase": (qq * 128 + ((hi ^ ((qq >> 1) & 7)) << 4)).astype(np.uint32),
            "vbase": (p6gen.VRING_BASE + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2).astype(np.uint32),
            "lim0": (Ck - 1 - 4 * hi).astype(np.int64).astype(np.uint32), "lim1": (Ck - 1 - 4 * hi).astype(np.int64).astype(np.uint32),
            "vv": vv, "lv": np.where(hi == 0, qq * lsz, OOB).astype(np.uint32),
            "ewa": (qq * 128 + ((hi ^ (qq & 7)) << 4)).astype(np.uint32),
            "era": ((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4)).astype(np.uint32),
            "kv0": kv[0], "kv1": kv[1], "qv0": qv[0], "qv1": qv[1],
            "qlane": qq.astype(np.uint32), "hi4": (4 * hi).astype(np.uint32),
        })
        for db in range(2):
            col = 32 * db + 4 * (lane & 7)
            w.vn["ov%d" % db] = np.where(col < D, (lane >> 3) * ldo * osz + col * osz, OOB).astype(np.uint32)
        w.sn.update({"nt": nt, "maskfrom": Ck // 64, "scale2": scale2, "kinc": 64 * ldk2, "vinc": 64 * ldv2,
                     "ldsk": wave * 2048, "ldsv": p6gen.VRING_BASE + (wave >> 1) * 4096 + (wave & 1) * 2048,
                     "ldsq": p6gen.QIMG + wave * 8192, "qrel": p6gen.QIMG + wave * 8192, "ldsst": p6gen.STAGE + wave * 4096,
                     "nblk": len(blocks), "tbl": p6gen.TABLE, "wave64": wave * 64,
                     "ldq2": ldq2, "ldo": ldo * osz, "nrecq": R * ldq2, "nreck": Ck * ldk2, "nrecv": Ck * ldv2,
                     "nreco": R * ldo * osz, "nrecl": R * lsz, "cflag": int(bool(cfg.causal)) if cflag is None else int(cflag)})
    wg.run(order)
    for w in wg.waves:
        assert not w.lds_q, "LDS reads left in flight"
        wg.retire_vm(w, 0)
    if split:      # the merge of attn_fwd_combine, restated: m* = max m_s, w_s = 2^(m_s - m*), O = sum w_s O_s / sum w_s l_s, L = m* + log2 l*
        Os = om.view(np.float32).reshape(splits, H, R, D).astype(np.float64)
        ml = lm.view(np.float32).reshape(splits, H, R, 2).astype(np.float64)
        mstar = ml[..., 0].max(axis=0)
        wgt = np.exp2(ml[..., 0] - mstar[None])
        lstar = (wgt * ml[..., 1]).sum(axis=0)
        O = ((wgt[..., None] * Os).sum(axis=0) / lstar[..., None]).astype(np.float32)
        L = (mstar + np.log2(lstar)).astype(np.float32)
        return O, L, wg, (om, lm)
    if cfg.o16:
        O = h16_to_f32(om.view(np.uint16).astype(np.uint32), f16).reshape(H, R, ldo)[..., :D]
    else:
        O = om.view(np.float32).reshape(H, R, ldo)[..., :D]
    L = lm.view(np.float16).astype(np.float32).reshape(H, R) if cfg.l16 else lm.view(np.float32).reshape(H, R)
    return O, L, wg, (om, lm)

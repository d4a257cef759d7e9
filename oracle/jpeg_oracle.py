"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reconstruction half of a baseline JPEG decoder -- dequantisation, the
accurate integer 8x8 inverse DCT, chroma upsampling and YCbCr -> RGB -- with the integer arithmetic of IJG libjpeg 6b / libjpeg-turbo
(the decoder PIL links; a third-party dependency of the reference's frame loading, eval/video_score_stream_demo.py:212-225 through
decord / PIL; not vendored in /root/reference): jidctint.c `jpeg_idct_islow`, jdsample.c `h2v1_fancy_upsample` /
`h2v2_fancy_upsample` (+ jdmainct.c's edge-row replication), jdcolor.c `build_ycc_rgb_table` / `ycc_rgb_convert`.

Pinned: tests/test_oracle_golden.py-style CPU test `tests/test_jpeg_cpu.py` compares it BYTE FOR BYTE with PIL's own output on
generated images (every sampling mode, odd sizes, qualities, restart intervals), with the coefficients coming from the C ABI's host
entropy decoder (sm_jpeg_decode_coefs).  Only tests may import this module."""
import numpy as np

F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137, f1_961=16069,
         f2_053=16819, f2_562=20995, f3_072=25172)


def _idct_1d(v, shift):
    """v: int64 [..., 8] along the last axis -> the islow butterfly, descaled by `shift` with rounding"""
    i0, i1, i2, i3, i4, i5, i6, i7 = [v[..., k] for k in range(8)]
    z1 = (i2 + i6) * F["f0_541"]
    tmp2 = z1 + i6 * (-F["f1_847"])
    tmp3 = z1 + i2 * F["f0_765"]
    tmp0 = (i0 + i4) << 13
    tmp1 = (i0 - i4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F["f1_175"]
    t0, t1, t2, t3 = t0 * F["f0_298"], t1 * F["f2_053"], t2 * F["f3_072"], t3 * F["f1_501"]
    z1, z2, z3, z4 = z1 * -F["f0_899"], z2 * -F["f2_562"], z3 * -F["f1_961"] + z5, z4 * -F["f0_390"] + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    r = 1 << (shift - 1)
    out = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([(o + r) >> shift for o in out], axis=-1)


def idct_blocks(coefs, qt):
    """coefs int16 [nb, 64] (natural order), qt uint16 [64] -> samples uint8 [nb, 8, 8]"""
    x = coefs.astype(np.int64).reshape(-1, 8, 8) * qt.astype(np.int64).reshape(1, 8, 8)
    ws = _idct_1d(x.transpose(0, 2, 1), 11).transpose(0, 2, 1)        # pass 1: down the columns
    out = _idct_1d(ws, 18)                                            # pass 2: along the rows
    i = out & 0x3FF                                                   # the range-limit table of jdmaster.c, centred on 128
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def planes_from_coefs(coefs, qt, info):
    """one frame's coefficient image -> list of component sample planes (padded to whole MCUs)"""
    planes = []
    for c in range(info["ncomp"]):
        bx, by = info["blocks_x"][c], info["blocks_y"][c]
        blk = idct_blocks(coefs[info["coef_offset"][c]: info["coef_offset"][c] + bx * by * 64].reshape(-1, 64), qt[c])
        planes.append(blk.reshape(by, bx, 8, 8).transpose(0, 2, 1, 3).reshape(by * 8, bx * 8))
    return planes


def _h2v1(p, cw, W):
    p = p[:, :cw].astype(np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    even = (p * 3 + left + 1) >> 2
    odd = (p * 3 + right + 2) >> 2
    even[:, 0] = p[:, 0]
    odd[:, -1] = p[:, -1]
    out = np.empty((p.shape[0], 2 * cw), np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:, :W]


def _h2v2(p, cw, ch, W, H):
    p = p[:ch, :cw].astype(np.int32)
    up = np.concatenate([p[:1], p[:-1]], axis=0)            # context row above the first real row = that row (jdmainct.c)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)            # ... below the last REAL row = that row
    rows = np.empty((2 * ch, cw), np.int32)
    rows[0::2], rows[1::2] = p * 3 + up, p * 3 + dn        # column sums: 3 * nearer row + further row
    last = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
    nxt = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
    even = (rows * 3 + last + 8) >> 4
    odd = (rows * 3 + nxt + 7) >> 4
    even[:, 0] = (rows[:, 0] * 4 + 8) >> 4
    odd[:, -1] = (rows[:, -1] * 4 + 7) >> 4
    out = np.empty((2 * ch, 2 * cw), np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:H, :W]


def reconstruct(coefs, qt, info):
    """-> RGB uint8 [H, W, 3]"""
    W, H = info["width"], info["height"]
    planes = planes_from_coefs(coefs, qt, info)
    Y = planes[0][:H, :W].astype(np.int32)
    if info["ncomp"] == 1:
        return np.repeat(Y[:, :, None], 3, axis=2).astype(np.uint8)
    hs, vs = info["hs"][0], info["vs"][0]
    cw, ch = (W + hs - 1) // hs, (H + vs - 1) // vs
    if (hs, vs) == (1, 1):
        cb, cr = planes[1][:H, :W].astype(np.int32), planes[2][:H, :W].astype(np.int32)
    elif (hs, vs) == (2, 1):
        cb, cr = _h2v1(planes[1][:H], cw, W), _h2v1(planes[2][:H], cw, W)
    elif (hs, vs) == (2, 2):
        cb, cr = _h2v2(planes[1], cw, ch, W, H), _h2v2(planes[2], cw, ch, W, H)
    else:
        raise ValueError(f"sampling {hs}x{vs}")
    cb, cr = cb - 128, cr - 128
    r = Y + ((91881 * cr + 32768) >> 16)
    g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = Y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=2), 0, 255).astype(np.uint8)

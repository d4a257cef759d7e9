"""JPEG test material: generated with PIL (the encoder is irrelevant to what is checked: the DECODE must equal PIL's decode)."""
import ctypes as C
import io

import numpy as np


def test_image(w, h, seed, gray=False):
    """smooth structure + edges + noise: every coefficient band and both chroma planes are exercised"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + seed) * np.cos(yy / 7.0), 128 + 110 * np.sin((xx + yy) / 13.0), 255.0 * ((xx // 11 + yy // 5) % 2)], axis=2)
    img = np.clip(base + rng.normal(0, 18, (h, w, 3)), 0, 255).astype(np.uint8)
    return img[:, :, 0] if gray else img


def encode(img, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **kw)
    return buf.getvalue()


def pil_decode(b):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))


def strip_dht(b):
    out, pos = b[:2], 2
    while b[pos + 1] != 0xDA:
        seg = 2 + int.from_bytes(b[pos + 2:pos + 4], "big")
        if b[pos + 1] != 0xC4:
            out += b[pos:pos + seg]
        pos += seg
    return out + b[pos:]


CASES = [   # (w, h, gray, save kwargs)
    (336, 336, False, dict(quality=75, subsampling=2)),
    (336, 336, False, dict(quality=90, subsampling=0)),
    (336, 336, False, dict(quality=50, subsampling=1)),
    (47, 33, False, dict(quality=85, subsampling=2)),
    (33, 47, False, dict(quality=85, subsampling=1)),
    (17, 9, False, dict(quality=95, subsampling=0)),
    (640, 360, False, dict(quality=30, subsampling=2)),
    (101, 77, True, dict(quality=80)),
    (250, 130, False, dict(quality=75, subsampling=2, restart_marker_blocks=7)),
    (250, 130, False, dict(quality=75, subsampling=0, restart_marker_rows=1)),
    (64, 48, False, dict(quality=100, subsampling=2)),
    (320, 240, False, dict(quality=75, subsampling=2, optimize=True)),
]


def host_coefs(lib, b):
    """the C ABI's host half: geometry + entropy decode -> (info dict, coefs int16, qt uint16 [3, 64])"""
    from streammind_amd import _lib
    info = lib.sm_jpeg_info.argtypes[2]._type_()          # the struct type this handle was typed with (the package may be re-imported by alias tests)
    buf = (C.c_ubyte * len(b)).from_buffer_copy(b)
    _lib.check(lib.sm_jpeg_info(buf, len(b), C.byref(info)), "sm_jpeg_info")
    coefs = np.zeros(info.coef_count, np.int16)
    qt = np.zeros((3, 64), np.uint16)
    _lib.check(lib.sm_jpeg_decode_coefs(buf, len(b), C.byref(info), coefs.ctypes.data, qt.ctypes.data), "sm_jpeg_decode_coefs")
    d = dict(width=info.width, height=info.height, ncomp=info.ncomp, hs=list(info.hs), vs=list(info.vs), blocks_x=list(info.blocks_x),
             blocks_y=list(info.blocks_y), coef_offset=list(info.coef_offset), coef_count=info.coef_count)
    return d, coefs, qt

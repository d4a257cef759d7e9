"""f2 codec half on CPU: the C ABI's host entropy decoder (sm_jpeg_info / sm_jpeg_decode_coefs -- plain host code, no GPU) and the
oracle's numpy restatement of libjpeg's reconstruction, together against PIL (libjpeg-turbo) BYTE FOR BYTE.  This pins both: a wrong
coefficient or a wrong rounding anywhere shows as a byte."""
import numpy as np
import pytest

from oracle import jpeg_oracle as J
from tests import util_jpeg as U


@pytest.fixture(scope="module")
def lib():
    from streammind_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("case", range(len(U.CASES)))
def test_host_decoder_and_oracle_equal_pil(lib, case):
    w, h, gray, kw = U.CASES[case]
    b = U.encode(U.test_image(w, h, case, gray), **kw)
    info, coefs, qt = U.host_coefs(lib, b)
    assert (info["width"], info["height"]) == (w, h)
    got = J.reconstruct(coefs, qt, info)
    want = U.pil_decode(b)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} bytes differ, max {int(np.abs(got.astype(int) - want).max())}"


def test_implied_huffman_tables_and_errors(lib):
    """Motion-JPEG frames without DHT segments (AVI1 convention) decode with the standard tables; what the decoder does not
    implement is refused with the reason (the caller keeps PIL for those)."""
    from streammind_amd import _lib
    import ctypes as C
    b = U.encode(U.test_image(96, 64, 3), quality=75, subsampling=2, optimize=False)
    bare = U.strip_dht(b)
    assert b"\xff\xc4" not in bare[:bare.find(b"\xff\xda")]
    info, coefs, qt = U.host_coefs(lib, bare)
    assert np.array_equal(J.reconstruct(coefs, qt, info), U.pil_decode(b))
    prog = U.encode(U.test_image(96, 64, 3), quality=75, progressive=True)
    inf = lib.sm_jpeg_info.argtypes[2]._type_()
    buf = (C.c_ubyte * len(prog)).from_buffer_copy(prog)
    with pytest.raises(_lib.StreamMindHipError, match="progressive"):
        _lib.check(lib.sm_jpeg_info(buf, len(prog), C.byref(inf)), "sm_jpeg_info")
    with pytest.raises(_lib.StreamMindHipError, match="not a JPEG"):
        _lib.check(lib.sm_jpeg_info((C.c_ubyte * 8)(*b"notjpeg!"), 8, C.byref(inf)), "sm_jpeg_info")
    trunc = b[:len(b) // 2]
    tb = (C.c_ubyte * len(trunc)).from_buffer_copy(trunc)
    _lib.check(lib.sm_jpeg_info(tb, len(trunc), C.byref(inf)))
    out = np.zeros(inf.coef_count, np.int16)
    q = np.zeros((3, 64), np.uint16)
    rc = lib.sm_jpeg_decode_coefs(tb, len(trunc), C.byref(inf), out.ctypes.data, q.ctypes.data)      # truncated stream: zeros are fed, never a crash
    assert rc in (0, -1, -2, -3)


def test_golden_g19_jpeg_fixture(lib):
    """committed fixture (oracle/make_jpeg_golden.py): JPEG byte strings + the frames PIL / libjpeg-turbo decoded them to when the fixture was
    minted -- the host entropy decoder + the oracle reproduce them byte for byte, whatever PIL this machine has"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g19_jpeg.npz"))
    for i in range(int(g["n"])):
        info, coefs, qt = U.host_coefs(lib, bytes(g[f"jpeg{i}"]))
        assert np.array_equal(J.reconstruct(coefs, qt, info), g[f"rgb{i}"]), i
    info, coefs, qt = U.host_coefs(lib, bytes(g["jpeg_bare"]))
    assert np.array_equal(J.reconstruct(coefs, qt, info), g["rgb0"])

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


def _decode_rc(lib, b):
    """sm_jpeg_info + sm_jpeg_decode_coefs on hostile bytes: returns the status codes, never raises"""
    import ctypes as C
    inf = lib.sm_jpeg_info.argtypes[2]._type_()
    buf = (C.c_ubyte * len(b)).from_buffer_copy(b)
    rc = lib.sm_jpeg_info(buf, len(b), C.byref(inf))
    if rc:
        return rc, None
    out = np.zeros(max(inf.coef_count, 1) + 64, np.int16)
    guard = out[inf.coef_count:].copy()
    q = np.zeros((3, 64), np.uint16)
    rc2 = lib.sm_jpeg_decode_coefs(buf, len(b), C.byref(inf), out.ctypes.data, q.ctypes.data)
    assert np.array_equal(out[inf.coef_count:], guard), "coefficients written past coef_count"
    return rc, rc2


def test_hostile_streams_are_refused_not_crashed(lib):
    """Advisor (round 4, high): an over-subscribed DHT used to index the 9-bit lookahead table ~130 KB past its end; a second SOF
    between the scans of a per-component file changed the geometry under the coefficient layout.  Both are now SM_EINVAL."""
    from streammind_amd import _lib
    b = U.encode(U.test_image(96, 64, 3), quality=75, subsampling=2, optimize=False)
    # --- over-subscribed Huffman table: bits[1] = 255 (no prefix code has 255 one-bit codes), and the milder bits[1] = 3
    dht = b.find(b"\xff\xc4")
    for bad in (255, 3):
        m = bytearray(b)
        m[dht + 5] = bad                              # marker(2) length(2) Tc/Th(1) -> bits[1]
        rc, rc2 = _decode_rc(lib, bytes(m))
        assert (rc2 if rc == 0 else rc) != 0, bad
        assert "Huffman" in lib.sm_last_error().decode(errors="replace") or "jpeg" in lib.sm_last_error().decode(errors="replace")
    # --- a second (larger) frame header behind the first: refused whether it comes before the first scan or between two scans
    sof = b.find(b"\xff\xc0")
    seg = b[sof:sof + 2 + int.from_bytes(b[sof + 2:sof + 4], "big")]
    big = bytearray(seg)
    big[5:7] = (4000).to_bytes(2, "big"); big[7:9] = (4000).to_bytes(2, "big")
    sos = b.find(b"\xff\xda")
    rc, rc2 = _decode_rc(lib, b[:sos] + bytes(big) + b[sos:])
    assert rc == 0 and rc2 != 0 and "second frame header" in lib.sm_last_error().decode(errors="replace")
    # per-component scans (three SOS segments): build one from the interleaved coefficients is beyond a test's means -- PIL cannot write
    # non-interleaved baseline -- so the rescan path is exercised by the marker walk alone: a SOF placed where the next scan header is
    # looked for (behind the entropy segment) must be refused as well when components are still missing.  A grey image has one scan, so
    # use the colour file truncated to its first MCU row with a SOF appended: the walker meets EOI-less data, then the SOF.
    rc, rc2 = _decode_rc(lib, b[:-2] + bytes(big) + b"\xff\xd9")
    assert rc == 0 and rc2 in (0, -1, -2, -3)         # all components were done after the single interleaved scan: nothing is re-parsed
    # --- byte-mutation fuzz over headers and entropy data: any status, no crash, nothing written past coef_count
    rng = np.random.default_rng(0)
    for t in range(300):
        m = bytearray(b)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(2, len(m)))] = int(rng.integers(0, 256))
        _decode_rc(lib, bytes(m))


def test_scan_prepare_is_markers_only_and_says_why_not():
    """sm_jpeg_scan_prepare (the host half of the GPU entropy decode): restart interval, interval count, segment offset and tables of a frame with restart
    markers; a file without DRI is one serial stream (restart 0, no intervals: the self-synchronising decode takes it); progressive files and per-component
    scans are refused with the reason (the caller keeps the host decoder)."""
    import ctypes as C
    from streammind_amd import _lib
    lib = _lib.load()
    img = U.test_image(200, 120, 3)
    b = U.encode(img, quality=85, subsampling=2, restart_marker_rows=1)
    inf = lib.sm_jpeg_info.argtypes[2]._type_()               # (the structure classes of the library object's own signatures: the ABI test re-imports the module)
    _lib.check(lib.sm_jpeg_info(C.cast(C.c_char_p(b), C.c_void_p), len(b), C.byref(inf)))
    sc = lib.sm_jpeg_scan_prepare.argtypes[3]._type_()
    _lib.check(lib.sm_jpeg_scan_prepare(C.cast(C.c_char_p(b), C.c_void_p), len(b), C.byref(inf), C.byref(sc)))
    assert sc.restart == inf.mcus_x and sc.n_intervals == inf.mcus_y and sc.ncomp == 3
    assert b[sc.scan_offset - 2 - 12:sc.scan_offset - 12] == b"\xff\xda" and sc.scan_offset + sc.scan_len == len(b)
    assert b.count(b"\xff\xd0") + sum(b.count(bytes([0xFF, 0xD0 + i])) for i in range(1, 8)) == sc.n_intervals - 1
    # luminance DC table of Annex K / libjpeg: the 2-bit code 00 is category 0
    assert sc.dc[0].fast[0] >> 8 == 2 and sc.dc[0].fast[0] & 0xFF == 0
    plain = U.encode(img, quality=85)
    _lib.check(lib.sm_jpeg_scan_prepare(C.cast(C.c_char_p(plain), C.c_void_p), len(plain), None, C.byref(sc)))
    assert sc.restart == 0 and sc.n_intervals == 0 and sc.ncomp == 3 and sc.scan_offset + sc.scan_len == len(plain)
    for bad, why in ((U.encode(img, quality=85, progressive=True), b"not baseline"),):
        assert lib.sm_jpeg_scan_prepare(C.cast(C.c_char_p(bad), C.c_void_p), len(bad), None, C.byref(sc)) < 0
        assert why in lib.sm_last_error(), lib.sm_last_error()

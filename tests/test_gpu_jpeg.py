"""f2 codec half on the GPU: the HIP reconstruction (inverse DCT, fancy upsampling, colour conversion: csrc/jpeg.hip) against PIL /
libjpeg-turbo and against the oracle BYTE FOR BYTE, batches, and a Motion-JPEG AVI read straight into HBM."""
import ctypes as C
import io

import numpy as np
import pytest
import torch

from oracle import jpeg_oracle as J
from tests import util_jpeg as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    from streammind_amd import native
    assert torch.cuda.is_available()
    return native.JpegDecoder(threads=4)


@pytest.mark.parametrize("case", range(len(U.CASES)))
def test_gpu_decode_equals_pil_and_oracle(dec, case):
    w, h, gray, kw = U.CASES[case]
    b = U.encode(U.test_image(w, h, case, gray), **kw)
    got = dec.decode([b])[0].cpu().numpy()
    want = U.pil_decode(b)
    assert got.shape == want.shape and np.array_equal(got, want), f"{int((got != want).sum())} bytes differ"
    info, coefs, qt = U.host_coefs(dec.lib, b)
    assert np.array_equal(got, J.reconstruct(coefs, qt, info))


def test_gpu_decode_batch_and_implied_tables(dec):
    """28 frames of one geometry in one call (host threads + one reconstruct launch pair); half of them without DHT segments"""
    frames = [U.encode(U.test_image(336, 336, 100 + i), quality=60 + i, subsampling=2, optimize=False) for i in range(28)]
    sent = [U.strip_dht(f) if i % 2 else f for i, f in enumerate(frames)]
    got = dec.decode(sent).cpu().numpy()
    for i, f in enumerate(frames):
        assert np.array_equal(got[i], U.pil_decode(f)), i
    with pytest.raises(Exception, match="batch was opened"):
        dec.decode([frames[0], U.encode(U.test_image(320, 240, 1), quality=75)])


def test_mjpeg_avi_straight_into_hbm(dec, tmp_path):
    """video_io.MjpegAviVideo.get_batch_gpu == get_batch (PIL) for a clip, and the frames feed the ingest + tower path as they are"""
    from tests.test_host_cpu import _write_mjpeg_avi
    from streammind_amd import video_io
    w, h, n = 400, 300, 9
    jpegs = [U.encode(U.test_image(w, h, 40 + i), quality=80, subsampling=2) for i in range(n)]
    _write_mjpeg_avi(tmp_path / "cam.avi", jpegs, w, h, 30, 1)
    vr = video_io.open_video(str(tmp_path / "cam.avi"))
    ids = [0, 3, 4, 8]
    gpu = vr.get_batch_gpu(ids, decoder=dec)
    assert gpu.is_cuda and gpu.dtype == torch.uint8 and tuple(gpu.shape) == (4, h, w, 3)
    assert np.array_equal(gpu.cpu().numpy(), vr.get_batch(ids).asnumpy())
    from streammind_amd import native
    sq = native.ingest_frames(gpu, pad_square=False)
    ref = native.ingest_frames(torch.from_numpy(vr.get_batch(ids).asnumpy()).cuda(), pad_square=False)
    assert torch.equal(sq, ref)


def test_stream_frames_gpu_equals_the_host_stream(dec, tmp_path):
    """video_io.stream_frames_gpu (MJPEG -> HBM -> tower-sized frames, batches) == stream_frames (PIL on the host) + the same ingest"""
    from tests.test_host_cpu import _write_mjpeg_avi
    from streammind_amd import video_io, native
    w, h, n = 352, 288, 31
    jpegs = [U.encode(U.test_image(w, h, 70 + i), quality=75, subsampling=2) for i in range(n)]
    _write_mjpeg_avi(tmp_path / "cam.avi", jpegs, w, h, 30, 1)
    host = list(video_io.stream_frames(str(tmp_path / "cam.avi"), 10))                    # every 3rd frame
    got_ids, got = [], []
    for ids, fr in video_io.stream_frames_gpu(str(tmp_path / "cam.avi"), 10, batch=4, decoder=dec):
        assert fr.is_cuda and fr.dtype == torch.uint8 and tuple(fr.shape[1:]) == (336, 336, 3)
        got_ids += ids
        got.append(fr)
    assert got_ids == [i for i, _ in host]
    want = native.ingest_frames(torch.from_numpy(np.stack([f for _, f in host])).cuda(), False, 336)
    assert torch.equal(torch.cat(got), want)



def test_gpu_decode_equals_golden_g19(dec):
    """the committed fixture through the HIP kernels: byte for byte the frames recorded from PIL / libjpeg-turbo"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g19_jpeg.npz"))
    for i in range(int(g["n"])):
        assert np.array_equal(dec.decode([bytes(g[f"jpeg{i}"])])[0].cpu().numpy(), g[f"rgb{i}"]), i
    assert np.array_equal(dec.decode([bytes(g["jpeg_bare"])])[0].cpu().numpy(), g["rgb0"])

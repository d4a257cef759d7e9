"""Video ingest on the host side: frame sources with the decord `VideoReader` surface the reference's
callers use, and the streaming sampler `read_video_stream` (eval/video_score_stream_demo.py:212-225).

The reference decodes on the CPU with decord (one thread, video_score_stream_demo.py:218; mm_utils.py:419), imageio (.gif,
mm_utils.py:400-407) or moviepy (.webm, :409-418).  None of them -- nor ffmpeg, OpenCV, PyAV, rocDecode -- exists in this
image, so the codec is an ADAPTOR: `open_video` serves what can be decoded with what is here (frame arrays, .npy / .npz,
directories of still images, multi-frame .gif / .tiff / .webp through PIL, Motion-JPEG .avi through a RIFF walk + PIL) and hands every other container to whichever of
decord / cv2 / imageio the deployment has installed.  Everything after the decoder (sampling, expand2square, PIL-exact
resize, centre crop, normalisation) is this build's own and runs on the GPU (`mm_utils.process_video`, `sm_ingest_frames`).

A source offers: len(src), src.get_avg_fps(), src[i].asnumpy() -> HWC uint8, src.get_batch(ids).asnumpy() -> [n,H,W,3]."""
from __future__ import annotations

import os
from typing import Iterator, List, Sequence, Tuple

import numpy as np

IMAGE_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".webp", ".ppm")


class _Frames:
    """what decord returns from __getitem__ / get_batch: an array-like with .asnumpy() (and .numpy(), mm_utils.py:432-435)"""

    def __init__(self, arr: np.ndarray):
        self._a = arr

    def asnumpy(self) -> np.ndarray:
        return self._a

    numpy = asnumpy

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    @property
    def shape(self):
        return self._a.shape


class ArrayVideo:
    """frames already in memory: [n, H, W, 3] uint8 (also the .npy / .npz source)"""

    def __init__(self, frames: np.ndarray, fps: float = 30.0):
        frames = np.asarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[-1] != 3:
            raise ValueError(f"ArrayVideo: expected [n, H, W, 3] uint8 frames, got {frames.dtype} {frames.shape}")
        self.frames, self.fps = frames, float(fps)

    def __len__(self) -> int:
        return len(self.frames)

    def get_avg_fps(self) -> float:
        return self.fps

    def __getitem__(self, i) -> _Frames:
        return _Frames(self.frames[int(i)])

    def get_batch(self, ids: Sequence[int]) -> _Frames:
        return _Frames(self.frames[np.asarray(ids, dtype=np.int64)])


class ImageSequenceVideo:
    """a directory of still images in name order (decoded lazily with PIL); fps from `fps.txt` in the directory, else 30"""

    def __init__(self, directory: str, fps: float | None = None):
        self.files = sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.lower().endswith(IMAGE_EXT))
        if not self.files:
            raise FileNotFoundError(f"no still images under {directory}")
        fp = os.path.join(directory, "fps.txt")
        self.fps = float(fps if fps is not None else (open(fp).read().strip() if os.path.exists(fp) else 30.0))

    def __len__(self) -> int:
        return len(self.files)

    def get_avg_fps(self) -> float:
        return self.fps

    def _load(self, i: int) -> np.ndarray:
        from PIL import Image
        with Image.open(self.files[int(i)]) as im:
            return np.asarray(im.convert("RGB"))

    def __getitem__(self, i) -> _Frames:
        return _Frames(self._load(i))

    def get_batch(self, ids: Sequence[int]) -> _Frames:
        return _Frames(np.stack([self._load(i) for i in ids]))


class PilMultiFrameVideo(ArrayVideo):
    """animated .gif / multi-page .tiff / animated .webp: all frames through PIL.  fps: 10 for .gif, the reference's constant
    (mm_utils.py:402), else from the container's frame duration when it states one"""

    def __init__(self, path: str):
        from PIL import Image, ImageSequence
        with Image.open(path) as im:
            frames = np.stack([np.asarray(f.convert("RGB")) for f in ImageSequence.Iterator(im)])
            dur = im.info.get("duration")
        fps = 10.0 if path.lower().endswith(".gif") else (1000.0 / dur if dur else 30.0)
        super().__init__(frames, fps)


class MjpegAviVideo:
    """Motion-JPEG in an AVI (RIFF) container -- what capture cards and many cameras write -- read without a codec library: the
    RIFF chunk walk is done here and every frame is a baseline JPEG that PIL decodes (lazily, random access through the chunk
    table).  Frames stored without Huffman tables (the `AVI1` convention: tables implied) get the standard JPEG tables spliced
    in before decoding.  fps = dwRate / dwScale of the video stream header, else 1e6 / dwMicroSecPerFrame.  Anything that is
    not MJPG raises, so `open_video` falls through to the installed decoders."""

    _std_dht = None

    def __init__(self, path: str):
        import mmap
        import struct
        self._file = open(path, "rb")              # mapped, not read: camera recordings run to gigabytes and frames are decoded lazily
        data = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ) if os.path.getsize(path) else b""
        if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"AVI ":
            raise ValueError(f"{path}: not a RIFF AVI file")
        self._data = data
        self._frames: List[Tuple[int, int]] = []          # (offset, size) of every video chunk, in stream order
        self.fps = 0.0
        handler = compression = None
        usec = 0

        def walk(lo: int, hi: int):
            nonlocal handler, compression, usec
            pos = lo
            while pos + 8 <= hi:
                cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
                body = pos + 8
                if cid == b"RIFF":                             # OpenDML: `RIFF....AVIX` continuation segments follow the first RIFF
                    if data[body:body + 4] in (b"AVI ", b"AVIX"):
                        walk(body + 4, min(body + size, hi))
                elif cid == b"LIST":
                    if data[body:body + 4] in (b"hdrl", b"strl", b"movi", b"rec "):
                        walk(body + 4, min(body + size, hi))
                elif cid == b"avih" and size >= 4:
                    usec = struct.unpack_from("<I", data, body)[0]
                elif cid == b"strh" and size >= 28 and data[body:body + 4] == b"vids" and handler is None:
                    handler = data[body + 4:body + 8]
                    scale, rate = struct.unpack_from("<II", data, body + 20)
                    if scale and rate:
                        self.fps = rate / scale
                elif cid == b"strf" and size >= 20 and compression is None and handler is not None:
                    compression = data[body + 16:body + 20]
                elif cid[2:4] in (b"dc", b"db") and cid[:2].isdigit():
                    if size > 0:
                        self._frames.append((body, size))
                    elif self._frames:                          # zero-length chunk = dropped frame: the player repeats the previous one,
                        self._frames.append(self._frames[-1])   # so frame index keeps tracking time
                pos = body + size + (size & 1)                # chunks are word aligned

        walk(0, len(data))                                      # every top-level RIFF segment (`AVI ` then any number of `AVIX`)
        tags = {t.upper() for t in (handler, compression) if t}
        if not tags & {b"MJPG", b"JPEG", b"AVRN", b"LJPG"} or not self._frames:
            raise ValueError(f"{path}: AVI video stream is {handler!r}/{compression!r}, not Motion-JPEG")
        if not self.fps:
            self.fps = 1e6 / usec if usec else 30.0

    @classmethod
    def _standard_tables(cls) -> bytes:
        """the DHT segments of the standard (ITU T.81 Annex K) Huffman tables, taken from a JPEG libjpeg writes with its defaults"""
        if cls._std_dht is None:
            import io
            from PIL import Image
            buf = io.BytesIO()
            Image.new("RGB", (16, 16), (120, 60, 200)).save(buf, "JPEG", quality=75, optimize=False)
            b, pos, out = buf.getvalue(), 2, b""
            while pos + 4 <= len(b) and b[pos] == 0xFF and b[pos + 1] != 0xDA:
                seg = 2 + int.from_bytes(b[pos + 2:pos + 4], "big")
                if b[pos + 1] == 0xC4:
                    out += b[pos:pos + seg]
                pos += seg
            cls._std_dht = out
        return cls._std_dht

    def _jpeg(self, i: int) -> bytes:
        off, size = self._frames[int(i)]
        j = bytes(self._data[off:off + size])
        sos = j.find(b"\xff\xda")
        if sos > 0 and j.find(b"\xff\xc4", 0, sos) < 0:         # no DHT before the scan: tables implied
            j = j[:sos] + self._standard_tables() + j[sos:]
        return j

    def __len__(self) -> int:
        return len(self._frames)

    def get_avg_fps(self) -> float:
        return float(self.fps)

    def _load(self, i: int) -> np.ndarray:
        import io
        from PIL import Image
        with Image.open(io.BytesIO(self._jpeg(i))) as im:
            return np.asarray(im.convert("RGB"))

    def __getitem__(self, i) -> _Frames:
        return _Frames(self._load(i))

    def get_batch(self, ids: Sequence[int]) -> _Frames:
        return _Frames(np.stack([self._load(i) for i in ids]))

    def get_batch_gpu(self, ids: Sequence[int], decoder=None):
        """the same frames as a u8 tensor [n, H, W, 3] ALREADY IN HBM, decoded by the C ABI's JPEG path (host Huffman threads + GPU
        inverse DCT / upsampling / colour: byte for byte `get_batch`) -- nothing is decoded by PIL, no RGB frame crosses PCIe.  Frames
        the native decoder does not implement (progressive, CMYK ...) take the PIL path and an upload."""
        import torch
        from . import native
        from ._lib import StreamMindHipError
        if decoder is None:
            decoder = self._gpu_decoder = getattr(self, "_gpu_decoder", None) or native.JpegDecoder()
        chunks = [bytes(self._data[o:o + n]) for o, n in (self._frames[int(i)] for i in ids)]       # raw chunks: implied tables are the decoder's business
        try:
            return decoder.decode(chunks)
        except StreamMindHipError:
            return torch.from_numpy(self.get_batch(ids).asnumpy()).to(decoder.device)


class _DecordVideo:
    def __init__(self, path: str):
        from decord import VideoReader, cpu
        self.vr = VideoReader(uri=path, ctx=cpu(0), num_threads=1)       # mm_utils.py:419 (one thread: no fork deadlock)

    def __len__(self):
        return len(self.vr)

    def get_avg_fps(self):
        return float(self.vr.get_avg_fps())

    def __getitem__(self, i):
        return self.vr[int(i)]

    def get_batch(self, ids):
        return self.vr.get_batch(list(ids))


class _Cv2Video(ArrayVideo):
    def __init__(self, path: str):
        import cv2
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            raise IOError(f"cv2 cannot open {path}")
        fps, frames = cap.get(cv2.CAP_PROP_FPS) or 30.0, []
        while True:
            ok, bgr = cap.read()
            if not ok:
                break
            frames.append(bgr[:, :, ::-1])
        cap.release()
        super().__init__(np.ascontiguousarray(np.stack(frames)), fps)


class _ImageioVideo(ArrayVideo):
    def __init__(self, path: str):
        import imageio
        rd = imageio.get_reader(path)
        fps = float(rd.get_meta_data().get("fps", 30.0))
        super().__init__(np.stack([np.asarray(f)[..., :3] for f in rd]), fps)


def open_video(src, fps: float | None = None):
    """path / directory / array -> a frame source (see the module docstring)."""
    if isinstance(src, np.ndarray):
        return ArrayVideo(src, 30.0 if fps is None else fps)
    if hasattr(src, "get_avg_fps") and hasattr(src, "__len__"):
        return src                                          # already a reader (decord.VideoReader or one of the above)
    if not isinstance(src, str):
        raise TypeError(f"open_video: unsupported source {type(src)}")
    if os.path.isdir(src):
        return ImageSequenceVideo(src, fps)
    if not os.path.exists(src):
        raise FileNotFoundError(src)
    low = src.lower()
    if low.endswith(".npy"):
        return ArrayVideo(np.load(src), 30.0 if fps is None else fps)
    if low.endswith(".npz"):
        z = np.load(src)
        return ArrayVideo(z["frames"], float(z["fps"]) if "fps" in z.files and fps is None else (fps or 30.0))
    if low.endswith((".gif", ".tif", ".tiff", ".webp")):
        return PilMultiFrameVideo(src)
    errors: List[str] = []
    if low.endswith(".avi"):
        try:
            return MjpegAviVideo(src)                           # Motion-JPEG: decodable with what this image has (RIFF walk + PIL)
        except ValueError as e:
            errors.append(f"built-in MJPEG reader: {e}")
    for name, cls in (("decord", _DecordVideo), ("cv2", _Cv2Video), ("imageio", _ImageioVideo)):
        try:
            return cls(src)
        except ImportError:
            errors.append(f"{name}: not installed")
        except Exception as e:                                    # installed but cannot read this file: try the next decoder
            errors.append(f"{name}: {e!r}")
    raise ImportError(f"no decoder for {src}: " + "; ".join(errors) + ". This image ships no video codec (no decord / OpenCV / imageio / "
                      "ffmpeg / rocDecode): install one of decord, opencv-python, imageio[ffmpeg], or pass decoded frames / a .npy file / "
                      "a directory of stills.")


def get_index_stream(start_frame: int, end_frame: int, vidoe_fps: float, cur_fps: float = 2) -> np.ndarray:
    """eval/video_score_stream_demo.py:212-215: every int(video_fps / cur_fps)-th frame of [start, end)"""
    seg_size = int(vidoe_fps / cur_fps)
    return np.arange(start_frame, end_frame, seg_size, dtype=int)


def read_video_stream(video_path, cur_fps: float) -> Tuple[np.ndarray, object]:
    """eval/video_score_stream_demo.py:217-225: (frame indices at `cur_fps`, the open reader); the LAST frame is never sampled
    (end = len - 1, exclusive)."""
    vr = open_video(video_path)
    max_frame = len(vr) - 1
    video_fps = float(vr.get_avg_fps())
    return get_index_stream(start_frame=0, end_frame=max_frame, vidoe_fps=video_fps, cur_fps=cur_fps), vr


def stream_frames(video_path, cur_fps: float) -> Iterator[Tuple[int, np.ndarray]]:
    """(frame id, HWC uint8 frame) in stream order: what the demo loop feeds one by one (video_score_stream_demo.py:283-287),
    and what `StreamingSession.run` takes as its frame iterator (the pinned ring buffer sits behind it)."""
    ids, vr = read_video_stream(video_path, cur_fps)
    for fid in ids:
        yield int(fid), np.asarray(vr[int(fid)].asnumpy())


def stream_frames_gpu(video_path, cur_fps: float, batch: int = 28, image_size: int = 336, pad_square: bool = False, decoder=None):
    """The same stream as `stream_frames`, `batch` sampled frames at a time, ALREADY IN HBM and already at the tower's input size:
    (frame ids, u8 cuda tensor [n, image_size, image_size, 3]).  For a Motion-JPEG source nothing is decoded by PIL and no RGB frame
    crosses PCIe: host Huffman threads -> coefficient upload -> GPU inverse DCT / upsampling / colour (`get_batch_gpu`, byte for byte
    PIL's frames) -> PIL-exact resize / crop (`sm_ingest_frames`); any other source is decoded by its reader and uploaded.  The tensors
    feed `NativeStream.push_frames` / `StreamingSession` directly."""
    import torch
    from . import native
    ids, vr = read_video_stream(video_path, cur_fps)
    for k in range(0, len(ids), batch):
        chunk = [int(i) for i in ids[k:k + batch]]
        if hasattr(vr, "get_batch_gpu"):
            fr = vr.get_batch_gpu(chunk, decoder=decoder)
        else:
            fr = torch.from_numpy(np.ascontiguousarray(vr.get_batch(chunk).asnumpy())).cuda()
        if tuple(fr.shape[1:3]) != (image_size, image_size):
            fr = native.ingest_frames(fr.contiguous(), pad_square, image_size)
        yield chunk, fr


"""Frame/text plumbing of the streaming path (mirror of streammind/mm_utils.py: process_video list branch :377,449-467,
tokenizer_MMODAL_token :567-604, KeywordsStoppingCriteria :616-647).  Integer / string logic only."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .constants import IMAGE_TOKEN_INDEX, MAX_FRAMES, MMODAL_INDEX_TOKEN, NUM_FRAMES, NUM_FRAMES_PER_SECOND


def frame_sample(duration: int, mode: str = "uniform", num_frames: int = NUM_FRAMES, local_fps=None) -> List[int]:
    """mm_utils.py:378-397 (the nested helper of process_video): which decoded frames feed the model."""
    if mode == "uniform":
        seg_size = float(duration - 1) / num_frames
        return [(int(np.round(seg_size * i)) + int(np.round(seg_size * (i + 1)))) // 2 for i in range(num_frames)]
    if mode == "fps":
        assert local_fps is not None
        segment_len = min(local_fps // NUM_FRAMES_PER_SECOND, duration)
        return np.arange(segment_len // 2, duration, segment_len, dtype=int).tolist()
    raise ImportError(f"Unsupported frame sampling mode: {mode}")


def _crop_side(processor) -> int:
    """side of the processor's centre crop: `crop_size` is a dict in transformers 4.x, a SizeDict object (attribute AND item
    access, not a dict subclass) in 5.x, an int in very old versions; 336 (CLIP-L/14-336) when there is no processor"""
    size = getattr(processor, "crop_size", None)
    if size is None:
        return 336
    if isinstance(size, int):
        return size
    try:
        return int(size["height"])
    except (TypeError, KeyError):
        return int(getattr(size, "height"))


def process_video(video_path, processor=None, aspect_ratio="pad", num_frames: int = NUM_FRAMES, image_grid: bool = False,
                  sample_scheme: str = "uniform") -> torch.Tensor:
    """list of PIL images / HWC arrays (or an [n,H,W,3] uint8 array) -> uint8 frames tensor [n,S,S,3], S = the
    processor's crop size (336).

    The reference normalises on the CPU here (HF CLIPImageProcessor) and ships fp32/fp16 pixel_values; this build keeps
    the frames as u8 -- the (x/255 - mean)/std affine is fused into the device-side patchify kernel -- so the
    host->device copy is 4x smaller.  Frames that already are S x S pass through on the host (resize and crop are
    identities); any other size goes through the device ingest front-end (`sm_ingest_frames`:
    expand2square with int(image_mean * 255) when aspect_ratio == "pad", PIL-exact bicubic shortest-edge resize, centre
    crop) and comes back as a CUDA tensor.  A path goes through `video_io.open_video` (decoder adaptor) and the reference's
    frame sampling first.  Parameter names and defaults are the reference's (mm_utils.py:377; `video_path` is a path OR the frame
    list, as in its list branch :449-451; the streaming caller always passes aspect_ratio=None through its partial)."""
    video = video_path
    if isinstance(video, str):
        # mm_utils.py:399-435: open, sample `num_frames` ids ("uniform") or one per second ("fps"), cap at MAX_FRAMES, fetch.
        # The decoder is an adaptor (video_io.open_video): .gif at the reference's constant 10 fps, other containers through
        # whichever decoder the deployment has; this image ships none, see video_io.
        from .video_io import open_video
        vr = open_video(video)
        duration, local_fps = len(vr), float(vr.get_avg_fps())
        ids = frame_sample(duration, mode=sample_scheme, num_frames=num_frames, local_fps=local_fps)
        if len(ids) > MAX_FRAMES:
            ids = np.linspace(0, duration - 1, MAX_FRAMES, dtype=int).tolist()
        if video.lower().endswith(".gif"):
            ids = sorted(set(int(i) for i in ids))          # mm_utils.py:407 keeps each sampled index once, in file order
        video = np.asarray(vr.get_batch(ids).asnumpy())
        num_frames = len(ids)
    if image_grid:
        raise NotImplementedError("image_grid=True (photo grid prepended, mm_utils.py:447-450) is not on the streaming path")
    if isinstance(video, np.ndarray):
        frames = video
    else:
        frames = np.stack([np.asarray(x) for x in video])
    assert len(frames) == num_frames, (len(frames), num_frames)
    assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.shape[-1] == 3
    S = _crop_side(processor)
    if frames.shape[1] == S and frames.shape[2] == S:
        return torch.from_numpy(np.ascontiguousarray(frames))
    from . import native
    mean = getattr(processor, "image_mean", None) or (0.48145466, 0.4578275, 0.40821073)
    dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    return native.ingest_frames(dev, pad_square=(aspect_ratio == "pad"), image_size=S, pad_rgb=tuple(int(x * 255) for x in mean))


def tokenizer_MMODAL_token(prompt: str, tokenizer, MMODAL_token_index: int = IMAGE_TOKEN_INDEX, return_tensors=None):
    """split on '<video>', tokenise each chunk, re-join with the sentinel id; one BOS kept (mm_utils.py:567-604)."""
    tag = f"<{MMODAL_INDEX_TOKEN[MMODAL_token_index].lower()}>"
    prompt_chunks = [tokenizer(chunk).input_ids for chunk in prompt.split(tag)]

    def insert_separator(X, sep):
        return [ele for sublist in zip(X, [sep] * len(X)) for ele in sublist][:-1]

    input_ids: List[int] = []
    offset = 0
    if len(prompt_chunks) > 0 and len(prompt_chunks[0]) > 0 and prompt_chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        input_ids.append(prompt_chunks[0][0])
    for x in insert_separator(prompt_chunks, [MMODAL_token_index] * (offset + 1)):
        input_ids.extend(x[offset:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(input_ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return input_ids


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors=None):
    """mm_utils.py:545-565: the `<image>` form of the same split-and-rejoin."""
    return tokenizer_MMODAL_token(prompt, tokenizer, image_token_index, return_tensors)


def get_model_name_from_path(model_path: str) -> str:
    """mm_utils.py:607-613: last path component, `<parent>_<checkpoint-N>` for trainer checkpoints."""
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


def expand2square(pil_img, background_color):
    """mm_utils.py:257-268 on a PIL image or an HWC uint8 array: centre the picture on a square canvas of `background_color`
    (the streaming path does this on the GPU inside sm_ingest_frames; this host form serves callers that hold single images)."""
    img = pil_img
    arr = np.asarray(img)
    h, w = arr.shape[:2]
    if h == w:
        return img
    side = max(h, w)
    out = np.empty((side, side, arr.shape[2]), dtype=arr.dtype)
    out[...] = np.asarray(background_color, dtype=arr.dtype)
    y0, x0 = ((side - h) // 2, 0) if w > h else (0, (side - w) // 2)
    out[y0:y0 + h, x0:x0 + w] = arr
    if isinstance(img, np.ndarray):
        return out
    from PIL import Image
    return Image.fromarray(out, mode=img.mode)


def process_image(image_path, processor=None, aspect_ratio="pad", num_frames: int = NUM_FRAMES, image_grid: bool = False) -> torch.Tensor:
    """mm_utils.py:356-374 for an already decoded picture (PIL image or HWC uint8 array; opening a path is left to the caller):
    the image repeated `num_frames` times through the same front-end as process_video."""
    image = image_path
    if isinstance(image, str):
        raise NotImplementedError("image decoding is outside this build: pass a PIL image or an HWC uint8 array")
    frame = np.asarray(image.convert("RGB") if hasattr(image, "convert") else image)
    return process_video([frame] * num_frames, processor, aspect_ratio=aspect_ratio, num_frames=num_frames, image_grid=image_grid)


class KeywordsStoppingCriteria:
    """stop when the tail ids equal a keyword's ids or the decoded tail contains a keyword (mm_utils.py:616-647).
    Callable as criterion(output_ids[LongTensor 1 x n], scores) like an HF StoppingCriteria."""

    def __init__(self, keywords: Sequence[str], tokenizer, input_ids: torch.Tensor):
        self.keywords = list(keywords)
        self.keyword_ids = []
        self.max_keyword_len = 0
        for keyword in keywords:
            cur = tokenizer(keyword).input_ids
            if len(cur) > 1 and cur[0] == tokenizer.bos_token_id:
                cur = cur[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(cur))
            self.keyword_ids.append(torch.tensor(cur))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids: torch.Tensor, scores=None, **kwargs) -> bool:
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        for keyword_id in self.keyword_ids:
            keyword_id = keyword_id.to(output_ids.device)
            if output_ids.shape[1] >= keyword_id.shape[0] and (output_ids[0, -keyword_id.shape[0]:] == keyword_id).all():
                return True
        outputs = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(keyword in outputs for keyword in self.keywords)

    def __call__(self, output_ids: torch.Tensor, scores, **kwargs) -> bool:
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))

"""Frame/text plumbing of the streaming path (mirror of streammind/mm_utils.py: process_video list branch :377,449-467,
tokenizer_MMODAL_token :567-604, KeywordsStoppingCriteria :616-647).  Integer / string logic only."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .constants import IMAGE_TOKEN_INDEX, MMODAL_INDEX_TOKEN, NUM_FRAMES


def process_video(video, processor=None, aspect_ratio=None, num_frames: int = NUM_FRAMES, image_grid: bool = False,
                  sample_scheme: str = "uniform") -> torch.Tensor:
    """list of PIL images / HWC arrays (or an [n,H,W,3] uint8 array) -> uint8 frames tensor [n,H,W,3].

    The reference normalises on the CPU here (HF CLIPImageProcessor) and ships fp32/fp16 pixel_values; this build keeps
    the frames as u8 -- the (x/255 - mean)/std affine is fused into the device-side patchify kernel -- so the
    host->device copy is 4x smaller.  `processor` is accepted for signature compatibility and only consulted for the
    expected size.  Sources that are not already image_size x image_size need the resize/crop front-end
    (SURVEY 8f f2), which is out of scope: they are rejected."""
    if isinstance(video, str):
        raise NotImplementedError("video decoding (decord) is the ingest front-end, out of scope (SURVEY 8f f2)")
    if isinstance(video, np.ndarray):
        frames = video
    else:
        frames = np.stack([np.asarray(x) for x in video])
    assert len(frames) == num_frames, (len(frames), num_frames)
    if aspect_ratio == "pad" and frames.shape[1] != frames.shape[2]:
        raise NotImplementedError("expand2square + resize is the ingest front-end, out of scope (SURVEY 8f f2)")
    size = getattr(processor, "crop_size", None)
    if isinstance(size, dict) and (frames.shape[1] != size["height"] or frames.shape[2] != size["width"]):
        raise NotImplementedError(f"frames must already be {size['height']}x{size['width']} (resize is out of scope)")
    assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.shape[-1] == 3
    return torch.from_numpy(np.ascontiguousarray(frames))


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

    def __call__(self, output_ids: torch.Tensor, scores=None, **kwargs) -> bool:
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))

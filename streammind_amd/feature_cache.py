"""BASELINE config-1 plumbing: the on-disk CLIP feature cache.

    encode_video_features  <- encode_all_videos_score, videollama2_arch.py:211-282 : frames -> chunks of <= 500 frames through
                              the vision tower -> `{name}_encode_feature_frame_{s}_{e}.pt` holding [1, n, 576, 1024]
    process_file / subsample_features <- process_clip_encoder.py:55-57,69-84 : torch.load(p)[:, ::segment] with
                              segment = fps // target = 12, written under `features_video_encode_ddp_fps`

Frames must already be image_size x image_size u8 (the resize / expand2square front-end is row f2)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch

CHUNK = 500                    # videollama2_arch.py:252-257
FPS, TARGET = 25, 2            # process_clip_encoder.py:55-56
SEGMENT = FPS // TARGET


def chunk_name(video_name: str, start: int) -> str:
    """videollama2_arch.py:277-281: the name always says start + 500, also for the short last chunk (golden g13)"""
    return f"{video_name}_encode_feature_frame_{start}_{start + CHUNK}.pt"


def rank_slice(videos: Sequence[str], rank: int, world: int) -> List[str]:
    """videollama2_arch.py:236-242: contiguous blocks of len // world videos per rank (the remainder is not encoded by anyone,
    as in the reference)."""
    per = len(videos) // world
    return list(videos[rank * per:(rank + 1) * per])


def output_dir(video_path: str) -> Tuple[str, str]:
    """videollama2_arch.py:243-244,277-279: (directory of the chunk files, `half` prefix) for a source video path"""
    half = os.path.basename(video_path).split("_224p.mkv")[0]
    return os.path.dirname(video_path.replace("features_video", "features_video_encode_ddp")), half


def encode_video_features(tower, frames_u8: torch.Tensor, out_dir: str, video_name: str, dtype=torch.bfloat16) -> List[str]:
    """tower: streammind_amd NativeModel.  frames_u8 [N,H,W,3] (host or device): all frames of ONE video -> chunk files of
    <= 500 frames each, [1, n, P, C] (videollama2_arch.py:252-281).  Which videos a rank encodes: `rank_slice`."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    n = frames_u8.shape[0]
    chunks = [(s, min(s + CHUNK, n)) for s in range(0, n, CHUNK)]
    B = tower.cfg.max_frames_per_call
    for s, e in chunks:
        feats = []
        for i in range(s, e, B):
            fr = frames_u8[i:min(i + B, e)].to(tower.device).contiguous()
            _, f = tower.vit_encode(fr, return_feats=True)
            feats.append(f)
        out = torch.cat(feats).unsqueeze(0).to(dtype).cpu()               # [1, n, P, C]
        p = os.path.join(out_dir, chunk_name(video_name, s))
        torch.save(out, p)
        paths.append(p)
    return paths


def subsample_features(feats: torch.Tensor, segment: int = SEGMENT) -> torch.Tensor:
    return feats[:, ::segment]                                            # process_clip_encoder.py:75


def process_file(video_encoder_path: str, progress_bar=None, lock=None) -> str:
    """stride one cached chunk; output goes to the sibling `..._fps` tree (process_clip_encoder.py:69-84).  progress_bar / lock:
    the reference's tqdm bar and threading.Lock (updated under the lock, :79-80), optional here.  Returns the output path (the
    reference returns a status sentence built around it, and swallows errors into that sentence; errors raise here)."""
    path = video_encoder_path
    out = path.replace("features_video_encode_ddp", "features_video_encode_ddp_fps")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    torch.save(subsample_features(torch.load(path)), out)
    if progress_bar is not None:
        if lock is not None:
            with lock:
                progress_bar.update(1)
        else:
            progress_bar.update(1)
    return out

"""BASELINE config-1 plumbing: the on-disk CLIP feature cache.

    encode_video_features  <- encode_all_videos_score, videollama2_arch.py:211-282 : frames -> chunks of <= 500 frames through
                              the vision tower -> `{name}_encode_feature_frame_{s}_{e}.pt` holding [1, n, 576, 1024]
    process_file / subsample_features <- process_clip_encoder.py:55-57,69-84 : torch.load(p)[:, ::segment] with
                              segment = fps // target = 12, written under `features_video_encode_ddp_fps`

Frames must already be image_size x image_size u8 (the resize / expand2square front-end is row f2)."""
from __future__ import annotations

import os
from typing import List

import torch

CHUNK = 500                    # videollama2_arch.py:252-257
FPS, TARGET = 25, 2            # process_clip_encoder.py:55-56
SEGMENT = FPS // TARGET


def chunk_name(video_name: str, start: int, end: int) -> str:
    return f"{video_name}_encode_feature_frame_{start}_{end}.pt"          # videollama2_arch.py:277-281


def encode_video_features(tower, frames_u8: torch.Tensor, out_dir: str, video_name: str, dtype=torch.bfloat16,
                          rank: int = 0, world: int = 1) -> List[str]:
    """tower: streammind_amd NativeModel.  frames_u8 [N,H,W,3] (host or device).  Chunks are rank-sliced like the
    reference's video list (videollama2_arch.py:239-242)."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    n = frames_u8.shape[0]
    chunks = [(s, min(s + CHUNK, n)) for s in range(0, n, CHUNK)]
    B = tower.cfg.max_frames_per_call
    for ci, (s, e) in enumerate(chunks):
        if ci % world != rank:
            continue
        feats = []
        for i in range(s, e, B):
            fr = frames_u8[i:min(i + B, e)].to(tower.device).contiguous()
            _, f = tower.vit_encode(fr, return_feats=True)
            feats.append(f)
        out = torch.cat(feats).unsqueeze(0).to(dtype).cpu()               # [1, n, P, C]
        p = os.path.join(out_dir, chunk_name(video_name, s, e))
        torch.save(out, p)
        paths.append(p)
    return paths


def subsample_features(feats: torch.Tensor, segment: int = SEGMENT) -> torch.Tensor:
    return feats[:, ::segment]                                            # process_clip_encoder.py:75


def process_file(path: str) -> str:
    """stride one cached chunk; output goes to the sibling `..._fps` tree (process_clip_encoder.py:71-76)."""
    out = path.replace("features_video_encode_ddp", "features_video_encode_ddp_fps")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    torch.save(subsample_features(torch.load(path)), out)
    return out

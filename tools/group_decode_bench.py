"""Batched decode across streams (sm_group_llm_decode) at Mistral-7B shapes: aggregate tokens/s per group size.
    python tools/group_decode_bench.py [sizes, e.g. 1,8,32]      (SM_DECODE_ATTN_FUSED=0 / SM_NO_FUSED_ROPE=1: A/B switches; FP8=1: weight-only fp8 weights, up to 32 streams)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig

sizes = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 4, 8, 16, 32)
cfg = PathConfig(llm_layers=32, max_frames_per_call=1, vit_layers=2, weights_fp8=int(os.environ.get("FP8", "0")))
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
bench.random_llm_weights_into(model, cfg, 2)
model.finalize()
print(bench.group_decode_leg(model, cfg, sizes=sizes))

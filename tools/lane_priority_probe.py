"""LLM lane on a high-priority HIP stream?  bench.live_overlap_leg with SM_LLM_LANE_PRIORITY = 0 / -1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
cfg = PathConfig(llm_layers=32, max_frames_per_call=56)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
bench.random_llm_weights_into(model, cfg, 2)
model.finalize()
frames = bench.synthetic_frames_gpu(224, 336, 1234, 0)
stream = model.open_stream(max_frames=65536, max_seq=2048)
for pr in ("0", "-1", "0", "-1"):
    os.environ["SM_LLM_LANE_PRIORITY"] = pr
    r = bench.live_overlap_leg(model, stream, cfg, frames, 56)
    print("priority", pr, {k: r[k] for k in ("serial_seconds", "overlapped_seconds", "speedup")}, r["alone"], r["sharing_the_chip"]["decode_tokens_per_s"], r["sharing_the_chip"]["frames_per_s"])

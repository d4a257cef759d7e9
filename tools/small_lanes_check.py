"""Frame lanes of a small call: pooled features of an F-frame call must equal F one-frame calls bit for bit (full-size tower)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
cfg = PathConfig(llm_layers=0, max_frames_per_call=8)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
model.finalize()
frames = torch.randint(0, 256, (8, 336, 336, 3), dtype=torch.uint8, device="cuda")
one = torch.cat([model.vit_encode(frames[i:i + 1]) for i in range(8)])
for F in (2, 3, 4, 5, 8):
    for rep in range(3):
        got = model.vit_encode(frames[:F].contiguous())
        print(F, rep, "equal" if torch.equal(got, one[:F]) else f"DIFF {(got - one[:F]).abs().max().item():.3e}")

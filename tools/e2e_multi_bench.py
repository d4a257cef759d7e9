"""bench.e2e_multi_leg on its own: S concurrent streams, one frame per stream per tick, scheduled fires in staggered cohorts, the cohort's replies decoded
together (lock step), then the same with the replies in flight across ticks (continuous, 4 / 8 / 16 / 32 decode steps per tick).
    python tools/e2e_multi_bench.py [S=32] [ticks=112] [cohorts=4]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 112
cohorts = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.set_grad_enabled(False)
cfg = PathConfig(llm_layers=32, max_frames_per_call=max(S, 28))
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
bench.random_llm_weights_into(model, cfg, 2)
model.finalize()
frames = bench.synthetic_frames_gpu(4 * S, 336, 1, 0)
print(bench.e2e_multi_leg(model, cfg, frames, S=S, ticks=ticks, cohorts=cohorts))
for chunk in (4, 8, 16, 32):
    print(bench.e2e_multi_leg(model, cfg, frames, S=S, ticks=ticks, cohorts=cohorts, continuous=True, chunk=chunk))

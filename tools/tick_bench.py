"""One-frame (or F-frame) tick micro-benchmark: full-size perception model, N push_frames calls of F frames each
(for rocprofv3 kernel traces of the reference's own operating point, n_new = 1: SURVEY a3).
    python tools/tick_bench.py [F=1] [N=200]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = PathConfig(llm_layers=0, max_frames_per_call=max(F, 8))
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
model.finalize()
s = model.open_stream(max_frames=N * F + 64, max_seq=64)
frames = torch.randint(0, 256, (F, 336, 336, 3), dtype=torch.uint8, device="cuda")
for _ in range(10):
    s.push_frames(frames)
s.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    s.push_frames(frames)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print({"frames_per_call": F, "calls": N, "ms_per_call": round(dt * 1e3, 4), "frames_per_s": round(F / dt, 1)})

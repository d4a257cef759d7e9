"""Is the one-frame tick host-bound?  Host issue time of push_frames (no sync) next to the GPU time between two events."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = PathConfig(llm_layers=0, max_frames_per_call=8)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
model.finalize()
s = model.open_stream(max_frames=8192, max_seq=64)
frames = torch.randint(0, 256, (F, 336, 336, 3), dtype=torch.uint8, device="cuda")
for _ in range(20):
    s.push_frames(frames)
torch.cuda.synchronize()
# (a) one tick at a time: host issue time, then GPU completion
iss, tot = [], []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); s.push_frames(frames); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    iss.append(t1 - t0); tot.append(t2 - t0)
iss.sort(); tot.sort()
print(f"F={F}: one tick from an idle GPU: host issue {iss[25]*1e3:.3f} ms, until complete {tot[25]*1e3:.3f} ms")
# (b) 100 ticks back to back: host issue of all, then drain
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    s.push_frames(frames)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"F={F}: 100 ticks back to back: host issued in {(t1-t0)*10:.3f} ms/tick, complete {(t2-t0)*10:.3f} ms/tick (drain after the last issue {(t2-t1)*1e3:.3f} ms)")

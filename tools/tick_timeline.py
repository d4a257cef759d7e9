"""In-situ phase stamps of the LAST 128x128 GEMM launch of a one-frame tick (the ViT's last fc2 split-K pass: 160 blocks x 16 k-tiles),
with the -DSM_GEMM128_TIMELINE variant library (STREAMMIND_HIP_LIB)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
raw = ctypes.CDLL(os.environ["STREAMMIND_HIP_LIB"])
cfg = PathConfig(llm_layers=0, max_frames_per_call=8)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
model.finalize()
s = model.open_stream(max_frames=4096, max_seq=64)
frames = torch.randint(0, 256, (1, 336, 336, 3), dtype=torch.uint8, device="cuda")
tl = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
raw.sm_debug_set_timeline128.argtypes = [ctypes.c_void_p]
for _ in range(20):
    s.push_frames(frames)
torch.cuda.synchronize()
assert raw.sm_debug_set_timeline128(tl.data_ptr()) == 0
for _ in range(5):
    s.push_frames(frames)
torch.cuda.synchronize()
h = tl.cpu().view(-1, 8)[:160]
t0 = int(h[:, 0].min())
us = (h[:, :4] - t0).double() * 0.01
print(f"last GEMM of the tick (fc2 split-K pass, 160 blocks): span {float(us[:, 3].max()):.1f} us; start p50/p100 {float(us[:,0].median()):.1f}/{float(us[:,0].max()):.1f}; "
      f"first tile +{float((us[:,1]-us[:,0]).mean()):.2f}; loop {float((us[:,2]-us[:,1]).mean()):.2f} (max {float((us[:,2]-us[:,1]).max()):.2f}); epilogue {float((us[:,3]-us[:,2]).mean()):.2f}")

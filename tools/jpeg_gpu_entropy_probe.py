"""GPU entropy decode of restart intervals (sm_jpeg_entropy_decode): frames/s against batch size and interval density, 720p 4:2:0 q85.
    python tools/jpeg_gpu_entropy_probe.py [B=28] [rows=1 | -N = one interval per N MCUs]      (run under rocprofv3 --kernel-trace --stats for kernel times)"""
import io, os, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native
rng = np.random.default_rng(3)
H, W = 720, 1280
B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1
yy, xx = np.mgrid[0:H, 0:W]
jp = []
for i in range(min(B, 28)):
    img = np.stack([128 + 100 * np.sin(xx / (9.0 + i)) * np.cos(yy / 7.0), 128 + 110 * np.sin((xx + yy) / 13.0), 255.0 * ((xx // 11 + yy // 5 + i) % 2)], axis=2)
    img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO(); Image.fromarray(img).save(buf, "JPEG", quality=85, subsampling=2, **({"restart_marker_rows": rows} if rows > 0 else {"restart_marker_blocks": -rows})); jp.append(buf.getvalue())
jp = (jp * ((B + 27) // 28))[:B]
dec = native.JpegDecoder(threads=16)
out = dec.decode(jp, entropy="gpu")
ok = bool(np.array_equal(out[0].cpu().numpy(), np.asarray(Image.open(io.BytesIO(jp[0])).convert("RGB"))))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    dec.decode(jp, entropy="gpu")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"B={B} intervals={'%d MCU rows' % rows if rows > 0 else '%d MCUs' % -rows} ({len(jp[0]) / 1e6:.2f} MB per frame): {B / dt:8.1f} frames/s, {dt * 1e3:7.2f} ms per batch, byte-identical to PIL: {ok}")

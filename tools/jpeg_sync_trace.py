"""One batch shape of the self-synchronising JPEG decode, repeated (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import util_jpeg as U
from streammind_amd import native
q = int(os.environ.get("Q", "85")); n = int(os.environ.get("N", "28"))
dec = native.JpegDecoder(threads=16)
fr = [U.encode(U.test_image(1280, 720, 40 + i), quality=q, subsampling=2) for i in range(min(n, 28))] * max(1, n // 28)
dec.decode(fr, entropy="gpu")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    dec.decode(fr, entropy="gpu")
torch.cuda.synchronize()
print("ms per batch", (time.perf_counter() - t0) / 10 * 1e3, "kB per frame", sum(map(len, fr)) / len(fr) / 1e3)

"""where does the persistent 256x256 GEMM differ from the one-tile-per-block kernel?  (debug aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native
M, N, K = 16156, 3072, 1024
g = torch.Generator().manual_seed(1)
w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
x = torch.randn(M, K, generator=g).bfloat16().cuda()
wp = native.pack_weight(w)
ref = native.linear(x, wp, N, K, out_dtype=torch.bfloat16, tile_hint=2561)
for it in range(3):
    y = native.linear(x, wp, N, K, out_dtype=torch.bfloat16, tile_hint=256)
    bad = (y != ref)
    nb = int(bad.sum())
    print("iter", it, "mismatches", nb, "of", y.numel())
    if nb:
        idx = bad.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print(" rows: min", int(rows.min()), "max", int(rows.max()), " distinct row tiles", sorted(set((rows // 256).tolist()))[:20])
        print(" cols: distinct col tiles", sorted(set((cols // 256).tolist())), " col%256 distinct count", len(set((cols % 256).tolist())))
        t = (rows // 256) * 12 + cols // 256
        print(" distinct tiles", len(set(t.tolist())), " sample", idx[:8].tolist())
        r0, c0 = int(rows[0]), int(cols[0])
        print(" first bad: y", float(y[r0, c0]), "ref", float(ref[r0, c0]), "  row%256", r0 % 256, "col%256", c0 % 256)
        # is y a permutation within the tile?
        tm, tn = r0 // 256, c0 // 256
        yt, rt = y[tm*256:(tm+1)*256, tn*256:(tn+1)*256].float(), ref[tm*256:(tm+1)*256, tn*256:(tn+1)*256].float()
        print(" tile bad count", int((yt != rt).sum()), " bad rows in tile", sorted(set((yt != rt).nonzero()[:, 0].tolist()))[:16], " bad cols", sorted(set((yt != rt).nonzero()[:, 1].tolist()))[:40])

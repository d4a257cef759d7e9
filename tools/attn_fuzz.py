"""The three attention entry points on RANDOM shapes against fp32 softmax attention of the same bf16 operands: the ViT's non-causal kernel (batch, tokens, heads),
the causal prefill kernel (new queries behind a cache, GQA, sliding window, the paired query tiles), single-token decode (one-launch kernel and key-split + merge,
sliding window).  bf16 outputs: 8e-3 of the largest value.  One JSON line; exit code 1 on a mismatch.      python tools/attn_fuzz.py [N=300] [seed=1]"""
import os, sys, json, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from streammind_amd._lib import load, check, StreamMindHipError

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
lib = load()
st = lambda: torch.cuda.current_stream().cuda_stream
rb = lambda *s: torch.randn(*s, generator=g, device="cuda").bfloat16()
relerr = lambda a, b: float((a.float() - b.float()).abs().max()) / float(b.float().abs().max().clamp_min(1e-6))
bad, refused, kinds = [], {}, {"vit": 0, "prefill": 0, "decode": 0}
t0 = time.time()
for case in range(n_cases):
    kind = ["vit", "prefill", "decode"][int(rng.integers(0, 3))]
    try:
        if kind == "vit":
            B, S, H = int(rng.integers(1, 5)), int(rng.choice([1, 17, 50, 64, 65, 130, 257, 577, 600])), int(rng.choice([1, 2, 4, 16]))
            dh = int(rng.choice([64, 64, 128]))
            D = H * dh
            qkv = rb(B * S, 3 * D)
            ctx = torch.empty(B * S, D, device="cuda", dtype=torch.bfloat16)
            check(lib.sm_vit_attention(qkv.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0, 0, st()))
            q = qkv[:, :D].float().reshape(B, S, H, dh).transpose(1, 2); k = qkv[:, D:2 * D].float().reshape(B, S, H, dh).transpose(1, 2)
            v = qkv[:, 2 * D:].float().reshape(B, S, H, dh).transpose(1, 2)
            ref = (torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(B * S, D)
            desc = {"kind": kind, "B": B, "S": S, "H": H, "dh": dh}
        else:
            KV = int(rng.choice([1, 2, 4, 8])); rep = int(rng.choice([1, 2, 4, 8])); H = KV * rep
            dh = int(rng.choice([64, 128, 128]))
            if kind == "prefill":
                n = int(rng.choice([1, 2, 17, 40, 63, 64, 65, 128, 129, 255, 256, 257, 300, 511, 512, 640, 1000, 2048]))
                pos0 = int(rng.choice([0, 0, 1, 63, 64, 100, 500, 1000, 3000]))
            else:
                n, pos0 = 1, int(rng.choice([0, 1, 30, 31, 32, 63, 64, 255, 256, 383, 384, 385, 511, 513, 1000, 2047, 2048, 3000, 5000]))
            S = pos0 + n
            S_max = (S + 63) // 64 * 64 + 64 * int(rng.integers(0, 3))
            W = 0 if rng.random() < 0.5 else int(rng.choice([1, 16, 33, 64, 100, 1000, 4096]))
            q = rb(n, H, dh); k = rb(S_max, KV, dh); v = rb(S_max, KV, dh)
            vt = v.permute(1, 2, 0).contiguous()
            ctx = torch.empty(n, H * dh, device="cuda", dtype=torch.bfloat16)
            if kind == "prefill":
                if W:
                    check(lib.sm_llm_attention_window(q.data_ptr(), k.data_ptr(), vt.data_ptr(), n, pos0, H, KV, dh, S_max, W, ctx.data_ptr(), st()))
                else:
                    check(lib.sm_llm_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), n, pos0, H, KV, dh, S_max, ctx.data_ptr(), st()))
            else:
                ws = torch.empty(32 * H * (dh + 2), device="cuda")
                if W:
                    check(lib.sm_llm_decode_attention_window(q.data_ptr(), k.data_ptr(), vt.data_ptr(), pos0, H, KV, dh, S_max, W, ws.data_ptr(), 32, ctx.data_ptr(), st()))
                else:
                    check(lib.sm_llm_decode_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), pos0, H, KV, dh, S_max, ws.data_ptr(), 32, ctx.data_ptr(), st()))
            kk, vv = k[:S].float().repeat_interleave(rep, dim=1), v[:S].float().repeat_interleave(rep, dim=1)
            s = torch.einsum("qhd,khd->hqk", q.float(), kk) * dh ** -0.5
            pos = torch.arange(pos0, pos0 + n, device="cuda")
            keys = torch.arange(S, device="cuda")
            mask = keys[None, :] > pos[:, None]
            if W:
                mask = mask | (keys[None, :] <= pos[:, None] - W)
            ref = torch.einsum("hqk,khd->qhd", torch.softmax(s.masked_fill(mask[None], float("-inf")), -1), vv).reshape(n, H * dh)
            desc = {"kind": kind, "n": n, "pos0": pos0, "W": W, "H": H, "KV": KV, "dh": dh, "S_max": S_max}
    except StreamMindHipError as e:
        key = str(e)[:70]
        refused[key] = refused.get(key, 0) + 1
        continue
    torch.cuda.synchronize()
    kinds[kind] += 1
    err = relerr(ctx, ref)
    if not (err < 8e-3) or not bool(torch.isfinite(ctx.float()).all()):
        bad.append({**desc, "err": err})
print(json.dumps({"cases": sum(kinds.values()), "by_kind": kinds, "seed": seed, "refused_with_reason": refused, "mismatches": len(bad), "first_bad": bad[:8], "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)

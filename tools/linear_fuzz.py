"""sm_linear on RANDOM shapes and option sets against an fp64 reference on the same rounded operands: rows around every dispatch boundary (16 / 17, 32 / 33,
64 / 65, 128 / 129, 191 / 192, 256 / 257, ...), N from 2 to 6144, K from 32 to 14336, bf16 / fp32 (plain and hi/lo) / fp16 activations, bf16 / fp16 / fp8 weights,
bias, activation, residual, 16-bit copy, dual (gate / up) weights, SwiGLU-dual image, post-LayerNorm / RMSNorm.  One JSON line; exit code 1 on a mismatch.
    python tools/linear_fuzz.py [N=600] [seed=1]"""
import os, sys, json, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from streammind_amd import native, _lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
MS = [1, 2, 3, 4, 5, 8, 15, 16, 17, 20, 28, 31, 32, 33, 40, 48, 63, 64, 65, 100, 127, 128, 129, 191, 192, 200, 255, 256, 257, 300, 328, 511, 512, 577, 600, 1154, 2048]
NS = [2, 6, 16, 48, 96, 100, 256, 384, 1000, 1024, 2048, 3072, 4096, 6144]
KS = [32, 64, 96, 128, 256, 512, 1024, 1056, 4096, 14336]
ACTS = {0: lambda t: t, 1: lambda t: t * torch.sigmoid(1.702 * t), 2: lambda t: torch.nn.functional.leaky_relu(t, 0.01), 3: torch.nn.functional.softplus,
        4: torch.nn.functional.silu, 5: lambda t: torch.nn.functional.gelu(t)}


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=g, device="cuda") * scale


def fp8_dequant(w):          # the mode's definition: per-row scale max|w| / 448, e4m3 round to nearest even, q * s in fp32
    s = w.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-30) / 448.0
    q = (w.float() / s).to(torch.float8_e4m3fn).float()
    return q * s


bad, done, skipped = [], 0, 0
kinds = {}
t0 = time.time()
while done < n_cases:
    M, N, K = int(rng.choice(MS)), int(rng.choice(NS)), int(rng.choice(KS))
    if M * N * K > 2.2e10 or N * K > 6144 * 4096 * 2:
        continue
    wk = rng.choice(["bf16", "bf16", "bf16", "fp16", "fp8"])
    xk = rng.choice(["bf16", "f32", "f32p"]) if wk != "fp16" else "f16"        # (native.linear expresses fp16 operands through fp16 x)
    if M > 32 and xk != "bf16" and xk != "f16":
        xk = "f16" if wk == "fp16" else "bf16"                               # the tiled kernels take 16-bit activations
    dual = bool(rng.random() < 0.2) and (M <= 32 or (wk == "fp8" and M <= 64 and xk == "bf16")) and N >= 16 and not (wk == "fp16" and M > 32)
    swig = (not dual) and bool(rng.random() < 0.12) and M > 16 and xk in ("bf16", "f16") and N % 256 == 0 and wk != "fp8"
    use_bias, act = bool(rng.random() < 0.5), int(rng.choice([0, 0, 1, 2, 3, 4, 5]))
    use_res = bool(rng.random() < 0.4) and not dual and not swig
    out16 = bool(rng.random() < 0.3) and not swig
    pln = None
    if not dual and not swig and not out16 and rng.random() < 0.25 and N % 4 == 0:
        pln = "ln" if rng.random() < 0.5 else "rms"
    if swig:
        act = 6; use_bias = bool(rng.random() < 0.3)
    if dual:
        act = 0
    wdt = torch.float16 if wk == "fp16" else torch.bfloat16
    w = rnd(N, K, scale=K ** -0.5).to(wdt)
    w2 = rnd(N, K, scale=K ** -0.5).to(wdt) if dual else None
    xf = rnd(M, K)
    if xk in ("bf16", "f16"):
        x = xf.to(torch.bfloat16 if xk == "bf16" else torch.float16); x_ref = x.double()
    else:
        x = xf
        x_ref = x.double() if xk == "f32p" else x.to(wdt).double()           # hi / lo pair ~ fp32; plain: one rounding to the operand type
    if wk == "fp8":
        wp, sc = native.pack_weight_fp8(w); w_ref = fp8_dequant(w).double()
        w2p, sc2 = native.pack_weight_fp8(w2) if dual else (None, None); w2_ref = fp8_dequant(w2).double() if dual else None
        # the fp8 image is streamed (scale on the fp32 sums) up to 16 rows always, up to 32 rows (64 of bf16 x) when the LDS-shared kernel takes the shape
        # (>= 8 k-steps, N a multiple of 4 or below 4); otherwise it is expanded with the scale folded in: bf16(q * s)
        lds_ok = K >= 256 and (N % 4 == 0 or N < 4)
        streamed = M <= 16 or (lds_ok and (M <= 32 or (M <= 64 and xk == "bf16")))
        if not streamed:
            w_ref = w_ref.float().bfloat16().double()
            if dual:
                w2_ref = w2_ref.float().bfloat16().double()
    else:
        wp, sc, w_ref = native.pack_weight(w), None, w.double()
        w2p, sc2, w2_ref = (native.pack_weight(w2), None, w2.double()) if dual else (None, None, None)
    bias = rnd(N, scale=0.1) if use_bias else None
    res = rnd(M, N) if use_res else None
    y = x_ref @ w_ref.t()
    if swig:
        F = N // 2
        if bias is not None:
            y = y + bias.double()
        ref = torch.nn.functional.silu(y[:, :F]) * y[:, F:]
    elif dual:
        if bias is not None:
            y = y + bias.double()
        ref = torch.nn.functional.silu(y) * (x_ref @ w2_ref.t())
    else:
        if bias is not None:
            y = y + bias.double()
        ref = ACTS[act](y)
        if res is not None:
            ref = ref + res.double()
    kw = dict(bias=bias, act=act, residual=res, precise=(xk == "f32p"), w_scale=sc, w2p=w2p, w2_scale=sc2)
    o16 = torch.empty(M, N, dtype=torch.float16 if wk == "fp16" else torch.bfloat16, device="cuda") if out16 and not (dual or swig) else None
    ln_out = None
    if pln:
        gamma, beta = rnd(N).abs() + 0.5, (rnd(N, scale=0.1) if pln == "ln" else None)
        ln_out = torch.empty(M, N, dtype=torch.float32 if rng.random() < 0.5 else wdt, device="cuda")
        kw["post_ln"] = (gamma, beta, 1e-5, ln_out)
    desc = {"M": M, "N": N, "K": K, "w": wk, "x": xk, "dual": dual, "swiglu_image": swig, "bias": use_bias, "act": act, "residual": use_res, "out16": o16 is not None, "post_ln": pln}
    out_is_16 = False
    try:
        if swig:
            out = torch.empty(M, N // 2, dtype=wdt, device="cuda")
            native.linear(x, wp, N, K, out=out, **kw)
            got = out.double(); out_is_16 = True
        elif dual and rng.random() < 0.5:
            out = torch.empty(M, N, dtype=wdt, device="cuda")
            native.linear(x, wp, N, K, out=out, **kw)
            got = out.double(); out_is_16 = True
        else:
            got = native.linear(x, wp, N, K, out16=o16, **kw).double()
    except _lib.StreamMindHipError as e:
        skipped += 1                                                         # a documented refusal of the combination (the message says which)
        kinds["refused: " + str(e)[:60]] = kinds.get("refused: " + str(e)[:60], 0) + 1
        continue
    torch.cuda.synchronize()
    scale = float(ref.abs().max().clamp_min(1e-6))
    tol = 2.0 ** -7 if out_is_16 else 3e-5
    if xk == "f32p":
        tol = max(tol, 2e-4)                                                 # the hi / lo pair carries 16 mantissa bits of x
    err = float((got - ref).abs().max()) / scale
    fails = []
    if not (err < tol):
        fails.append(("out", err, tol))
    if o16 is not None:
        e16 = float((o16.double() - ref).abs().max()) / scale
        if not (e16 < 2.0 ** -7):
            fails.append(("out16", e16, 2.0 ** -7))
    if pln:
        r32 = got                                                            # the norm is taken of the fp32 rows the call wrote
        if pln == "ln":
            mu = r32.mean(dim=1, keepdim=True); var = ((r32 - mu) ** 2).mean(dim=1, keepdim=True)
            lref = (r32 - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
        else:
            lref = r32 / torch.sqrt((r32 ** 2).mean(dim=1, keepdim=True) + 1e-5) * gamma.double()
        el = float((ln_out.double() - lref).abs().max()) / float(lref.abs().max().clamp_min(1e-6))
        tl = 3e-5 if ln_out.dtype == torch.float32 else 2.0 ** -7
        if not (el < tl):
            fails.append(("post_ln", el, tl))
    if fails:
        bad.append({**desc, "fails": fails})
    done += 1
print(json.dumps({"cases": done, "seed": seed, "refused_with_reason": skipped, "refusals": kinds, "mismatches": len(bad), "first_bad": bad[:8], "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)

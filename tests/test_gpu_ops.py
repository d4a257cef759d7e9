"""Operator-level parity on a real MI355X: every C-ABI kernel against the oracle's plain-torch arithmetic on the
same seeded inputs.  Integer/index results are exact; floating point tolerances are stated per test."""
import ctypes as C

import numpy as np

import pytest
import torch

from oracle import streammind_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from streammind_amd import native, _lib
    _lib.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return native


def rnd(shape, seed, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


def relerr(a, b):
    return ((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (3, 48, 256), (8, 4096, 4096), (16, 288, 8192), (1, 4096, 14336),
                                   (5, 1024, 96), (2, 2, 4096), (1, 8192, 256), (17, 64, 128), (28, 4096, 4096), (32, 288, 8192)])
@pytest.mark.parametrize("mode", ["f32_precise", "f32_single", "bf16"])
def test_skinny_linear(nat, M, N, K, mode):
    """weight-streaming GEMV path vs fp32 matmul on the bf16-representable weights.
    precise (hi/lo split) : 2e-5 relative (fp32-class);  single bf16 activation rounding : exact vs the oracle that
    rounds x the same way, 2e-5."""
    if K % 8:
        pytest.skip("K")
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = rnd((M, K), 2)
    bias = rnd((N,), 3, 0.1)
    res = rnd((M, N), 4)
    wp = nat.pack_weight(w.cuda().bfloat16())
    if mode == "bf16":
        xg, xr = x.cuda().bfloat16(), O.bf16_round(x)
    elif mode == "f32_single":
        xg, xr = x.cuda(), O.bf16_round(x)
    else:
        xg, xr = x.cuda(), x
    y = nat.linear(xg, wp, N, K, bias=bias.cuda(), act=2, residual=res.cuda(), precise=(mode == "f32_precise"))
    ref = O.leaky_relu(xr.double() @ w.double().t() + bias.double()).float() + res
    assert relerr(y, ref) < 2e-5


@pytest.mark.parametrize("M", [4, 28])
def test_skinny_dual_swiglu(nat, M):
    N, K = 14336, 4096
    wg, wu = O.bf16_round(rnd((N, K), 1, K ** -0.5)), O.bf16_round(rnd((N, K), 2, K ** -0.5))
    x = rnd((M, K), 3)
    y = nat.linear(x.cuda(), nat.pack_weight(wg.cuda().bfloat16()), N, K, w2p=nat.pack_weight(wu.cuda().bfloat16()), precise=True)
    ref = (O.silu(x.double() @ wg.double().t()) * (x.double() @ wu.double().t())).float()
    assert relerr(y, ref) < 3e-5


@pytest.mark.parametrize("M,N,K", [(33, 128, 64), (300, 384, 256), (577, 1024, 1024), (1154, 3072, 1024), (1000, 200, 4096)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_tiled_gemm(nat, M, N, K, out_dtype):
    """LDS-tiled MFMA GEMM: bf16 operands, fp32 accumulate -> 1e-5 relative against fp64 on the same bf16 inputs
    (bf16 output: one extra rounding, 2^-8)."""
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = O.bf16_round(rnd((M, K), 2))
    bias = rnd((N,), 3, 0.1)
    res = rnd((M, N), 4)
    y = nat.linear(x.cuda().bfloat16(), nat.pack_weight(w.cuda().bfloat16()), N, K, bias=bias.cuda(), act=1,
                   residual=res.cuda(), out_dtype=out_dtype)
    ref = (O.quick_gelu(x.double() @ w.double().t() + bias.double()) + res.double()).float()
    assert relerr(y, ref) < (1e-5 if out_dtype == torch.float32 else 5e-3)


@pytest.mark.parametrize("M,N,K,f16", [(577, 1024, 1024, False), (577, 1024, 4096, False), (1154, 1024, 4096, True), (16156, 1024, 1024, False), (300, 384, 256, False)])
def test_linear_post_ln(nat, M, N, K, f16):
    """sm_linear_t.post_ln_*: the LayerNorm of the finished row behind a residual product (ViT out-proj -> LN2, fc2 -> next LN1).  One
    frame (577 rows, N = 1024: few tiles) runs it as split-K slabs + ONE slab-sum / bias / residual / LayerNorm pass; many rows and other
    widths end with the separate norm launch.  Either way: the fp32 rows are the product (1e-5 of fp64 on the same 16-bit operands), and
    the 16-bit LayerNorm output is what sm_norm_ex writes for those very fp32 rows: the same two-pass arithmetic -- equal bit for bit
    where the call ends in that launch, and within ONE 16-bit ulp on at most 0.1 % of the elements where the fused pass computed it
    (a last-bit difference of an fp32 intermediate landing on a rounding boundary)."""
    from streammind_amd._lib import load, check, SM_OP_F16, SM_OP_BF16
    lib = load()
    dt = torch.float16 if f16 else torch.bfloat16
    w = rnd((N, K), 1, K ** -0.5).to(dt).float()
    x = rnd((M, K), 2).to(dt).float()
    bias, res = rnd((N,), 3, 0.1), rnd((M, N), 4)
    g, b = 1 + rnd((N,), 5, 0.1), rnd((N,), 6, 0.1)
    ln_out = torch.empty(M, N, device="cuda", dtype=dt)
    resg, gg, bg = res.cuda(), g.cuda(), b.cuda()
    y = nat.linear(x.cuda().to(dt), nat.pack_weight(w.cuda().to(dt)), N, K, bias=bias.cuda(), residual=resg, out=resg,
                   post_ln=(gg, bg, 1e-5, ln_out))
    ref = (x.double() @ w.double().t() + bias.double() + res.double()).float()
    assert relerr(y, ref) < 1e-5
    want = torch.empty(M, N, device="cuda", dtype=dt)
    check(lib.sm_norm_ex(y.data_ptr(), M, N, N, gg.data_ptr(), bg.data_ptr(), 1e-5, 0, None, want.data_ptr(), N,
                         SM_OP_F16 if f16 else SM_OP_BF16, torch.cuda.current_stream().cuda_stream))
    a16, b16 = ln_out.cpu().view(torch.int16).int(), want.cpu().view(torch.int16).int()
    diff = (a16 - b16).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3, (int(diff.max()), float((diff > 0).float().mean()))
    if M > 4096 or N != 1024:
        assert torch.equal(ln_out.cpu(), want.cpu())          # the unfused route IS that launch
    assert relerr(ln_out, O.layer_norm(ref, g, b, 1e-5)) < (2e-3 if f16 else 1e-2)


@pytest.mark.parametrize("M", [1, 7, 16, 17, 28, 32])
@pytest.mark.parametrize("kind", ["rms_f32", "ln_leaky_f32", "rms_bf16"])
def test_skinny_linear_post_norm(nat, M, kind):
    """the connector / gate pass's norms behind their products (sm_linear_t.post_ln_*, weight-streaming path): RMSNorm or LayerNorm (+ leaky_relu)
    of the finished row, fp32 or 16-bit out.  17..32 rows: K-slice slabs + ONE slab-sum / residual / norm launch; fewer rows: the product,
    then the norm launch.  The fp32 row is the product (3e-5 of fp64: hi/lo activations), the normalised row the oracle's norm of it (2e-5;
    bf16 out: one rounding)."""
    N, K = 4096, 4096
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x, res = rnd((M, K), 2), rnd((M, N), 3)
    g, b = 1 + rnd((N,), 5, 0.1), rnd((N,), 6, 0.1)
    ln = kind.startswith("ln")
    out_dt = torch.bfloat16 if kind.endswith("bf16") else torch.float32
    nout = torch.empty(M, N, device="cuda", dtype=out_dt)
    gg, bg, resg = g.cuda(), b.cuda(), res.cuda()
    y = nat.linear(x.cuda(), nat.pack_weight(w.cuda().bfloat16()), N, K, residual=resg, precise=True,
                   post_ln=(gg, bg if ln else None, 1e-5, nout), post_ln_act=2 if ln else 0)
    ref = (x.double() @ w.double().t() + res.double()).float()
    assert relerr(y, ref) < 3e-5
    want = O.leaky_relu(O.layer_norm(y.cpu(), g, b, 1e-5)) if ln else O.rms_norm(y.cpu(), g, 1e-5)
    assert relerr(nout, want) < (8e-3 if out_dt == torch.bfloat16 else 2e-5)


@pytest.mark.parametrize("M", [1, 16, 28])
def test_skinny_linear_repeated_column_groups(nat, M):
    """sm_linear_t.x_rep: the event gate's repeat_kv folded into o_proj's operand addressing (builder.py:553-562 at seq-len 1: o_proj reads
    every kv head H / KV times).  Bit for bit the product on the materialised repeat."""
    KV, rep, dh, N = 8, 4, 128, 4096
    K = KV * rep * dh
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    v = rnd((M, KV * dh), 2)
    vrep = v.reshape(M, KV, 1, dh).expand(M, KV, rep, dh).reshape(M, K).contiguous()
    wp = nat.pack_weight(w.cuda().bfloat16())
    a = nat.linear(v.cuda(), wp, N, K, precise=True, x_rep=(rep, dh))
    bq = nat.linear(vrep.cuda(), wp, N, K, precise=True)
    assert torch.equal(a, bq)
    assert relerr(a, (vrep.double() @ w.double().t()).float()) < 3e-5


def test_pack_layout(nat):
    """the packed image is the documented permutation of W (integer-exact)."""
    N, K = 40, 70
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251
    wp = nat.pack_weight(w.cuda().bfloat16()).cpu().float()
    KS = (K + 31) // 32
    for n, k in [(0, 0), (17, 33), (39, 69), (15, 31), (16, 32)]:
        idx = ((n >> 4) * KS + (k >> 5)) * 512 + ((((k & 31) >> 3) << 4) + (n & 15)) * 8 + (k & 7)
        assert wp[idx].item() == w[n, k].item()
    assert wp.numel() == 3 * KS * 512 and wp.sum().item() == w.sum().item()


@pytest.mark.parametrize("kind", ["ln", "rms"])
def test_norm(nat, kind):
    from streammind_amd._lib import load, check
    lib = load()
    M, D = 37, 1024
    x, g, b = rnd((M, D), 1, 2.0) + 0.3, 1 + rnd((D,), 2, 0.1), rnd((D,), 3, 0.1)
    xg, gg, bg = x.cuda(), g.cuda(), b.cuda()
    of = torch.empty(M, D, device="cuda")
    ob = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.sm_norm(xg.data_ptr(), M, D, D, gg.data_ptr(), bg.data_ptr() if kind == "ln" else None, 1e-5, 0,
                      of.data_ptr(), ob.data_ptr(), D, st))
    ref = O.layer_norm(x, g, b, 1e-5) if kind == "ln" else O.rms_norm(x, g, 1e-5)
    assert relerr(of, ref) < 1e-5
    assert torch.equal(ob.cpu(), of.cpu().bfloat16())


@pytest.mark.parametrize("vmode", ["row_major_v", "pre_transposed_vt"])
@pytest.mark.parametrize("B,S,H,dh", [(2, 17, 2, 64), (1, 577, 16, 64), (3, 130, 4, 64), (2, 70, 2, 128)])
def test_vit_attention(nat, B, S, H, dh, vmode):
    """non-causal attention vs the oracle's mixed-precision statement (P rounded to bf16 for PV, fp32 normaliser):
    output bf16, tolerance 1 bf16 ulp-ish (8e-3 relative to max)."""
    from streammind_amd._lib import load, check
    lib = load()
    D = H * dh
    qkv = O.bf16_round(rnd((B * S, 3 * D), 1))
    Spad = (S + 63) // 64 * 64
    v = qkv[:, 2 * D:].reshape(B, S, H, dh)
    vt = torch.zeros(B, H, dh, Spad)
    vt[:, :, :, :S] = v.permute(0, 2, 3, 1)
    qg, vg = qkv.cuda().bfloat16(), vt.cuda().bfloat16()
    ctx = torch.empty(B * S, D, device="cuda", dtype=torch.bfloat16)
    if vmode == "row_major_v":
        check(lib.sm_vit_attention(qg.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0, 0, torch.cuda.current_stream().cuda_stream))
    else:
        check(lib.sm_vit_attention(qg.data_ptr(), vg.data_ptr(), ctx.data_ptr(), B, S, H, dh, Spad, 0, torch.cuda.current_stream().cuda_stream))
    q = qkv[:, :D].reshape(B, S, H, dh).transpose(1, 2)
    k = qkv[:, D:2 * D].reshape(B, S, H, dh).transpose(1, 2)
    vv = v.transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    e = torch.exp(s - s.max(-1, keepdim=True).values)
    ref = ((O.bf16_round(e) @ vv) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * S, D)
    assert relerr(ctx, ref) < 8e-3


def test_preprocess_matches_oracle_and_golden(nat, gold):
    """a1: u8 frame -> (x/255 - mean)/std; fp32 pixel_values within 1e-6 of the oracle / the reference golden, and
    the bf16 patch matrix equal to the bf16 rounding of the oracle's patchified tensor up to 1 bf16 ulp."""
    from streammind_amd._lib import load, check
    lib = load()
    g = gold("g1_preprocess")
    frames = O.synthetic_frames(2, 336, seed=int(g["seed"]))
    fg = frames.cuda()
    P, ldp = 576, 640
    patches = torch.empty(2 * P, ldp, device="cuda", dtype=torch.bfloat16)
    pix = torch.empty(2, 3, 336, 336, device="cuda")
    mean = (C.c_float * 3)(*O.CLIP_MEAN)
    std = (C.c_float * 3)(*O.CLIP_STD)
    check(lib.sm_preprocess_patches(fg.data_ptr(), 2, 336, 336, 14, mean, std, patches.data_ptr(), ldp, pix.data_ptr(), 0,
                                    torch.cuda.current_stream().cuda_stream))
    ref = O.preprocess_frames(frames)
    assert (pix.cpu() - ref).abs().max().item() < 1e-6
    assert (pix.cpu().flatten()[torch.as_tensor(g["idx"])] - torch.as_tensor(g["pixel_values_sample"])).abs().max().item() < 2e-6
    pref = O.vit_patchify(ref, O.VitCfg())
    got = patches.cpu().float().reshape(2, P, ldp)
    assert (got[:, :, 588:] == 0).all()
    assert (got[:, :, :588] - pref).abs().max().item() <= 2 ** -7 * pref.abs().max().item()


def test_mamba_step_kernels(nat):
    from streammind_amd._lib import load, check
    lib = load()
    cfg = O.ConnCfg(mm_hidden=64, d_model=128)
    di, ds, R, M = cfg.d_inner, cfg.d_state, cfg.dt_rank, 5
    W = O.make_conn_weights(cfg, 3)
    xz = rnd((M, 2 * di), 1)
    st = torch.cuda.current_stream().cuda_stream
    cs = torch.zeros(di, cfg.d_conv, device="cuda")
    xc = torch.empty(M, di, device="cuda")
    cw = W[O.CONN + "mixer.conv1d.weight"].reshape(di, -1).cuda().contiguous()
    cb = W[O.CONN + "mixer.conv1d.bias"].cuda()
    xzg = xz.cuda()
    check(lib.sm_mamba_conv_step(xzg.data_ptr(), M, di, cfg.d_conv, cs.data_ptr(), cw.data_ptr(), cb.data_ptr(), xc.data_ptr(), st))
    # oracle conv
    conv = torch.zeros(di, cfg.d_conv)
    ref_xc = []
    for m in range(M):
        conv = torch.roll(conv, -1, -1); conv[:, -1] = xz[m, :di]
        ref_xc.append(O.silu((conv * cw.cpu()).sum(-1) + cb.cpu()))
    ref_xc = torch.stack(ref_xc)
    assert relerr(xc, ref_xc) < 1e-5 and relerr(cs, conv) < 1e-6
    ldx = 64
    xdbl = torch.zeros(M, ldx); xdbl[:, :R + 2 * ds] = rnd((M, R + 2 * ds), 2)
    delta = O.softplus(rnd((M, di), 3))
    h = torch.zeros(di, ds, device="cuda")
    y = torch.empty(M, di, device="cuda")
    Al, Dp = W[O.CONN + "mixer.A_log"].cuda(), W[O.CONN + "mixer.D"].cuda()
    xdg, dg = xdbl.cuda(), delta.cuda()
    check(lib.sm_mamba_ssm_step(xc.data_ptr(), dg.data_ptr(), xdg.data_ptr(), ldx, R, xzg.data_ptr(), M, di, ds,
                                Al.data_ptr(), Dp.data_ptr(), h.data_ptr(), y.data_ptr(), st))
    A = -torch.exp(Al.cpu())
    hs = torch.zeros(di, ds)
    ref_y = []
    xcc = xc.cpu()
    for m in range(M):
        Bm, Cm = xdbl[m, R:R + ds], xdbl[m, R + ds:R + 2 * ds]
        hs = torch.exp(delta[m][:, None] * A) * hs + (delta[m] * xcc[m])[:, None] * Bm[None]
        ref_y.append((hs @ Cm + Dp.cpu() * xcc[m]) * O.silu(xz[m, di:]))
    assert relerr(y, torch.stack(ref_y)) < 2e-5 and relerr(h, hs) < 2e-5


def test_gate_decide_and_argmax(nat):
    from streammind_amd._lib import load, check
    lib = load()
    st = torch.cuda.current_stream().cuda_stream
    lg = torch.tensor([[0.3, 0.3], [0.1, 0.2], [0.5, -0.5], [-1.0, -0.9]], device="cuda")
    dec = torch.empty(4, dtype=torch.int32, device="cuda")
    check(lib.sm_gate_decide(lg.data_ptr(), 4, dec.data_ptr(), st))
    assert dec.tolist() == [O.gate_decision(l) for l in lg.cpu()] == [0, 1, 0, 1]
    v = rnd((32000,), 5); v[123] = v[31999] = v.max() + 1          # tie -> lowest index (torch.argmax)
    out = torch.empty(1, dtype=torch.int32, device="cuda")
    vg = v.cuda()
    check(lib.sm_argmax(vg.data_ptr(), 32000, out.data_ptr(), st))
    assert out.item() == 123 == int(torch.argmax(v))


@pytest.mark.parametrize("pos,H,KV,dh", [(0, 32, 8, 128), (37, 32, 8, 128), (1000, 32, 8, 128), (4095, 32, 8, 128), (200, 2, 1, 128), (77, 4, 4, 64),
                                         (30, 32, 8, 128), (31, 32, 8, 128), (32, 32, 8, 128), (255, 32, 8, 128), (256, 32, 8, 128), (511, 32, 8, 128),
                                         (513, 32, 8, 128), (2047, 32, 8, 128), (2048, 32, 8, 128), (300, 16, 1, 128)])
def test_decode_attention_split_keys(nat, pos, H, KV, dh):
    """single-token decode attention vs plain softmax attention over the cache: bf16 output, 8e-3 of max.  Up to 2048 keys at
    head_dim 128 and <= 384 keys this is the one-launch kernel (in-block merge of the 8 per-wave partials; its longer contexts are
    covered through the batched decode, test_decode_attention_many_streams), beyond that key-split + merge."""
    from streammind_amd._lib import load, check
    lib = load()
    S_max = 4096
    g = torch.Generator().manual_seed(pos + H)
    q = O.bf16_round(torch.randn(H, dh, generator=g))
    k = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    v = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    k[pos + 1:] = 0; v[pos + 1:] = 0
    qg, kg = q.cuda().bfloat16(), k.cuda().bfloat16()
    vt = v.permute(1, 2, 0).contiguous().cuda().bfloat16()             # [KV][dh][S_max]
    ws = torch.empty(32 * H * (dh + 2), device="cuda")
    ctx = torch.empty(H, dh, device="cuda", dtype=torch.bfloat16)
    check(lib.sm_llm_decode_attention(qg.data_ptr(), kg.data_ptr(), vt.data_ptr(), pos, H, KV, dh, S_max, ws.data_ptr(), 32,
                                      ctx.data_ptr(), torch.cuda.current_stream().cuda_stream))
    rep = H // KV
    kk = k[:pos + 1].repeat_interleave(rep, dim=1)
    vv = v[:pos + 1].repeat_interleave(rep, dim=1)
    s = torch.einsum("hd,khd->hk", q, kk) * dh ** -0.5
    ref = torch.einsum("hk,khd->hd", torch.softmax(s, -1), vv)
    assert relerr(ctx, ref) < 8e-3
    if dh == 128 and pos < 384:
        # a caller-owned cache need not be clean behind the last key: NaN there must not reach the output
        kg[pos + 1:] = float("nan"); vt[:, :, pos + 1:] = float("nan")
        ctx2 = torch.empty_like(ctx)
        check(lib.sm_llm_decode_attention(qg.data_ptr(), kg.data_ptr(), vt.data_ptr(), pos, H, KV, dh, S_max, ws.data_ptr(), 32,
                                          ctx2.data_ptr(), torch.cuda.current_stream().cuda_stream))
        assert torch.equal(ctx2, ctx)


@pytest.mark.parametrize("pos,W,H,KV,dh", [(10, 64, 32, 8, 128), (63, 64, 32, 8, 128), (64, 64, 32, 8, 128), (200, 64, 32, 8, 128), (1000, 100, 32, 8, 128),
                                           (4095, 4096, 32, 8, 128), (5000, 4096, 32, 8, 128), (6143, 4096, 32, 8, 128), (3000, 1000, 4, 4, 64), (700, 33, 2, 1, 128)])
def test_decode_attention_sliding_window(nat, pos, W, H, KV, dh):
    """Mistral's sliding window at decode time (HF MistralModel: keys k <= p - window are masked, i.e. the query at position p sees the
    newest `window` keys, itself included): one-launch kernel and key-split + merge, window starting inside / at / before a key tile,
    contexts beyond the window (the case the checkpoint's sliding_window = 4096 exists for).  bf16 output, 8e-3 of max; with the window
    at least as long as the context the result is bit for bit the plain call's."""
    from streammind_amd._lib import load, check
    lib = load()
    S_max = 6144
    g = torch.Generator().manual_seed(pos + W)
    q = O.bf16_round(torch.randn(H, dh, generator=g))
    k = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    v = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    qg, kg = q.cuda().bfloat16(), k.cuda().bfloat16()
    vt = v.permute(1, 2, 0).contiguous().cuda().bfloat16()
    ws = torch.empty(32 * H * (dh + 2), device="cuda")
    ctx = torch.empty(H, dh, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.sm_llm_decode_attention_window(qg.data_ptr(), kg.data_ptr(), vt.data_ptr(), pos, H, KV, dh, S_max, W, ws.data_ptr(), 32, ctx.data_ptr(), st))
    lo = max(0, pos - W + 1)
    rep = H // KV
    kk, vv = k[lo:pos + 1].repeat_interleave(rep, dim=1), v[lo:pos + 1].repeat_interleave(rep, dim=1)
    ref = torch.einsum("hk,khd->hd", torch.softmax(torch.einsum("hd,khd->hk", q, kk) * dh ** -0.5, -1), vv)
    assert relerr(ctx, ref) < 8e-3
    if W > pos:
        plain = torch.empty_like(ctx)
        check(lib.sm_llm_decode_attention(qg.data_ptr(), kg.data_ptr(), vt.data_ptr(), pos, H, KV, dh, S_max, ws.data_ptr(), 32, plain.data_ptr(), st))
        assert torch.equal(plain, ctx)


@pytest.mark.parametrize("n,pos0,W,H,KV,dh", [(40, 0, 16, 4, 2, 128), (300, 0, 64, 8, 2, 128), (130, 500, 200, 8, 8, 64), (257, 4000, 4096, 32, 8, 128), (64, 100, 1000, 4, 1, 128)])
def test_prefill_attention_sliding_window(nat, n, pos0, W, H, KV, dh):
    """the causal prefill kernel with the window mask: n new queries at positions pos0.. against the cache [0, pos0 + n), each seeing
    keys (p - W, p].  vs fp32 softmax attention on the same bf16 operands, bf16 output: 8e-3 of max."""
    from streammind_amd._lib import load, check
    lib = load()
    S_max = ((pos0 + n + 63) // 64) * 64
    g = torch.Generator().manual_seed(n + W)
    q = O.bf16_round(torch.randn(n, H, dh, generator=g))
    k = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    v = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    qg, kg = q.reshape(n, H * dh).cuda().bfloat16(), k.cuda().bfloat16()
    vt = v.permute(1, 2, 0).contiguous().cuda().bfloat16()
    ctx = torch.empty(n, H * dh, device="cuda", dtype=torch.bfloat16)
    check(lib.sm_llm_attention_window(qg.data_ptr(), kg.data_ptr(), vt.data_ptr(), n, pos0, H, KV, dh, S_max, W, ctx.data_ptr(), torch.cuda.current_stream().cuda_stream))
    S = pos0 + n
    rep = H // KV
    kk, vv = k[:S].repeat_interleave(rep, dim=1), v[:S].repeat_interleave(rep, dim=1)
    s = torch.einsum("qhd,khd->hqk", q, kk) * dh ** -0.5
    pos = torch.arange(pos0, pos0 + n)
    mask = (torch.arange(S)[None, :] > pos[:, None]) | (torch.arange(S)[None, :] <= pos[:, None] - W)
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(s.masked_fill(mask[None], float("-inf")), -1), vv).reshape(n, H * dh)
    assert relerr(ctx, ref) < 8e-3


@pytest.mark.parametrize("n,pos0,W,H,KV", [(2048, 0, 0, 32, 8), (1000, 0, 0, 32, 8), (257, 4000, 4096, 32, 8), (300, 100, 64, 8, 2), (130, 7, 0, 4, 4), (64, 0, 0, 6, 3),
                                           (1, 50, 0, 8, 2), (513, 1535, 0, 32, 8), (700, 0, 200, 12, 4)])
def test_prefill_attention_kernel_equals_the_tile_kernel_bit_for_bit(nat, n, pos0, W, H, KV):
    """round 6: the dedicated causal prefill kernel (head_dim 128: one query tile per block, longest first, K / V^T by LDS-DMA, batched fragment reads) keeps the
    general tile kernel's arithmetic -- same outputs bit for bit on full tiles, ragged tails, a cache offset, windows, head counts that are not a multiple of 8
    -- and both sit within 8e-3 of fp32 softmax attention on the same bf16 operands."""
    from streammind_amd._lib import load, check
    from streammind_amd import native
    lib = load()
    dh = 128
    S_max = ((pos0 + n + 63) // 64) * 64
    g = torch.Generator().manual_seed(n + pos0 + W)
    q = O.bf16_round(torch.randn(n, H, dh, generator=g))
    k = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    v = O.bf16_round(torch.randn(S_max, KV, dh, generator=g))
    qg, kg = q.reshape(n, H * dh).cuda().bfloat16(), k.cuda().bfloat16()
    vt = v.permute(1, 2, 0).contiguous().cuda().bfloat16()
    out = []
    try:
        for on in (1, 0):
            native.set_prefill_attention_kernel(on)
            ctx = torch.full((n, H * dh), float("nan"), device="cuda", dtype=torch.bfloat16)
            check(lib.sm_llm_attention_window(qg.data_ptr(), kg.data_ptr(), vt.data_ptr(), n, pos0, H, KV, dh, S_max, W, ctx.data_ptr(), torch.cuda.current_stream().cuda_stream))
            out.append(ctx)
    finally:
        native.set_prefill_attention_kernel(-1)
    assert torch.equal(out[0], out[1])
    S = pos0 + n
    rep = H // KV
    kk, vv = k[:S].repeat_interleave(rep, dim=1), v[:S].repeat_interleave(rep, dim=1)
    s_ = torch.einsum("qhd,khd->hqk", q, kk) * dh ** -0.5
    pos = torch.arange(pos0, pos0 + n)
    mask = torch.arange(S)[None, :] > pos[:, None]
    if W > 0:
        mask = mask | (torch.arange(S)[None, :] <= pos[:, None] - W)
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(s_.masked_fill(mask[None], float("-inf")), -1), vv).reshape(n, H * dh)
    assert relerr(out[0], ref) < 8e-3


@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (3, 48, 96), (8, 4096, 4096), (1, 4096, 14336), (16, 288, 8192), (2, 2, 4096)])
@pytest.mark.parametrize("mode", ["f32_precise", "bf16"])
def test_skinny_linear_fp8_weights(nat, M, N, K, mode):
    """opt-in weight-only fp8 (config 5): the HIP quantiser + fp8 streaming kernel against the oracle's definition
    (per-row scale max|w|/448, OCP e4m3fn, RNE).  2e-5 relative when activations are carried as hi/lo, 2e-5 with the
    same single bf16 rounding of x on both sides."""
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = rnd((M, K), 2)
    bias = rnd((N,), 3, 0.1)
    wq, sc = nat.pack_weight_fp8(w.cuda().bfloat16())
    wd, s_ref = O.fp8_quantize_rows(w)
    assert relerr(sc, s_ref) < 1e-6
    xg, xr = (x.cuda(), x) if mode == "f32_precise" else (x.cuda().bfloat16(), O.bf16_round(x))
    y = nat.linear(xg, wq, N, K, bias=bias.cuda(), w_scale=sc, precise=(mode == "f32_precise"))
    ref = (xr.double() @ wd.double().t() + bias.double()).float()
    assert relerr(y, ref) < 2e-5
    # and the quantisation itself is a ~2^-4 relative perturbation per weight, a few % on the outputs: sanity bound vs bf16 weights
    ref_bf16 = (xr.double() @ w.double().t() + bias.double()).float()
    assert relerr(y, ref_bf16) < 0.2


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (1, 4096, 14336), (5, 64, 96), (16, 256, 8192), (2, 32, 32 * 37), (1, 48, 64 * 67), (3, 16, 32)])
@pytest.mark.parametrize("x16,out16", [(True, False), (False, True)])
def test_skinny_fp8_ring_and_vector_epilogue(nat, M, N, K, x16, out16):
    """the decode shapes of the fp8 weight-streaming kernel: no bias, residual rows, N a multiple of 16 -- the epilogue that
    fetches scales and residual as vectors ahead of the weights -- over chunk counts that leave every kind of tail behind
    the register ring (fewer chunks than the ring holds, a ragged rest, an odd number of k-steps, idle waves).  Against the
    oracle's definition, 2e-5 relative with the same single bf16 rounding of x on both sides."""
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = rnd((M, K), 2)
    res = rnd((M, N), 3)
    wq, sc = nat.pack_weight_fp8(w.cuda().bfloat16())
    wd = O.fp8_quantize_rows(w)[0]
    xr = O.bf16_round(x)
    xg = x.cuda().bfloat16() if x16 else xr.cuda()
    y = nat.linear(xg, wq, N, K, w_scale=sc, residual=res.cuda(), out_dtype=torch.bfloat16 if out16 else torch.float32)
    ref = (xr.double() @ wd.double().t() + res.double()).float()
    if out16:
        assert relerr(y.float(), ref) < 4e-3          # half a bf16 ulp of the largest output is 2^-8 = 3.9e-3
    else:
        assert relerr(y, ref) < 2e-5


def test_skinny_dual_fp8(nat):
    M, N, K = 4, 1024, 512
    wg, wu = O.bf16_round(rnd((N, K), 1, K ** -0.5)), O.bf16_round(rnd((N, K), 2, K ** -0.5))
    x = rnd((M, K), 3)
    qg, sg = nat.pack_weight_fp8(wg.cuda().bfloat16())
    qu, su = nat.pack_weight_fp8(wu.cuda().bfloat16())
    y = nat.linear(x.cuda(), qg, N, K, w2p=qu, w_scale=sg, w2_scale=su, precise=True)
    dg, du = O.fp8_quantize_rows(wg)[0], O.fp8_quantize_rows(wu)[0]
    ref = (O.silu(x.double() @ dg.double().t()) * (x.double() @ du.double().t())).float()
    assert relerr(y, ref) < 3e-5


@pytest.mark.parametrize("M,N,K,dual", [(1, 256, 4096, False), (1, 512, 4096, True), (3, 64, 256, False), (16, 48, 1024, True),
                                         (1, 32000, 4096, False), (2, 96, 64, False), (1, 128, 9216, False)])
def test_skinny_linear_fused_rmsnorm(nat, M, N, K, dual):
    """decode path: MistralRMSNorm folded into the weight-streaming product (norm_gamma).  Against the unfused pair
    sm_norm -> sm_linear within one bf16 ulp of a few activations (the row sum is taken in a different order), and against
    the fp32 definition bf16(gamma * x * rsqrt(mean x^2 + eps)) @ W^T."""
    x = rnd((M, K), 1, 3.0)
    gamma = 1.0 + rnd((K,), 2, 0.2)
    w = O.bf16_round(rnd((N, K), 3, K ** -0.5))
    wp = nat.pack_weight(w.cuda().bfloat16())
    w2 = O.bf16_round(rnd((N, K), 4, K ** -0.5)) if dual else None
    w2p = nat.pack_weight(w2.cuda().bfloat16()) if dual else None
    eps = 1e-5
    y = nat.linear(x.cuda(), wp, N, K, w2p=w2p, norm_gamma=gamma.cuda(), norm_eps=eps)
    xn = O.bf16_round(gamma * (x * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + eps).float()))
    f = (lambda t: (O.silu(t.double() @ w.double().t()) * (t.double() @ w2.double().t())).float()) if dual else \
        (lambda t: (t.double() @ w.double().t()).float())
    assert relerr(y, f(xn)) < 2e-3                      # a handful of activations may round the other way (2^-9 each)
    from streammind_amd._lib import load, check
    xg, gg = x.cuda(), gamma.cuda()
    xn_gpu = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    check(load().sm_norm(xg.data_ptr(), M, K, K, gg.data_ptr(), None, eps, 0, None, xn_gpu.data_ptr(), K,
                         torch.cuda.current_stream().cuda_stream))
    y2 = nat.linear(xn_gpu, wp, N, K, w2p=w2p)
    assert relerr(y, y2.cpu()) < 2e-3
    # (not bit-identical in general: the row sum is taken in another order, and ONE activation rounding the other way after a
    #  1-ulp change of rstd moves the low bits of every output)
    if not dual or N % 16 == 0:
        # the same with fp8 weights (config 5): against the unfused fp8 pair
        wq, sc = nat.pack_weight_fp8(w.cuda().bfloat16())
        w2q, sc2 = nat.pack_weight_fp8(w2.cuda().bfloat16()) if dual else (None, None)
        y8 = nat.linear(x.cuda(), wq, N, K, w2p=w2q, w_scale=sc, w2_scale=sc2, norm_gamma=gamma.cuda(), norm_eps=eps)
        y8u = nat.linear(xn_gpu, wq, N, K, w2p=w2q, w_scale=sc, w2_scale=sc2)
        assert relerr(y8, y8u.cpu()) < 2e-3


@pytest.mark.parametrize("M,N,K,dual", [(20, 512, 1056, False), (28, 1024, 512, True), (32, 4096, 4096, False), (17, 2048, 14336, True), (48, 384, 1024, False), (64, 1024, 4096, True), (40, 4096, 512, False),
                                        (300, 768, 4096, False)])
def test_linear_fp8_weights_more_than_16_rows(nat, M, N, K, dual):
    """fp8 weights with M > 16 rows.  17..64 rows (a batched decode step of up to 64 streams, the connector / gate pass): the LDS-shared weight-streaming
    kernel reads the fp8 image itself and the row scale is applied to the fp32 sums, as for <= 16 rows -- the mode's own definition (q * s in fp32)
    to 1e-5 relative.  More rows (prefill chunks, teacher-forced evaluation): the packed fp8 image is expanded to a bf16 scratch image with the row
    scale folded in (bf16(q*s), RNE) and the bf16 kernels run on it -- that definition exactly (1e-5), and within the extra 2^-9-per-weight
    rounding of q*s in fp32 (4e-3 relative)."""
    streamed = M <= 64                                         # the fp8 image is what the kernel reads (33..64 rows: four 16-row blocks per weight load)
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = O.bf16_round(rnd((M, K), 2))
    wq, sc = nat.pack_weight_fp8(w.cuda().bfloat16())
    wd = O.fp8_quantize_rows(w)[0]
    if dual:
        w2 = O.bf16_round(rnd((N, K), 5, K ** -0.5))
        w2q, sc2 = nat.pack_weight_fp8(w2.cuda().bfloat16())
        w2d = O.fp8_quantize_rows(w2)[0]
        y = nat.linear(x.cuda().bfloat16(), wq, N, K, w2p=w2q, w_scale=sc, w2_scale=sc2)
        f = lambda a, b: (O.silu(x.double() @ a.double().t()) * (x.double() @ b.double().t())).float()
        assert relerr(y, f(wd, w2d) if streamed else f(O.bf16_round(wd), O.bf16_round(w2d))) < 1e-5
        assert relerr(y, f(wd, w2d)) < 4e-3
    else:
        bias = rnd((N,), 3, 0.1)
        y = nat.linear(x.cuda().bfloat16(), wq, N, K, bias=bias.cuda(), w_scale=sc)
        f = lambda a: (x.double() @ a.double().t() + bias.double()).float()
        assert relerr(y, f(wd) if streamed else f(O.bf16_round(wd))) < 1e-5
        assert relerr(y, f(wd)) < 4e-3


@pytest.mark.parametrize("M,N,K,act,use_res,out16", [(17, 256, 128, 0, False, False), (28, 512, 1024, 0, True, False), (300, 768, 4096, 1, False, True),
                                                   (328, 6144, 4096, 0, False, False), (577, 200, 384, 0, True, False), (2048, 1024, 14336, 0, True, False)])
def test_linear_fp8_mfma_vs_its_definition(nat, M, N, K, act, use_res, out16):
    """BASELINE configs[4] ("CDNA4 fp8 MFMA"): SM_W_FP8_MFMA with more than 16 rows -- activation rows quantised to e4m3 by the HIP
    quantiser (scale max|x|/448, the weights' rule), fp8 x fp8 products on v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 1), fp32
    accumulation, sx[m] * sw[n] on the way out.  Against the mode's definition in fp64 on the SAME quantised operands
    (O.fp8_quantize_rows for both): every fp8 product is exact in fp32, so what is left is accumulation order -- 2e-5 of the
    largest output (fp32 outputs), one 16-bit rounding more for bf16 outputs.  Shapes: one and many k-blocks, ragged M / N, the
    64- and the 128-row tile (M = 2048 x N = 1024: 512 tiles), K = 14336 (Mistral's down-projection)."""
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = O.bf16_round(rnd((M, K), 2))
    x[3] *= 37.0; x[M - 1] *= 0.01                      # rows of very different magnitude: the scale is per row
    x = O.bf16_round(x)
    bias = rnd((N,), 3, 0.1)
    res = rnd((M, N), 4) if use_res else None
    wq, sc = nat.pack_weight_fp8(w.cuda().bfloat16())
    wd, sw = O.fp8_quantize_rows(w)
    xd, sx = O.fp8_quantize_rows(x)
    assert torch.equal(sc.cpu(), sw)
    y = nat.linear(x.cuda().bfloat16(), wq, N, K, bias=bias.cuda(), act=act, residual=None if res is None else res.cuda(), w_scale=sc,
                   fp8_mfma=True, out_dtype=torch.bfloat16 if out16 else torch.float32)
    ref = xd.double() @ wd.double().t() + bias.double()
    if act == 1:
        ref = O.quick_gelu(ref)
    if res is not None:
        ref = ref + res.double()
    assert relerr(y, ref.float()) < (5e-3 if out16 else 2e-5)
    # it is NOT the weight-only product: bf16 activations against the same fp8 weights differ by the activations' e4m3 rounding
    ref_wo = x.double() @ wd.double().t() + bias.double()
    if act == 0 and res is None:
        assert relerr(y, ref_wo.float()) > 1e-3


@pytest.mark.parametrize("H,W,pad", [(360, 640, True), (360, 640, False), (500, 280, True), (500, 280, False), (120, 160, True),
                                      (336, 336, True), (337, 335, False), (1080, 1920, True)])
def test_ingest_frames_bit_exact(nat, H, W, pad):
    """f2: sm_ingest_frames (expand2square + PIL-exact bicubic resize + centre crop, uint8) against the oracle restatement of
    PIL's ImagingResample -- byte-for-byte."""
    rng = np.random.default_rng(H * 7 + W)
    n = 1 if H * W > 1_000_000 else 3
    base = rng.integers(0, 256, (n, H // 8 + 1, W // 8 + 1, 3), dtype=np.uint8).repeat(8, axis=1).repeat(8, axis=2)[:, :H, :W]
    frames = (base.astype(np.int32) // 2 + rng.integers(0, 128, (n, H, W, 3))).astype(np.uint8)
    got = nat.ingest_frames(torch.from_numpy(frames).cuda(), pad_square=pad, image_size=336)
    want = O.ingest_frames(list(frames), "pad" if pad else None, 336)
    assert torch.equal(got.cpu(), want)


def test_ingest_frames_random_sizes_bit_exact(nat):
    """the same on 60 random source sizes (8..1400 a side, both pad modes, aspect ratios up to 1:12): every byte against the oracle's PIL restatement
    (pinned to the reference's process_video by golden g10, tests/test_oracle_golden.py)."""
    rng = np.random.default_rng(2024)
    for case in range(60):
        H, W = int(rng.integers(8, 1401)), int(rng.integers(8, 1401))
        if case % 7 == 0:
            H, W = (int(rng.integers(8, 120)), int(rng.integers(600, 1401))) if case % 2 else (int(rng.integers(600, 1401)), int(rng.integers(8, 120)))
        pad = bool(rng.integers(0, 2))
        n = int(rng.integers(1, 3))
        frames = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        if case % 3 == 0:                                       # smooth content: the rounding of near-constant sums
            frames = (frames // 32 + np.linspace(0, 220, W, dtype=np.float32)[None, None, :, None]).astype(np.uint8)
        got = nat.ingest_frames(torch.from_numpy(frames).cuda(), pad_square=pad, image_size=336)
        want = O.ingest_frames(list(frames), "pad" if pad else None, 336)
        assert torch.equal(got.cpu(), want), (case, H, W, pad, int((got.cpu() != want).sum()))


def test_torch_library_ops_call_the_native_kernels(nat):
    """torch.ops.streammind_hip.* are the same kernels as the ctypes path (bit-identical outputs), incl. the handle-taking ops."""
    import streammind_amd.torch_ops  # noqa: F401
    from tests.util_models import build_native, conn_gate_weights
    ops = torch.ops.streammind_hip
    w = O.bf16_round(rnd((384, 256), 1, 256 ** -0.5)).cuda().bfloat16()
    x = O.bf16_round(rnd((300, 256), 2)).cuda().bfloat16()
    bias = rnd((384,), 3, 0.1).cuda()
    wp = ops.pack_weight(w)
    assert torch.equal(wp, nat.pack_weight(w))
    y = ops.linear(x, wp, 384, 256, bias, 1, None, torch.bfloat16)
    assert torch.equal(y, nat.linear(x, wp, 384, 256, bias=bias, act=1, out_dtype=torch.bfloat16))
    xf = rnd((37, 1024), 4).cuda()
    g, b = (1 + rnd((1024,), 5, 0.1)).cuda(), rnd((1024,), 6, 0.1).cuda()
    ln = ops.norm(xf, g, b, 1e-5, torch.float32)
    assert relerr(ln, O.layer_norm(xf.cpu(), g.cpu(), b.cpu(), 1e-5)) < 1e-5
    assert ops.norm(xf, g, None, 1e-5, torch.float16).dtype == torch.float16
    TVc = O.VitCfg(image_size=56, patch=14, hidden=128, heads=2, mlp=256, layers=3)
    TCc, TGc = O.ConnCfg(mm_hidden=128, d_model=256), O.LmCfg.gate(hidden=256, heads=2, kv_heads=1, mlp=512)
    m = build_native(TVc, TCc, TGc, O.make_vit_weights(TVc, 41), conn_gate_weights(TCc, TGc, 86), max_frames_per_call=4)
    fr = O.synthetic_frames(3, 56, seed=5).cuda()
    assert torch.equal(ops.vit_encode(m.h.value, fr, 128), m.vit_encode(fr))
    s1, s2 = m.open_stream(16, 64), m.open_stream(16, 64)
    import streammind_amd.torch_ops as T
    st = T.new_stream_state("cuda")                      # stands for s1's hidden state: declared mutated, bumped once per call
    lg, dc = ops.stream_push_frames(s1.h.value, fr, st)
    lg2, dc2 = s2.push_frames(fr)
    assert torch.equal(lg, lg2) and torch.equal(dc, dc2) and s1.num_frames == 3 and int(st) == 1
    pooled = ops.pool_rows(torch.randn(3, 16, 128, device="cuda"))
    assert pooled.shape == (3, 128)


@pytest.mark.parametrize("n,pos0", [(100, 37), (64, 0), (31, 5), (2048, 0)])
def test_rope_kv_append_rows_and_tiles(n, pos0):
    """sm_rope_kv_append (HF MistralRotaryEmbedding + apply_rotary_pos_emb, rotate_half convention, then the KV-cache append of
    transformers' DynamicCache.update: the reference reaches both through videollama2_mistral.py:426-431): q rotated to 16-bit rows, k rotated into
    the cache row of ITS position, v into the TRANSPOSED cache.  31 rows take the row kernel, 64 / 100 (ragged, unaligned position) / 2048 the tile
    kernel of round 5 (64 tokens x one head per block, V^T through an LDS transpose): fp32 arithmetic a c - b s / b c + a s with ONE rounding --
    within one 16-bit ulp of the torch statement (fma contraction; absolute 2e-6 where the rotation cancels), v exact, untouched cache rows stay untouched."""
    from streammind_amd._lib import load, check
    lib = load()
    H, KV, dh, S_max = 32, 8, 128, 2304
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(n, (H + 2 * KV) * dh, generator=g)
    half = dh // 2
    inv = 1.0 / (1e6 ** (torch.arange(0, dh, 2).float() / dh))
    ang = torch.arange(S_max).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    q = torch.full((n, H * dh), 7.0, dtype=torch.bfloat16, device="cuda")
    kc = torch.full((S_max, KV * dh), 7.0, dtype=torch.bfloat16, device="cuda")
    vtc = torch.full((KV, dh, S_max), 7.0, dtype=torch.bfloat16, device="cuda")
    qg, cg, sg = qkv.cuda(), cos.cuda(), sin.cuda()
    check(lib.sm_rope_kv_append(qg.data_ptr(), n, pos0, H, KV, dh, cg.data_ptr(), sg.data_ptr(), q.data_ptr(), kc.data_ptr(), vtc.data_ptr(), S_max,
                                torch.cuda.current_stream().cuda_stream))
    x = qkv.view(n, H + 2 * KV, dh)
    c, s_ = cos[pos0:pos0 + n, None, :], sin[pos0:pos0 + n, None, :]
    a, b = x[:, :H + KV, :half], x[:, :H + KV, half:]
    rot = torch.cat([a * c - b * s_, b * c + a * s_], dim=-1)
    want_q = rot[:, :H].reshape(n, H * dh)
    want_k = rot[:, H:].reshape(n, KV * dh)
    for got, want in ((q, want_q), (kc[pos0:pos0 + n], want_k)):
        # one 16-bit ulp (2^-8 relative) -- except where a c - b s cancels: there the fused multiply-add of the kernel and torch's two roundings differ by
        # fp32 epsilon of the OPERANDS (1e-7 absolute on O(1) inputs), many ulps of a result that is itself ~1e-6
        err = (got.float().cpu() - want).abs()
        assert bool((err <= want.abs() * 2.0 ** -7 + 2e-6).all()), float((err - want.abs() * 2.0 ** -7).max())
        assert float((got.cpu() != want.bfloat16()).float().mean()) < 2e-2
    want_v = x[:, H + KV:, :].permute(1, 2, 0).bfloat16()                                # [KV][dh][n]
    assert torch.equal(vtc[:, :, pos0:pos0 + n].cpu(), want_v)
    assert bool((kc[:pos0] == 7).all()) and bool((kc[pos0 + n:] == 7).all())
    assert bool((vtc[:, :, :pos0] == 7).all()) and bool((vtc[:, :, pos0 + n:] == 7).all())


def test_linear_random_shapes_and_options_against_fp64():
    """tools/linear_fuzz.py, 400 cases: rows around every dispatch boundary of sm_linear, N 2..6144, K 32..14336, bf16 / fp32 (plain, hi/lo) / fp16 activations,
    bf16 / fp16 / fp8 weights, bias, activation, residual, 16-bit copy, dual weights, the SwiGLU-dual image, post-LayerNorm / RMSNorm -- every output against an
    fp64 product of the same rounded operands (fp32 outputs 3e-5 of the largest value, 16-bit outputs 2^-7); combinations the operator documents as
    unsupported must be refused with their message, not computed."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "linear_fuzz.py"), "400", "11"], capture_output=True, text=True, timeout=900)
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert r.returncode == 0 and res["mismatches"] == 0 and res["cases"] == 400, (res, r.stderr[-400:])
    assert all(k.startswith("refused: sm_linear failed (-1): sm_linear:") for k in res["refusals"]), res["refusals"]


def test_attention_random_shapes_against_fp32_softmax():
    """tools/attn_fuzz.py, 400 cases: the ViT kernel (batch, 1..600 tokens, heads, head_dim 64 / 128), the causal prefill kernel (1..2048 new queries behind
    0..3000 cached keys, GQA ratios 1..8, sliding windows, paired query tiles) and single-token decode (one-launch and key-split forms, windows) against
    fp32 softmax attention of the same bf16 operands: 8e-3 of the largest value, every output finite."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "attn_fuzz.py"), "400", "13"], capture_output=True, text=True, timeout=900)
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert r.returncode == 0 and res["mismatches"] == 0 and res["cases"] + sum(res["refused_with_reason"].values()) == 400, (res, r.stderr[-400:])


def test_norm_random_shapes_against_fp64():
    """sm_norm_ex on 150 random shapes: 1..3000 rows, widths 4..8192 (the block-per-row form up to 64 rows, the fixed 1024 / 4096 one-wave-per-row forms, the
    general one), row strides beyond the width, LayerNorm / RMSNorm, every activation behind it, fp32 and 16-bit outputs -- against fp64."""
    from streammind_amd._lib import load, check
    lib = load()
    rng = np.random.default_rng(77)
    g = torch.Generator(device="cuda").manual_seed(77)
    acts = {0: lambda t: t, 1: lambda t: t * torch.sigmoid(1.702 * t), 2: lambda t: torch.nn.functional.leaky_relu(t, 0.01), 3: torch.nn.functional.softplus,
            4: torch.nn.functional.silu, 5: torch.nn.functional.gelu}
    for case in range(150):
        M = int(rng.choice([1, 2, 5, 16, 28, 63, 64, 65, 100, 257, 577, 1000, 3000]))
        D = int(rng.choice([4, 8, 64, 100, 128, 256, 1000, 1024, 2048, 4096, 4100, 8192]))
        ldx = D + 4 * int(rng.integers(0, 3))
        ln = bool(rng.integers(0, 2))
        act = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5]))
        x = torch.randn(M, ldx, generator=g, device="cuda") * 3 + 0.5
        gamma, beta = torch.randn(D, generator=g, device="cuda").abs() + 0.3, torch.randn(D, generator=g, device="cuda") * 0.2
        o32 = torch.empty(M, D, device="cuda")
        o16 = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        check(lib.sm_norm_ex(x.data_ptr(), M, D, ldx, gamma.data_ptr(), beta.data_ptr() if ln else None, 1e-5, act, o32.data_ptr(), o16.data_ptr(), D, 0,
                             torch.cuda.current_stream().cuda_stream))
        xd = x[:, :D].double()
        if ln:
            mu = xd.mean(1, keepdim=True)
            ref = (xd - mu) / torch.sqrt(((xd - mu) ** 2).mean(1, keepdim=True) + 1e-5) * gamma.double() + beta.double()
        else:
            ref = xd / torch.sqrt((xd ** 2).mean(1, keepdim=True) + 1e-5) * gamma.double()
        ref = acts[act](ref)
        sc = float(ref.abs().max().clamp_min(1e-6))
        assert float((o32.double() - ref).abs().max()) / sc < 2e-5, (case, M, D, ldx, ln, act)
        assert float((o16.double() - ref).abs().max()) / sc < 2.0 ** -7, (case, M, D, ldx, ln, act)

"""Oracle parity of the 256x256 / 256x128 MFMA GEMM (csrc/gemm256.hip) -- the kernel behind the headline frames/s -- at the
shapes and the batch the bench runs it with (28 frames = 16156 token rows), through the C ABI (sm_linear with tile_hint).

Reference arithmetic: fp64 matmul of the SAME bf16 operands on the host (the oracle's definition of every linear on the
path: bf16 operands, wide accumulate), then bias / quick_gelu / residual in fp64.  Bars as in test_tiled_gemm: 1e-5 relative
for fp32 outputs (fp32 accumulation order), 5e-3 for bf16 outputs (one extra rounding, 2^-8)."""
import pytest
import torch

from oracle import streammind_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

SM_TILE_128, SM_TILE_256, SM_TILE_256x128, SM_TILE_256_ONE = 128, 256, 256128, 2561


@pytest.fixture(scope="module")
def nat():
    from streammind_amd import native, _lib
    _lib.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    torch.set_num_threads(max(1, min(32, torch.get_num_threads() * 2)))
    return native


def rnd(shape, seed, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


def relerr(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-6)).item()


def ref_linear(x, w, bias, act, res):
    """fp64 on the host, row-chunked (M = 16156 x N = 4096 fits easily, the chunks keep the working set in cache)"""
    out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float64)
    wt = w.double().t().contiguous()
    for i in range(0, x.shape[0], 2048):
        t = x[i:i + 2048].double() @ wt
        if bias is not None:
            t += bias.double()
        if act == 1:
            t = O.quick_gelu(t)
        elif act == 2:
            t = O.leaky_relu(t)
        if res is not None:
            t += res[i:i + 2048].double()
        out[i:i + 2048] = t
    return out


VIT_M = 28 * 577          # the bench's batch: 63.1 row tiles of 256


# (name, M, N, K, act, residual, out dtype): the four GEMMs of a CLIP-ViT-L layer at 28 frames + ragged variants
SHAPES = [
    ("qkv", VIT_M, 3072, 1024, 0, False, torch.bfloat16),
    ("out_proj", VIT_M, 1024, 1024, 0, True, torch.float32),
    ("fc1", VIT_M, 4096, 1024, 1, False, torch.bfloat16),
    ("fc2", VIT_M, 1024, 4096, 0, True, torch.float32),
    ("ragged_f32_res", VIT_M + 77, 1000, 1024, 0, True, torch.float32),
    ("ragged_bf16", VIT_M + 77, 1000, 1024, 0, False, torch.bfloat16),
    ("ragged_gelu", VIT_M + 77, 1000, 1024, 1, False, torch.bfloat16),
    ("ragged_n_not_x4_gelu_res", 49000, 250, 4096, 1, True, torch.float32),      # N % 4 != 0: the generic (unvectorised) epilogue
]


@pytest.mark.parametrize("name,M,N,K,act,use_res,out_dtype", SHAPES, ids=[s[0] for s in SHAPES])
def test_gemm256_vit_batch_shapes_vs_fp64(nat, name, M, N, K, act, use_res, out_dtype):
    """automatic dispatch at these sizes IS gemm256_kernel<*, 2> (>= 192 tiles of 256x256); the residual is added IN PLACE
    (residual == output buffer) exactly as vit_body does for out_proj / fc2."""
    w = O.bf16_round(rnd((N, K), 11, K ** -0.5))
    x = O.bf16_round(rnd((M, K), 12))
    bias = rnd((N,), 13, 0.1)
    res = rnd((M, N), 14) if use_res else None
    assert -(-M // 256) * -(-N // 256) >= 192
    wp = nat.pack_weight(w.cuda().bfloat16())
    if use_res:
        buf = res.cuda().clone()
        y = nat.linear(x.cuda().bfloat16(), wp, N, K, bias=bias.cuda(), act=act, residual=buf, out=buf)
    else:
        y = nat.linear(x.cuda().bfloat16(), wp, N, K, bias=bias.cuda(), act=act, out_dtype=out_dtype)
    ref = ref_linear(x, w, bias, act, res)
    assert relerr(y, ref) < (1e-5 if out_dtype == torch.float32 else 5e-3)
    # and the forced 256x128 tile (two blocks per CU) on the same inputs
    if use_res:
        buf = res.cuda().clone()
        y2 = nat.linear(x.cuda().bfloat16(), wp, N, K, bias=bias.cuda(), act=act, residual=buf, out=buf, tile_hint=SM_TILE_256x128)
    else:
        y2 = nat.linear(x.cuda().bfloat16(), wp, N, K, bias=bias.cuda(), act=act, out_dtype=out_dtype, tile_hint=SM_TILE_256x128)
    assert relerr(y2, ref) < (1e-5 if out_dtype == torch.float32 else 5e-3)


@pytest.mark.parametrize("M,N,K,act", [(VIT_M, 3072, 1024, 0), (VIT_M, 4096, 1024, 1), (VIT_M - 300, 2048, 4096, 0), (44 * 577, 3072, 256, 1),
                                       (VIT_M, 1024, 1024, 0)])
@pytest.mark.parametrize("f16", [False, True])
def test_persistent_gemm256_is_bit_identical(nat, M, N, K, act, f16):
    """gemm256p_kernel (one block per CU walking its tiles, the k-loop's DMA pipeline running across the tile seam, epilogue through
    the 32-KiB window) against gemm256_kernel (one tile per block) on the same inputs: same fragment and accumulation order, so the
    bf16 / fp16 outputs must be IDENTICAL -- for 3 / 4 / 8 tiles per CU, a ragged last row tile, a short k-loop (KS = 8) and a
    grid that does not qualify (N = 1024: one round, stays on the one-tile kernel).  Repeated: a race would show as a difference."""
    dt = torch.float16 if f16 else torch.bfloat16
    w = rnd((N, K), 51, K ** -0.5).to(dt)
    x = rnd((M, K), 52).to(dt)
    bias = rnd((N,), 53, 0.1)
    wp = nat.pack_weight(w.cuda())
    xg = x.cuda()
    ref = nat.linear(xg, wp, N, K, bias=bias.cuda(), act=act, out_dtype=dt, tile_hint=SM_TILE_256_ONE)
    for _ in range(3):
        y = nat.linear(xg, wp, N, K, bias=bias.cuda(), act=act, out_dtype=dt, tile_hint=SM_TILE_256)
        assert torch.equal(y, ref)
    assert relerr(ref, ref_linear(x.float(), w.float(), bias, act, None)) < (6e-4 if f16 else 5e-3)


def test_gemm256_patch_embed_remap_vs_fp64(nat):
    """the patch-embedding product of 28 frames: K = 588 zero-padded to 640, rows scattered to token rows b*577 + 1 + p with
    the position embedding (rows 1..576) added as a broadcast residual (remap_in/out/off) -- clip_encoder.py:50 -> HF
    CLIPVisionEmbeddings."""
    B, P, S, D, K, Kp = 28, 576, 577, 1024, 588, 640
    w = torch.zeros(D, Kp)
    w[:, :K] = O.bf16_round(rnd((D, K), 21, K ** -0.5))
    x = torch.zeros(B * P, Kp)
    x[:, :K] = O.bf16_round(rnd((B * P, K), 22))
    pos = rnd((S, D), 23)
    out = torch.full((B * S, D), 7.0, device="cuda")
    nat.linear(x.cuda().bfloat16(), nat.pack_weight(w.cuda().bfloat16()), D, Kp, residual=pos.cuda(), remap=(P, S, 1), out=out)
    ref = (x.double() @ w.double().t()).reshape(B, P, D) + pos[1:].double()
    got = out.cpu().reshape(B, S, D)
    assert (got[:, 0] == 7.0).all()                               # CLS rows are not touched by this product
    assert relerr(got[:, 1:], ref) < 1e-5


@pytest.mark.parametrize("tile", [SM_TILE_256, SM_TILE_256x128, SM_TILE_128])
@pytest.mark.parametrize("M,N,K", [(300, 384, 256), (577, 1024, 1024), (1000, 200, 4096), (257, 130, 96), (4700, 1024, 64)])
@pytest.mark.parametrize("act,out_dtype", [(0, torch.float32), (1, torch.bfloat16), (2, torch.float32), (2, torch.bfloat16)])
def test_gemm_forced_tiles_small_and_ragged(nat, tile, M, N, K, act, out_dtype):
    """every tile kernel forced on small / ragged problems: both wave layouts of gemm256 (256x256: WN = 2, 256x128: WN = 1),
    the runtime-activation instantiation (<-1>: leaky_relu), short K loops (1..3 k-steps: the counted-wait tails), M and N
    tails, and the 128x128 kernel on the same inputs."""
    if tile == SM_TILE_128 and K % 64:
        pytest.skip("the 128x128 kernel needs K padded to 64")
    w = O.bf16_round(rnd((N, K), 1, K ** -0.5))
    x = O.bf16_round(rnd((M, K), 2))
    bias = rnd((N,), 3, 0.1)
    res = rnd((M, N), 4)
    y = nat.linear(x.cuda().bfloat16(), nat.pack_weight(w.cuda().bfloat16()), N, K, bias=bias.cuda(), act=act, residual=res.cuda(),
                   out_dtype=out_dtype, tile_hint=tile)
    ref = ref_linear(x, w, bias, act, res)
    assert relerr(y, ref) < (1e-5 if out_dtype == torch.float32 else 5e-3)


def test_vit_attention_28_frames(nat):
    """vit_attn_kernel at the bench's batch (28 frames x 16 heads, S = 577) against the oracle's mixed-precision statement."""
    from streammind_amd._lib import load, check
    lib = load()
    B, S, H, dh = 28, 577, 16, 64
    D = H * dh
    qkv = O.bf16_round(rnd((B * S, 3 * D), 31))
    qg = qkv.cuda().bfloat16()
    ctx = torch.empty(B * S, D, device="cuda", dtype=torch.bfloat16)
    check(lib.sm_vit_attention(qg.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0, 0, torch.cuda.current_stream().cuda_stream))
    q = qkv[:, :D].reshape(B, S, H, dh).transpose(1, 2)
    k = qkv[:, D:2 * D].reshape(B, S, H, dh).transpose(1, 2)
    v = qkv[:, 2 * D:].reshape(B, S, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    e = torch.exp(s - s.max(-1, keepdim=True).values)
    ref = ((O.bf16_round(e) @ v) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * S, D)
    assert relerr(ctx, ref) < 8e-3


@pytest.mark.parametrize("tile", [0, SM_TILE_256, SM_TILE_256x128, SM_TILE_128])
@pytest.mark.parametrize("M,N,K,act,use_res,out16", [(VIT_M, 3072, 1024, 0, False, True), (VIT_M, 1024, 4096, 0, True, False), (VIT_M, 4096, 1024, 1, False, True),
                                                   (577, 1024, 1024, 1, True, True), (300, 200, 256, 2, True, False), (16, 128, 640, 0, True, False)])
def test_gemm_fp16_operands_vs_fp64(nat, tile, M, N, K, act, use_res, out16):
    """vit_fp16 mode (IEEE half operands, the reference demo's precision -- model/builder.py:54): the same tile kernels built
    with v_mfma_f32_16x16x32_f16.  fp64 reference on the SAME fp16 operands; fp32 outputs 1e-5, fp16 outputs 2^-11 (6e-4)."""
    w = rnd((N, K), 41, K ** -0.5).half()
    x = rnd((M, K), 42).half()
    bias = rnd((N,), 43, 0.1)
    res = rnd((M, N), 44) if use_res else None
    wp = nat.pack_weight(w.cuda())
    assert wp.dtype == torch.float16
    kw = dict(bias=bias.cuda(), act=act, tile_hint=tile)
    if out16:
        y = nat.linear(x.cuda(), wp, N, K, residual=None if res is None else res.cuda(), out_dtype=torch.float16, **kw)
        assert y.dtype == torch.float16
    else:
        y = nat.linear(x.cuda(), wp, N, K, residual=None if res is None else res.cuda(), out_dtype=torch.float32, **kw)
    ref = ref_linear(x.float(), w.float(), bias, act, res)
    assert relerr(y, ref) < (6e-4 if out16 else 1e-5)


def test_vit_attention_fp16_28_frames(nat):
    """vit_attn_kernel<fp16> at the bench's batch against the oracle's mixed statement with fp16 roundings (P rounded to fp16 for
    PV): output fp16, one ulp-ish (1e-3 of max)."""
    from streammind_amd._lib import load, check, SM_OP_F16
    lib = load()
    B, S, H, dh = 28, 577, 16, 64
    D = H * dh
    qkv = rnd((B * S, 3 * D), 31).half()
    qg = qkv.cuda()
    ctx = torch.empty(B * S, D, device="cuda", dtype=torch.float16)
    check(lib.sm_vit_attention(qg.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0, SM_OP_F16, torch.cuda.current_stream().cuda_stream))
    f = qkv.float()
    q = f[:, :D].reshape(B, S, H, dh).transpose(1, 2)
    k = f[:, D:2 * D].reshape(B, S, H, dh).transpose(1, 2)
    v = f[:, 2 * D:].reshape(B, S, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    e = torch.exp(s - s.max(-1, keepdim=True).values)
    ref = ((e.half().float() @ v) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * S, D)
    assert relerr(ctx, ref) < 1e-3


@pytest.mark.parametrize("M,F,K,f16,bias", [(2048, 3072, 1024, False, False), (2304, 4096, 4096, False, False), (2304, 4096, 1024, True, True), (2000, 3584, 512, False, True),
                                            (328, 14336, 4096, False, False), (200, 1024, 512, False, False)])
def test_swiglu_dual_epilogue(M, F, K, f16, bias):
    """SM_ACT_SWIGLU_DUAL (round 5): MistralMLP's act_fn(gate_proj(x)) * up_proj(x) (transformers MistralMLP.forward, the reference's LLM:
    videollama2_mistral.py:426-431 -> HF generate) in the EPILOGUE of the fused gate | up product -- the 256 x 256 kernel stages gate and up row groups
    alternately, so both values of an output element sit in one lane.  One-tile kernel (192 tiles), persistent kernel (288 tiles), ragged last row
    tile, fp16 operands, bias; (200 rows: the 128 x 128 kernel + the SwiGLU pass inside sm_linear).  The fused rows must equal the unfused arithmetic
    (fp32 product -> silu(g) * u -> 16 bits) BIT FOR BIT, and that is the fp64 value to 16-bit rounding."""
    from streammind_amd import native
    from streammind_amd._lib import SM_ACT_SWIGLU_DUAL
    dt = torch.float16 if f16 else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(2 * F, K, generator=g) * K ** -0.5).to(dt)
    x = torch.randn(M, K, generator=g).to(dt)
    b = (torch.randn(2 * F, generator=g) * 0.1) if bias else None
    wp = native.pack_weight(w.cuda())
    xg, bg = x.cuda(), (b.cuda() if bias else None)
    out = torch.empty(M, F, device="cuda", dtype=dt)
    native.linear(xg, wp, 2 * F, K, bias=bg, act=SM_ACT_SWIGLU_DUAL, out=out)
    gu = native.linear(xg, wp, 2 * F, K, bias=bg)                      # the same product as fp32 rows
    want = (torch.nn.functional.silu(gu[:, :F]) * gu[:, F:])
    ref64 = (x.double() @ w.double().t() + (b.double() if bias else 0))
    ref = (torch.nn.functional.silu(ref64[:, :F]) * ref64[:, F:]).float()
    scale = ref.abs().max().item()
    assert (out.float().cpu() - ref).abs().max().item() < (2e-3 if f16 else 1.2e-2) * scale
    # silu on the device is v_exp + v_rcp (1 ulp each), torch's is expf-based: compare against the library's own unfused route bit for bit instead
    unf = torch.empty(M, F, device="cuda", dtype=dt)
    native.linear(xg, wp, 2 * F, K, bias=bg, act=SM_ACT_SWIGLU_DUAL, out=unf, tile_hint=128)
    assert torch.equal(out, unf)
    assert (unf.float() - want).abs().max().item() < (2e-3 if f16 else 1.2e-2) * scale


@pytest.mark.parametrize("M,N,K,norm", [(2048, 4096, 4096, True), (2048, 4096, 14336, True), (1024, 4096, 4096, False), (1500, 2048, 4096, True)])
def test_splitk_slabs_on_the_256_tile(nat, M, N, K, norm):
    """Round 5: 64..128 tiles of 256 x 256 with K >= 4096 and N >= 2048 (the o_proj / down_proj of an LLM prefill chunk: transformers
    MistralAttention.o_proj / MistralMLP.down_proj at 1024..2048 rows) run as split-K slabs on the 256 x 256 kernel + ONE pass that sums the slabs in slab
    order, adds the residual (in place) and -- with post_ln -- writes the RMSNorm of the finished row as the next product's 16-bit operand.  fp32 rows
    1e-5 from fp64 on the same operands; the RMSNorm rows are the oracle's rms_norm of those fp32 rows to 16-bit rounding; tile_hint = 128 (the route
    these shapes took before) agrees to fp32 summation order."""
    w = rnd((N, K), 1, K ** -0.5).bfloat16().float()
    x = rnd((M, K), 2).bfloat16().float()
    res = rnd((M, N), 4)
    g = 1 + rnd((N,), 5, 0.1)
    wp = nat.pack_weight(w.cuda().bfloat16())
    xg, gg = x.cuda().bfloat16(), g.cuda()
    ref = (x.double() @ w.double().t() + res.double()).float()
    outs = []
    for hint in (0, SM_TILE_128):
        resg = res.cuda()
        ln_out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        y = nat.linear(xg, wp, N, K, residual=resg, out=resg, tile_hint=hint, post_ln=(gg, None, 1e-5, ln_out) if norm else None)
        assert ((y.cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-5
        if norm:
            want = O.rms_norm(y.cpu(), g, 1e-5)
            assert ((ln_out.float().cpu() - want).abs().max() / want.abs().max()).item() < 8e-3
        outs.append(y.cpu())
    assert ((outs[0] - outs[1]).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.parametrize("M,N,K,kind", [(33, 4096, 4096, "res_norm"), (64, 6144, 4096, "plain"), (100, 4096, 14336, "res_norm"), (128, 4096, 4096, "res"),
                                        (128, 32000, 4096, "bias"), (70, 4096, 1024, "res_f16"), (128, 28672, 4096, "dual"), (40, 7168, 512, "dual_bias")])
def test_weight_streaming_mfma_33_to_128_rows(nat, M, N, K, kind):
    """Round 5 (csrc/wstream.hip): 33..128 rows of 16-bit activations -- a batched decode step of 33..128 streams (sm_group_llm_decode, M = streams:
    the reference decodes one stream at a time, videollama2_mistral.py:426-431) -- as a weight stream: W fragments straight into per-wave register rings,
    X through LDS by a loader wave, K slabs + the slab pass (with the RMSNorm of the finished row) for narrow products, SwiGLU in the lane for gate | up.
    fp32 rows 1e-5 from fp64 on the same operands; tile_hint = 128 (the 128 x 128 tiled kernel these calls took before) agrees to summation order;
    the RMSNorm rows are the oracle's rms_norm of the fp32 rows; the SwiGLU rows agree with the unfused route to summation order.  By default only the
    WIDE products take this kernel (>= 192 column blocks: gate | up, lm_head -- the two cases of the list with N >= 28672); SM_WSTREAM=2 sends the
    narrow ones through its K-slab form too (measured no faster than the tiled kernel: csrc/linear.hip), which the remaining cases cover when set."""
    from streammind_amd._lib import SM_ACT_SWIGLU_DUAL
    f16 = kind.endswith("f16")
    dt = torch.float16 if f16 else torch.bfloat16
    w = rnd((N, K), 1, K ** -0.5).to(dt).float()
    x = rnd((M, K), 2).to(dt).float()
    wp = nat.pack_weight(w.cuda().to(dt))
    xg = x.cuda().to(dt)
    if kind.startswith("dual"):
        F = N // 2
        b = rnd((N,), 3, 0.1) if kind == "dual_bias" else None
        bg = b.cuda() if b is not None else None
        out, unf = torch.empty(M, F, device="cuda", dtype=dt), torch.empty(M, F, device="cuda", dtype=dt)
        nat.linear(xg, wp, N, K, bias=bg, act=SM_ACT_SWIGLU_DUAL, out=out)
        nat.linear(xg, wp, N, K, bias=bg, act=SM_ACT_SWIGLU_DUAL, out=unf, tile_hint=SM_TILE_128)
        ref64 = x.double() @ w.double().t() + (b.double() if b is not None else 0)
        ref = (torch.nn.functional.silu(ref64[:, :F]) * ref64[:, F:]).float()
        assert (out.float().cpu() - ref).abs().max().item() < 1.2e-2 * ref.abs().max().item()
        # same fp32 accumulators up to summation order -> the 16-bit rows agree to one ulp on a tiny fraction
        d = (out.cpu().view(torch.int16).int() - unf.cpu().view(torch.int16).int()).abs()
        assert (out.float() - unf.float()).abs().max().item() < 1e-2 * ref.abs().max().item() and float((d > 1).float().mean()) < 1e-3
        return
    bias = rnd((N,), 3, 0.1) if kind == "bias" else None
    res = rnd((M, N), 4) if kind.startswith("res") else None
    g = 1 + rnd((N,), 5, 0.1)
    ref = (x.double() @ w.double().t() + (bias.double() if bias is not None else 0) + (res.double() if res is not None else 0)).float()
    outs = []
    for hint in (0, SM_TILE_128):
        resg = res.cuda() if res is not None else None
        ln_out = torch.empty(M, N, device="cuda", dtype=dt) if kind == "res_norm" else None
        y = nat.linear(xg, wp, N, K, bias=bias.cuda() if bias is not None else None, residual=resg, out=resg, tile_hint=hint,
                       post_ln=(g.cuda(), None, 1e-5, ln_out) if ln_out is not None else None)
        assert ((y.cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-5
        if ln_out is not None:
            want = O.rms_norm(y.cpu(), g, 1e-5)
            assert ((ln_out.float().cpu() - want).abs().max() / want.abs().max()).item() < 8e-3
        outs.append(y.cpu())
    assert ((outs[0] - outs[1]).abs().max() / ref.abs().max()).item() < 2e-6


def test_round5_kernels_repeat_bit_for_bit(nat):
    """Race screen of the round-5 kernels: the weight-streaming MFMA kernel (two loader waves hand LDS stages to four consumer waves through raw barriers
    with hand-counted waits), the split-K slabs on the 256 x 256 kernel + the slab pass, and the SwiGLU epilogue of the persistent kernel, each launched 40
    times on the same inputs -- every result must equal the first bit for bit (any difference = a synchronisation bug)."""
    from streammind_amd._lib import SM_ACT_SWIGLU_DUAL
    g = 1 + rnd((4096,), 5, 0.1)
    cases = []
    w = rnd((28672, 4096), 1, 4096 ** -0.5).bfloat16().cuda(); x = rnd((128, 4096), 2).bfloat16().cuda()
    wp = nat.pack_weight(w)
    cases.append(("wstream dual 128 rows", lambda: nat.linear(x, wp, 28672, 4096, act=SM_ACT_SWIGLU_DUAL, out=torch.empty(128, 14336, device="cuda", dtype=torch.bfloat16))))
    x70 = rnd((70, 4096), 3).bfloat16().cuda()
    cases.append(("wstream 70 rows", lambda: nat.linear(x70, wp, 28672, 4096)))
    w2 = rnd((4096, 14336), 4, 14336 ** -0.5).bfloat16().cuda(); x2 = rnd((2048, 14336), 5).bfloat16().cuda(); res = rnd((2048, 4096), 6).cuda()
    wp2 = nat.pack_weight(w2)
    def slabs():
        ln = torch.empty(2048, 4096, device="cuda", dtype=torch.bfloat16)
        y = nat.linear(x2, wp2, 4096, 14336, residual=res, post_ln=(g.cuda(), None, 1e-5, ln))
        return torch.cat([y.view(torch.int32), ln.view(torch.int16).int()], dim=1)
    cases.append(("split-K slabs on the 256 tile + slab / residual / RMSNorm pass", slabs))
    x3 = rnd((2304, 4096), 7).bfloat16().cuda()
    cases.append(("persistent SwiGLU epilogue", lambda: nat.linear(x3, wp, 28672, 4096, act=SM_ACT_SWIGLU_DUAL, out=torch.empty(2304, 14336, device="cuda", dtype=torch.bfloat16))))
    for name, fn in cases:
        first = fn().clone()
        for it in range(40):
            assert torch.equal(fn(), first), (name, it)


@pytest.mark.parametrize("M,f16,persistent", [(VIT_M, False, True), (VIT_M, True, True), (24 * 577, False, False), (VIT_M + 77, False, False)],
                         ids=["28f_bf16_persistent", "28f_fp16_persistent", "24f_bf16_one_tile", "ragged_rows"])
def test_layernorm_folded_into_the_256_tile_products(nat, M, f16, persistent):
    """LayerNorm folding (sm_linear_t.fold_*, round 6): the PRODUCER -- out-proj's shape, fp32 + in-place residual -- leaves the fp32 rows, 16-bit(x * gamma)
    and the per-256-column-tile row sums; the CONSUMER -- fc1's shape (quick_gelu) and q|k|v's (none) -- multiplies the raw rows and applies mean / 1/std on
    its accumulators.  Against fp64 on the host: the producer's three outputs each to their own bar (fp32 rows 1e-5, the 16-bit copy one rounding of the
    GPU's own fp32 row, the sums 1e-5 of their magnitude), and the consumer against LayerNorm-then-linear in fp64 of the rows the GPU left (so only the fold's
    own arithmetic is measured: bf16 output, 5e-3 as every 16-bit-output product here; fp16 5e-4).  Row means are pushed away from zero (+0.7) so that the
    mu * (W gamma) term matters.  28 frames run the persistent consumer, 24 frames (55 row tiles: not whole XCD bands) the one-tile kernel's epilogue."""
    D, F, eps = 1024, 4096, 1e-5
    dt = torch.float16 if f16 else torch.bfloat16
    rd = (lambda t: t.half().float()) if f16 else O.bf16_round
    wo, ctx = rd(rnd((D, D), 21, D ** -0.5)), rd(rnd((M, D), 22))
    bo, x0 = rnd((D,), 23, 0.1), rnd((M, D), 24) + 0.7
    gam, bet = 1.0 + 0.2 * rnd((D,), 25), 0.1 * rnd((D,), 26)
    x = x0.cuda().clone()
    ht = torch.empty(M, D, dtype=dt, device="cuda")
    stats = torch.full((M, D // 256, 2), float("nan"), device="cuda")
    nat.linear(ctx.cuda().to(dt), nat.pack_weight(wo.cuda().to(dt)), D, D, bias=bo.cuda(), residual=x, out=x,
               post_ln=(gam.cuda(), bet.cuda(), eps, ht), fold_out=stats)
    torch.cuda.synchronize()
    want_x = ref_linear(ctx, wo, bo, 0, x0)
    assert relerr(x, want_x) < 1e-5
    xg = x.cpu()                                                        # what the consumer's LayerNorm sees
    assert torch.equal(ht.cpu(), (xg * gam).to(dt))                     # ONE rounding of the GPU's own fp32 row times gamma
    s_ref = torch.stack([xg.double().reshape(M, D // 256, 256).sum(-1), (xg.double() ** 2).reshape(M, D // 256, 256).sum(-1)], dim=-1)
    assert ((stats.cpu().double() - s_ref).abs() / s_ref.abs().clamp_min(1.0)).max().item() < 1e-5
    # consumers
    for N, act, seed in ((F, 1, 31), (3 * D, 0, 32)):
        w = rd(rnd((N, D), seed, D ** -0.5))
        b = rnd((N,), seed + 10, 0.1)
        g_vec = (w.double() @ gam.double()).float().cuda()
        c_vec = (w.double() @ bet.double() + b.double()).float().cuda()
        got = nat.linear(ht, nat.pack_weight(w.cuda().to(dt)), N, D, act=act, out_dtype=dt, fold_in=(stats, g_vec, c_vec, eps),
                         tile_hint=0 if persistent else SM_TILE_256_ONE)
        torch.cuda.synchronize()
        # fp64 of the SAME operands: rstd * (ht @ W^T - mu * (W gamma)) + (W beta + b), mu / rstd from the fp32 rows
        mu = xg.double().mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(xg.double().var(-1, unbiased=False, keepdim=True) + eps)
        want = torch.empty(M, N, dtype=torch.float64)
        wt = w.double().t().contiguous()
        for i in range(0, M, 2048):
            t = rstd[i:i + 2048] * (ht[i:i + 2048].cpu().double() @ wt - mu[i:i + 2048] * g_vec.cpu().double()) + c_vec.cpu().double()
            want[i:i + 2048] = O.quick_gelu(t) if act == 1 else t
        assert relerr(got, want) < (5e-4 if f16 else 5e-3)
        # ... and the fold IS the LayerNorm: the same product from LayerNorm(x) in fp64 differs only by where the one 16-bit rounding sits
        ln = (xg.double() - mu) * rstd * gam.double() + bet.double()
        plain = ref_linear(ln.float(), w, b, act, None)
        assert relerr(want, plain) < (2e-3 if f16 else 1.5e-2)


def test_layernorm_fold_refuses_what_the_256_tile_cannot_do(nat):
    """both sides are SM_EINVAL outside the 256 x 256 tile kernels -- never a silently LayerNorm-less product"""
    from streammind_amd._lib import StreamMindHipError
    M, D = 577, 1024
    w = nat.pack_weight(O.bf16_round(rnd((D, D), 1, D ** -0.5)).cuda().bfloat16())
    xin = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
    x = torch.zeros(M, D, device="cuda")
    ht = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(M, 4, 2, device="cuda")
    one = torch.ones(D, device="cuda")
    with pytest.raises(StreamMindHipError):       # one frame: 12 tiles, the 128 x 128 / split-K path
        nat.linear(xin, w, D, D, residual=x, out=x, post_ln=(one, one, 1e-5, ht), fold_out=stats)
    with pytest.raises(StreamMindHipError):
        nat.linear(xin, w, D, D, out_dtype=torch.bfloat16, fold_in=(stats, one, one, 1e-5))


@pytest.mark.parametrize("frames,N,act", [(8, 4096, 1), (16, 4096, 1), (9, 3072, 0), (12, 4096, 1)], ids=["8f_fc1", "16f_fc1", "9f_qkv", "12f_fc1"])
def test_whole_rounds_plus_remainder_row_split(nat, frames, N, act):
    """round 6: a 16-bit-output product whose 256 x 256 tile count is a little over whole rounds of the chip's CUs (fc1 at 8 frames: 304
    tiles on 256 CUs) is cut by ROWS into the whole rounds and a remainder that runs on the 128 x 128 kernel (linear.hip rowsplit_rows).  Whatever the cut,
    every row is the same product: against fp64 at the 16-bit-output bar, over the whole output (the seam rows included), bias + quick_gelu riding along."""
    M, K = frames * 577, 1024
    w = O.bf16_round(rnd((N, K), 41, K ** -0.5))
    x = O.bf16_round(rnd((M, K), 42))
    bias = rnd((N,), 43, 0.1)
    got = nat.linear(x.cuda().bfloat16(), nat.pack_weight(w.cuda().bfloat16()), N, K, bias=bias.cuda(), act=act, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    want = ref_linear(x, w, bias, act, None)
    assert relerr(got, want) < 5e-3
    seam = 4096 if frames <= 12 else 8192
    assert relerr(got[seam - 4:seam + 4], want[seam - 4:seam + 4]) < 5e-3

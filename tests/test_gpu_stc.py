"""STC connector family on the GPU (builder.py:574-796) through the C ABI: the stc.hip pieces against their torch
definitions, the host class against golden g17 (the reference's own classes at depth 0) and against the oracle (restated timm
RegStage, "parity unpinned") at small and at stock VideoLLaMA2 widths."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from oracle import streammind_oracle as O
from tests.test_oracle_golden import stc_cfg_from_golden

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
Fn = torch.nn.functional


@pytest.fixture(scope="module")
def lib():
    from streammind_amd import _lib
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def rnd(shape, seed, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


def relerr(a, b):
    return ((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("F,H,W,Cn", [(1, 1, 1, 4), (2, 5, 7, 12), (3, 24, 24, 64), (2, 13, 13, 256)])
def test_dwconv3x3_nhwc(lib, F, H, W, Cn):
    from streammind_amd._lib import check
    x, w = rnd((F, H, W, Cn), 1), rnd((Cn, 1, 3, 3), 2, 0.3)
    out = torch.empty(F, H, W, Cn, device="cuda")
    wt, xg = w.reshape(Cn, 9).t().contiguous().cuda(), x.cuda()
    check(lib.sm_dwconv3x3_nhwc(xg.data_ptr(), F, H, W, Cn, wt.data_ptr(), out.data_ptr(), st()), "dwconv")
    ref = Fn.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1, groups=Cn).permute(0, 2, 3, 1).float()
    assert relerr(out, ref) < 1e-6


def test_se_scale_and_add_act(lib):
    from streammind_amd import _lib
    from streammind_amd._lib import check
    F, P, Cn = 3, 10, 24
    x, g = rnd((F * P, Cn), 1), rnd((F, Cn), 2, 2.0)
    o16, o32 = torch.empty(F * P, Cn, dtype=torch.bfloat16, device="cuda"), torch.empty(F * P, Cn, device="cuda")
    xg, gg = x.cuda(), g.cuda()
    check(lib.sm_se_scale(xg.data_ptr(), gg.data_ptr(), F, P, Cn, o16.data_ptr(), o32.data_ptr(), _lib.SM_OP_BF16, st()), "se")
    ref = x * torch.sigmoid(g).repeat_interleave(P, 0)
    assert relerr(o32, ref) < 2e-6 and torch.equal(o16.cpu(), o32.cpu().bfloat16())
    a, b = rnd((7, 20), 3), rnd((7, 20), 4)
    ag, bg = a.cuda(), b.cuda()
    for act, fn in ((_lib.SM_ACT_SILU, Fn.silu), (_lib.SM_ACT_NONE, lambda v: v), (_lib.SM_ACT_GELU, Fn.gelu)):
        p32, p16 = torch.empty(7, 20, device="cuda"), torch.empty(7, 20, dtype=torch.bfloat16, device="cuda")
        check(lib.sm_add_act(ag.data_ptr(), bg.data_ptr(), 140, act, p32.data_ptr(), p16.data_ptr(), _lib.SM_OP_BF16, st()), "add_act")
        assert relerr(p32, fn((a + b).double()).float()) < 2e-6 and torch.equal(p16.cpu(), p32.cpu().bfloat16())
    p16 = torch.empty(7, 20, dtype=torch.bfloat16, device="cuda")
    check(lib.sm_add_act(ag.data_ptr(), None, 140, _lib.SM_ACT_NONE, None, p16.data_ptr(), _lib.SM_OP_BF16, st()), "cast")     # b = NULL: a cast
    assert torch.equal(p16.cpu(), a.bfloat16())


@pytest.mark.parametrize("B,T,H,W,Cn,k,pad", [(1, 4, 4, 4, 8, (2, 2, 2), 1), (2, 5, 6, 6, 16, (2, 2, 2), 1), (1, 3, 5, 5, 16, (1, 2, 2), 1),
                                              (1, 4, 6, 6, 8, (2, 2, 2), 0), (1, 2, 24, 24, 32, (2, 2, 2), 1)])
def test_conv3d_patches_gemm_equals_conv3d(lib, B, T, H, W, Cn, k, pad):
    """the gather + sm_linear pair IS nn.Conv3d(kernel = stride, padding): bf16 operands on both sides, fp64 reference"""
    from streammind_amd import native as nat
    from streammind_amd._lib import check
    x = O.bf16_round(rnd((B, T, H, W, Cn), 1))
    w = O.bf16_round(rnd((24, Cn) + k, 2, (Cn * k[0] * k[1] * k[2]) ** -0.5))
    bias = rnd((24,), 3, 0.1)
    To, Ho, Wo = [(n + 2 * pad - kk) // kk + 1 for n, kk in zip((T, H, W), k)]
    taps = k[0] * k[1] * k[2]
    patches = torch.empty(B * To * Ho * Wo, taps * Cn, dtype=torch.bfloat16, device="cuda")
    xg = x.cuda().bfloat16()
    check(lib.sm_conv3d_patches(xg.data_ptr(), B, T, H, W, Cn, k[0], k[1], k[2], pad, patches.data_ptr(), st()), "patches")
    wp = nat.pack_weight(w.permute(0, 2, 3, 4, 1).reshape(24, -1).contiguous().cuda().bfloat16())
    y = nat.linear(patches, wp, 24, taps * Cn, bias=bias.cuda())
    ref = Fn.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), bias.double(), stride=k, padding=pad).permute(0, 2, 3, 4, 1).reshape(-1, 24).float()
    assert y.shape == ref.shape and relerr(y, ref) < 2e-5


def test_avgpool3d_nhwc(lib):
    from streammind_amd import _lib
    from streammind_amd._lib import check
    B, T, H, W, Cn = 2, 5, 7, 6, 12
    x = rnd((B, T, H, W, Cn), 1)
    xg = x.cuda()
    for k in ((2, 2, 2), (1, 2, 2)):
        To, Ho, Wo = T // k[0], H // k[1], W // k[2]
        o = torch.empty(B, To, Ho, Wo, Cn, device="cuda")
        check(lib.sm_avgpool3d_nhwc(xg.data_ptr(), B, T, H, W, Cn, k[0], k[1], k[2], _lib.SM_ACT_SILU, o.data_ptr(), None, _lib.SM_OP_BF16, st()), "pool")
        ref = Fn.silu(Fn.avg_pool3d(x.permute(0, 4, 1, 2, 3).double(), k)).permute(0, 2, 3, 4, 1).float()
        assert relerr(o, ref) < 2e-6


def _build(cls_name, cfg, W, **kw):
    from types import SimpleNamespace
    from streammind_amd.model import stc_connector as S
    m = getattr(S, cls_name)(SimpleNamespace(mm_hidden_size=cfg.mm_hidden, hidden_size=cfg.hidden), downsample=cfg.downsample, depth=cfg.depth,
                             mlp_depth=cfg.mlp_depth, **kw)
    return m.load_state_dict(W)


def test_stc_family_depth0_vs_reference_golden(gold):
    """golden g17: the reference's STCConnector / SpatialConv / STCConnectorV35 / STPConnector / SpatialPool classes themselves
    (depth 0).  The HIP path multiplies bf16 operands where the reference ran fp32: against the oracle in its bf16-operand mode
    2e-3 of the largest token (fp32 summation order flips the bf16 rounding of a few sampler outputs / GELU outputs by one ulp,
    and at these widths -- K = 64 -- one flipped operand is visible), against the reference's fp32 tokens the bf16 budget 1e-2."""
    g = gold("g17_stc_connector")
    for i in range(int(g["n"])):
        c, cfg = stc_cfg_from_golden(g, i)
        W = O.make_stc_weights(cfg, c["seed"])
        x = torch.from_numpy(g[f"x{i}"])
        out = _build(c["cls"], cfg, W)(x.cuda())
        assert out.dtype == torch.float32 and tuple(out.shape) == tuple(g[f"ref{i}"].shape)
        mixed = O.stc_forward(O.bf16_round(x), W, cfg, O.MIXED)
        print(c["cls"], f"vs bf16-mode oracle {relerr(out, mixed):.2e}, vs reference fp32 {relerr(out, torch.from_numpy(g[f'ref{i}'])):.2e}")
        assert relerr(out, mixed) < 2e-3, (c["cls"], relerr(out, mixed))
        assert relerr(out, torch.from_numpy(g[f"ref{i}"])) < 1e-2, c["cls"]


@pytest.mark.parametrize("cin,c,F,H,W", [(128, 256, 3, 6, 6), (256, 256, 2, 5, 7), (1024, 4096, 1, 24, 24)])
def test_one_bottleneck_vs_oracle(cin, c, F, H, W):
    """ONE restated timm Bottleneck (1x1 -> LN -> SiLU -> depthwise 3x3 -> LN -> SiLU -> squeeze-excite -> 1x1 -> LN, shortcut,
    SiLU), widening (1x1 + LN shortcut) and width-preserving (identity shortcut), up to the stock 1024 -> 4096 at 24 x 24: a
    single block has ONE bf16-rounded intermediate (the squeeze-excite output feeding conv3), so the comparison with the
    oracle's bf16-operand mode is tight -- 1e-3 of the largest output (measured <= 4e-4) -- where the whole connector is not."""
    cfg = O.StcCfg(mm_hidden=cin, hidden=c, depth=1)
    W_ = O.make_stc_weights(cfg, 41)
    m = _build("STCConnector", cfg, W_)
    x = O.bf16_round(rnd((F, H, W, cin), 42))
    x16 = x.reshape(-1, cin).cuda().bfloat16()
    o16, o32 = m._block(m.s1[0], x16, None, F, H, W)
    ref = O.stc_bottleneck(x.permute(0, 3, 1, 2), W_, "s1.b1.", cfg, O.MIXED).permute(0, 2, 3, 1).reshape(-1, c)
    e = relerr(o32, ref)
    print(f"bottleneck {cin}->{c} {F}x{H}x{W}: vs bf16-mode oracle {e:.2e}")
    assert e < 1e-3 and torch.equal(o16.cpu(), o32.cpu().bfloat16())


@pytest.mark.parametrize("cls_name,cfg,shape", [
    ("STCConnector", O.StcCfg(mm_hidden=128, hidden=256, depth=2), (2, 4, 36)),
    ("STCConnector", O.StcCfg(mm_hidden=256, hidden=256, depth=1, mlp_depth=1), (1, 3, 16)),
    ("STPConnector", O.StcCfg(mm_hidden=128, hidden=128, depth=2, sampler="pool"), (1, 4, 64)),
    ("STCConnectorV35", O.StcCfg(mm_hidden=128, hidden=128, depth=1, pad=0), (20, 2, 16)),        # 40 frames: the squeeze-excite products in chunks
])
def test_stc_with_regstage_vs_oracle(cls_name, cfg, shape):
    """the whole connector with its RegStages (timm restated, parity unpinned) against the oracle.  A LayerNorm follows every
    product, which makes the bf16-operand arithmetic of this network chaotic at the level of its own rounding noise: in the
    oracle a 1e-6 relative perturbation of the LayerNorm gains moves the bf16-mode tokens by 5e-3 of the largest token (the fp32
    mode by 4e-6), and bf16 mode vs fp32 mode is 4.5e-3.  Two correct bf16 implementations therefore sit about that far apart;
    the bound is 2.5x the oracle's own bf16-vs-fp32 distance on the same input, against both modes (measured 0.5-1.0x).  The
    tight structural check is test_one_bottleneck_vs_oracle."""
    B, T, L = shape
    W = O.make_stc_weights(cfg, 21)
    x = O.bf16_round(rnd((B, T, L, cfg.mm_hidden), 22))
    out = _build(cls_name, cfg, W)(x.cuda())
    mixed, full = O.stc_forward(x, W, cfg, O.MIXED), O.stc_forward(x, W, cfg, O.FP32)
    assert out.shape == mixed.shape
    e1, e2, noise = relerr(out, mixed), relerr(out, full), relerr(mixed, full)
    print(f"{cls_name} {shape}: vs bf16-mode oracle {e1:.2e}, vs fp32 oracle {e2:.2e}; oracle bf16 vs fp32 {noise:.2e}")
    assert e1 < 2.5 * noise and e2 < 2.5 * noise and noise < 2e-2


def test_stc_stock_videollama2_widths():
    """stock VideoLLaMA2 shape (builder.py:575 defaults): CLIP width 1024 -> 4096, depth 4, two frames of 24 x 24 patch tokens ->
    2 x 13 x 13 = 338 tokens of width 4096, against the oracle (bound as in test_stc_with_regstage_vs_oracle)."""
    cfg = O.StcCfg(mm_hidden=1024, hidden=4096, depth=4)
    W = O.make_stc_weights(cfg, 31)
    x = O.bf16_round(rnd((1, 2, 576, 1024), 32))
    m = _build("STCConnector", cfg, W)
    out = m(x.cuda())
    assert out.shape == (1, 338, 4096) and torch.isfinite(out).all()
    mixed, full = O.stc_forward(x, W, cfg, O.MIXED), O.stc_forward(x, W, cfg, O.FP32)
    e1, e2, noise = relerr(out, mixed), relerr(out, full), relerr(mixed, full)
    print(f"stock widths: vs bf16-mode oracle {e1:.2e}, vs fp32 oracle {e2:.2e}; oracle bf16 vs fp32 {noise:.2e}")
    assert e1 < 2.5 * noise and e2 < 2.5 * noise and noise < 2e-2
    with pytest.raises(NotImplementedError):
        m(x.cuda(), cls_demo=True)


def test_pooled_projectors_vs_reference_golden(gold):
    """`linear` / `mlp{N}x_gelu` (builder.py:121-132 + the frame mean of videollama2_arch.py:293-294) on the C ABI against golden
    g17's reference outputs: the mean over the frames is one sm_pool_rows pass over [b][t][l*d], the MLP is sm_linear with the
    exact-GELU epilogue; bf16 operands against the reference's fp32 -> the bf16 budget, 2e-5 against the oracle's bf16 mode
    for the single-layer case."""
    from types import SimpleNamespace
    from streammind_amd.model import stc_connector as S
    g = gold("g17_stc_connector")
    for i in range(int(g["n_mlp"])):
        c = json.loads(str(g[f"mcfg{i}"]))
        W = O.make_mlp_projector_weights(c["mm_hidden"], c["hidden"], c["depth"], c["seed"], sequential=c["type"] != "linear")
        m = S.build_vision_projector(SimpleNamespace(mm_projector_type=c["type"], mm_hidden_size=c["mm_hidden"], hidden_size=c["hidden"]))
        assert isinstance(m, S.MlpGeluProjector) and m.mlp_depth == c["depth"]
        x = torch.from_numpy(g[f"mx{i}"])
        out = m.load_state_dict(W)(x.cuda())
        ref = torch.from_numpy(g[f"mref{i}"])
        mixed = O.mlp_projector_forward(x, W, c["depth"], c["type"] != "linear", O.MIXED)
        print(c["type"], f"vs bf16-mode oracle {relerr(out, mixed):.2e}, vs reference fp32 {relerr(out, ref):.2e}")
        assert out.shape == ref.shape and relerr(out, ref) < 1e-2 and relerr(out, mixed) < (2e-5 if c["depth"] == 1 else 2e-3)
        assert relerr(m(x.cuda().bfloat16()), O.mlp_projector_forward(O.bf16_round(x), W, c["depth"], c["type"] != "linear", O.MIXED)) < 2e-3


def test_stc_loader_and_dispatch():
    from types import SimpleNamespace
    from streammind_amd.model import stc_connector as S
    cfg = O.StcCfg(mm_hidden=128, hidden=128, depth=1)
    W = O.make_stc_weights(cfg, 5)
    c = SimpleNamespace(mm_hidden_size=128, hidden_size=128, mm_projector_type="stc_connector")
    m = S.build_vision_projector(c, depth=1)
    assert isinstance(m, S.STCConnector)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 16, 128))
    bad = dict(W); bad.pop("s2.b1.se.fc2.bias")
    with pytest.raises(KeyError, match="missing"):
        m.load_state_dict(bad)
    with pytest.raises(KeyError, match="unexpected"):
        m.load_state_dict({**W, "s3.b1.conv1.conv.weight": torch.zeros(1)})
    m.load_state_dict({**W, "cls_net.cls_model.lm_head.weight": torch.zeros(2, 128)})          # the unused classifier rides along in real checkpoints
    with pytest.raises(ValueError, match="square"):
        m(torch.zeros(1, 2, 15, 128))
    for t, k in (("stp_connector", S.STPConnector), ("stc_connector_v35", S.STCConnectorV35), ("spatial_conv", S.SpatialConv), ("spatial_pool", S.SpatialPool)):
        c.mm_projector_type = t
        assert type(S.build_vision_projector(c)) is k
    c.mm_projector_type = "nope"
    with pytest.raises(ValueError, match="Unknown projector type"):
        S.build_vision_projector(c)

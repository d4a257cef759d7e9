"""The STC connector family: the stock VideoLLaMA2 projectors the reference still carries next to its Mamba
connector -- /root/reference/streammind/model/multimodal_projector/builder.py:574-749 (`STCConnector`), :751-758
(`STPConnector`), :760-776 (`STCConnectorV35`), :779-796 (`SpatialConv`, `SpatialPool`), dispatched by
`build_vision_projector` (:139-154) on `config.mm_projector_type`.

Same constructor arguments, same state-dict names (timm's: `s1.b1.conv1.conv.weight`, `.conv1.bn.weight`, `.se.fc1.weight`,
`.downsample.conv.weight`, `sampler.0.weight`, `readout.0.weight` ...), same `forward(x)` result: tokens [b, t'*h'*w', hidden].
Every product runs through the C ABI (`sm_linear`, `sm_norm`, `sm_pool_rows` and the stc.hip pieces); torch holds the buffers.
The tensors stay position-major ("NHWC": the layout the tower emits), so none of the reference's einops rearranges exists here.

The classifier branches of the reference's forward (`cls_training` / `cls_inference` / `cls_demo`, builder.py:655-741) cannot be
executed upstream -- `forward` opens with a live `pdb.set_trace()` and writes to a developer's home directory (:640-653) -- and
the streaming path never reaches them (it uses the Mamba connector); they raise NotImplementedError here.

Parity: the reference's own part (sampler, readout, shapes) is pinned by golden g17; timm's RegStage (timm is absent from the
reference tree and from this image) is checked against a restatement of timm 1.0.x only: "parity unpinned" (DESIGN.md)."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from .. import _lib
from .. import native as nat
from .._lib import check

_ACT = _lib.SM_ACT_SILU


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Block:
    """one timm regnet Bottleneck: 1x1 conv -> LN -> SiLU -> depthwise 3x3 -> LN -> SiLU -> squeeze-excite -> 1x1 conv -> LN,
    plus the (1x1 conv -> LN | identity) shortcut, SiLU on the sum"""

    def __init__(self, sd: Dict[str, torch.Tensor], p: str, cin: int, c: int, dev):
        def f32(k):
            return sd[p + k].detach().to(dev, torch.float32).contiguous()

        def packed(k):
            w = sd[p + k].detach().to(dev, torch.bfloat16)
            return nat.pack_weight(w.reshape(w.shape[0], -1).contiguous())
        self.cin, self.c = cin, c
        self.w1, self.w3 = packed("conv1.conv.weight"), packed("conv3.conv.weight")
        self.n1, self.n2, self.n3 = [(f32(f"conv{i}.bn.weight"), f32(f"conv{i}.bn.bias")) for i in (1, 2, 3)]
        self.dw = f32("conv2.conv.weight").reshape(c, 9).t().contiguous()                 # [9][C] tap-major
        self.rd = sd[p + "se.fc1.weight"].shape[0]
        self.fc1, self.fc1_b = packed("se.fc1.weight"), f32("se.fc1.bias")
        self.fc2, self.fc2_b = packed("se.fc2.weight"), f32("se.fc2.bias")
        self.down = None
        if p + "downsample.conv.weight" in sd:
            self.down = (packed("downsample.conv.weight"), f32("downsample.bn.weight"), f32("downsample.bn.bias"))


class STCConnector:
    """builder.py:574-628.  `load_state_dict` takes the reference module's state dict (any dtype / device; `cls_net.*` entries
    are ignored: the classifier belongs to the branches that are not built)."""

    sampler_kind, sampler_pad = "conv", 1

    def __init__(self, config, downsample: Sequence[int] = (2, 2, 2), depth: int = 4, mlp_depth: int = 2, ln_eps: float = 1e-5,
                 device: str = "cuda:0"):
        self.encoder_hidden_size = int(config.mm_hidden_size)
        self.hidden_size = self.output_hidden_size = int(config.hidden_size)
        self.depth, self.mlp_depth, self.downsample = int(depth), int(mlp_depth), tuple(int(v) for v in downsample)
        self.ln_eps = float(ln_eps)                      # timm LayerNormAct2d default
        self.device = torch.device(device)
        if self.depth == 0 and self.encoder_hidden_size != self.hidden_size:
            raise ValueError("depth 0 leaves the tower width on the sampler's input: mm_hidden_size must equal hidden_size (builder.py:606-617)")
        self._loaded = False
        self.lib = _lib.load()          # fails loudly when the HIP library is missing: there is no other path

    # ---------------------------------------------------------------------------------------------- weights
    def expected_keys(self):
        keys = []
        for stage, cin in (("s1", self.encoder_hidden_size), ("s2", self.hidden_size)):
            for i in range(self.depth):
                p = f"{stage}.b{i + 1}."
                keys += [p + f"conv{j}.{k}" for j in (1, 2, 3) for k in ("conv.weight", "bn.weight", "bn.bias")]
                keys += [p + f"se.fc{j}.{k}" for j in (1, 2) for k in ("weight", "bias")]
                if i == 0 and cin != self.hidden_size:
                    keys += [p + "downsample.conv.weight", p + "downsample.bn.weight", p + "downsample.bn.bias"]
        if self.sampler_kind == "conv":
            keys += ["sampler.0.weight", "sampler.0.bias"]
        keys += [f"readout.{2 * j}.{k}" for j in range(self.mlp_depth) for k in ("weight", "bias")]
        return keys

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self.expected_keys() if k not in sd]
        if missing:
            raise KeyError(f"STC connector: missing tensors {missing[:6]}{' ...' if len(missing) > 6 else ''}")
        unexpected = [k for k in sd if k not in set(self.expected_keys()) and not k.startswith("cls_net.")]
        if strict and unexpected:
            raise KeyError(f"STC connector: unexpected tensors {unexpected[:6]}")
        dev, C = self.device, self.hidden_size
        self.s1 = [_Block(sd, f"s1.b{i + 1}.", self.encoder_hidden_size if i == 0 else C, C, dev) for i in range(self.depth)]
        self.s2 = [_Block(sd, f"s2.b{i + 1}.", C, C, dev) for i in range(self.depth)]
        if self.sampler_kind == "conv":
            w = sd["sampler.0.weight"].detach().to(dev, torch.bfloat16)                 # [O][I][kt][kh][kw] -> [O][kt][kh][kw][I]: tap-major columns
            self.samp_w = nat.pack_weight(w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous())
            self.samp_b = sd["sampler.0.bias"].detach().to(dev, torch.float32).contiguous()
        self.readout = []
        for j in range(self.mlp_depth):
            w = sd[f"readout.{2 * j}.weight"].detach().to(dev, torch.bfloat16).contiguous()
            self.readout.append((nat.pack_weight(w), sd[f"readout.{2 * j}.bias"].detach().to(dev, torch.float32).contiguous()))
        self._loaded = True
        return self

    # ---------------------------------------------------------------------------------------------- pieces
    def _norm(self, x: torch.Tensor, gb, act: int) -> torch.Tensor:
        R, C = x.shape
        out = torch.empty_like(x)
        check(self.lib.sm_norm_ex(x.data_ptr(), R, C, C, gb[0].data_ptr(), gb[1].data_ptr(), self.ln_eps, act, out.data_ptr(), None, C,
                                 _lib.SM_OP_BF16, _st()), "sm_norm")
        return out

    def _se_gate(self, e: torch.Tensor, F: int, P: int, blk: _Block) -> torch.Tensor:
        C = blk.c
        pooled = torch.empty(F, C, dtype=torch.float32, device=e.device)
        check(self.lib.sm_pool_rows(e.data_ptr(), _lib.SM_DT_F32, F, P, C, pooled.data_ptr(), _st()), "sm_pool_rows")
        gate = torch.empty(F, C, dtype=torch.float32, device=e.device)
        for f0 in range(0, F, 16):           # <= 16 rows per call: the weight-streaming kernels with fp32-class (hi / lo) activations
            z = nat.linear(pooled[f0:f0 + 16], blk.fc1, blk.rd, C, bias=blk.fc1_b, act=_ACT, precise=True)
            nat.linear(z, blk.fc2, C, blk.rd, bias=blk.fc2_b, precise=True, out=gate[f0:f0 + 16])
        return gate

    def _block(self, blk: _Block, x16: torch.Tensor, x32: Optional[torch.Tensor], F: int, H: int, W: int):
        """x16 [F*H*W][cin] bf16 (operand of the 1x1 convolutions), x32 the same values in fp32 (identity shortcut); returns the
        block's output as (bf16, fp32)"""
        R, C, P = F * H * W, blk.c, H * W
        a = self._norm(nat.linear(x16, blk.w1, C, blk.cin), blk.n1, _ACT)
        d = torch.empty_like(a)
        check(self.lib.sm_dwconv3x3_nhwc(a.data_ptr(), F, H, W, C, blk.dw.data_ptr(), d.data_ptr(), _st()), "sm_dwconv3x3_nhwc")
        e = self._norm(d, blk.n2, _ACT)
        gate = self._se_gate(e, F, P, blk)
        s16 = torch.empty(R, C, dtype=torch.bfloat16, device=e.device)
        check(self.lib.sm_se_scale(e.data_ptr(), gate.data_ptr(), F, P, C, s16.data_ptr(), None, _lib.SM_OP_BF16, _st()), "sm_se_scale")
        v = self._norm(nat.linear(s16, blk.w3, C, C), blk.n3, _lib.SM_ACT_NONE)
        if blk.down is None and x32 is None:
            x32 = x16.float()            # first block fed by the tower's 16-bit features: the identity shortcut adds exactly those values
        short = x32 if blk.down is None else self._norm(nat.linear(x16, blk.down[0], C, blk.cin), blk.down[1:], _lib.SM_ACT_NONE)
        o32, o16 = torch.empty_like(v), torch.empty(R, C, dtype=torch.bfloat16, device=v.device)
        check(self.lib.sm_add_act(v.data_ptr(), short.data_ptr(), R * C, _ACT, o32.data_ptr(), o16.data_ptr(), _lib.SM_OP_BF16, _st()), "sm_add_act")
        return o16, o32

    def _sample(self, x16: torch.Tensor, x32: Optional[torch.Tensor], B: int, T: int, H: int, W: int):
        """builder.py:606-617 / :757 / :766-776: the (temporal, height, width) downsampler + SiLU; returns (bf16, fp32, T', H', W')"""
        C = self.hidden_size
        kt, kh, kw = self.downsample
        if self.sampler_kind == "pool":
            To, Ho, Wo = T // kt, H // kh, W // kw
            if x32 is None:
                x32 = x16.float()            # depth 0: the tower's 16-bit features are the pool's input (a dtype cast, no arithmetic)
            o32 = torch.empty(B * To * Ho * Wo, C, dtype=torch.float32, device=x16.device)
            o16 = torch.empty(B * To * Ho * Wo, C, dtype=torch.bfloat16, device=x16.device)
            check(self.lib.sm_avgpool3d_nhwc(x32.data_ptr(), B, T, H, W, C, kt, kh, kw, _ACT, o32.data_ptr(), o16.data_ptr(), _lib.SM_OP_BF16, _st()),
                  "sm_avgpool3d_nhwc")
            return o16, o32, To, Ho, Wo
        pad = self.sampler_pad
        To, Ho, Wo = (T + 2 * pad - kt) // kt + 1, (H + 2 * pad - kh) // kh + 1, (W + 2 * pad - kw) // kw + 1
        R2, K2 = B * To * Ho * Wo, kt * kh * kw * C
        patches = torch.empty(R2, K2, dtype=torch.bfloat16, device=x16.device)
        check(self.lib.sm_conv3d_patches(x16.data_ptr(), B, T, H, W, C, kt, kh, kw, pad, patches.data_ptr(), _st()), "sm_conv3d_patches")
        o32 = torch.empty(R2, C, dtype=torch.float32, device=x16.device)
        o16 = torch.empty(R2, C, dtype=torch.bfloat16, device=x16.device)
        nat.linear(patches, self.samp_w, C, K2, bias=self.samp_b, act=_ACT, out=o32, out16=o16)
        return o16, o32, To, Ho, Wo

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, x: torch.Tensor, cls_inference: bool = False, cls_training: bool = False, cls_demo: bool = False,
                frames_features_shape=()) -> torch.Tensor:
        """x: tower features [b, t, l, d] (l a square number of patch tokens) or [b, t, h, w, d] -> tokens [b, t'*h'*w', hidden] fp32"""
        if cls_inference or cls_training or cls_demo:
            raise NotImplementedError("STC classifier branches: not executable in the reference (live pdb.set_trace() and hard-wired torch.save "
                                      "paths, builder.py:640-653); the streaming path uses the Mamba connector")
        if not self._loaded:
            raise RuntimeError("STC connector: load_state_dict() first")
        if x.dim() == 4:
            hw = int(x.shape[2] ** 0.5)
            if hw * hw != x.shape[2]:
                raise ValueError(f"STC connector: {x.shape[2]} patch tokens are not a square grid")
            H = W = hw
        elif x.dim() == 5:
            H, W = int(x.shape[2]), int(x.shape[3])
        else:
            raise ValueError(f"STC connector: expected [b, t, l, d] or [b, t, h, w, d], got {tuple(x.shape)}")
        B, T, D = int(x.shape[0]), int(x.shape[1]), int(x.shape[-1])
        if D != self.encoder_hidden_size:
            raise ValueError(f"STC connector: feature width {D} != mm_hidden_size {self.encoder_hidden_size}")
        x16 = x.to(self.device, torch.bfloat16).reshape(B * T * H * W, D).contiguous()
        x32 = None
        for blk in self.s1:
            x16, x32 = self._block(blk, x16, x32, B * T, H, W)
        x16, x32, To, Ho, Wo = self._sample(x16, x32, B, T, H, W)
        for blk in self.s2:
            x16, x32 = self._block(blk, x16, x32, B * To, Ho, Wo)
        C = self.hidden_size
        for j, (w, b) in enumerate(self.readout):
            last = j == self.mlp_depth - 1
            y = nat.linear(x16, w, C, C, bias=b, act=_lib.SM_ACT_NONE if last else _lib.SM_ACT_GELU,
                           out_dtype=torch.float32 if last else torch.bfloat16)
            x16 = y
        return x16.reshape(B, To * Ho * Wo, C)

    __call__ = forward


class STPConnector(STCConnector):          # builder.py:751-758: AvgPool3d(downsample) + SiLU as the sampler
    sampler_kind = "pool"


class STCConnectorV35(STCConnector):       # builder.py:760-776: the Conv3d without padding
    sampler_pad = 0


class SpatialConv(STCConnector):           # builder.py:779-785
    def __init__(self, config, downsample=(1, 2, 2), depth=0, mlp_depth=2, **kw):
        super().__init__(config, downsample=downsample, depth=depth, mlp_depth=mlp_depth, **kw)


class SpatialPool(STPConnector):           # builder.py:788-794
    def __init__(self, config, downsample=(1, 2, 2), depth=0, mlp_depth=2, **kw):
        super().__init__(config, downsample=downsample, depth=depth, mlp_depth=mlp_depth, **kw)


class MlpGeluProjector:
    """builder.py:121-132: `linear` / `mlp{N}x_gelu` -- nn.Linear(mm_hidden, hidden) [+ (GELU, nn.Linear(hidden, hidden)) x (N - 1)],
    applied by temporal_aggregator to the MEAN OVER THE FRAMES of the tower features (videollama2_arch.py:293-294): [b, t, l, d] ->
    [b, l, hidden].  State-dict names are nn.Sequential's ("0.weight", "2.weight", ...) or nn.Linear's ("weight", "bias")."""

    def __init__(self, config, mlp_depth: int = 1, device: str = "cuda:0"):
        self.encoder_hidden_size, self.hidden_size, self.mlp_depth = int(config.mm_hidden_size), int(config.hidden_size), int(mlp_depth)
        self.sequential = getattr(config, "mm_projector_type", "linear") != "linear"
        self.device = torch.device(device)
        self.lib = _lib.load()
        self._loaded = False

    def expected_keys(self):
        if not self.sequential:
            return ["weight", "bias"]
        return [f"{2 * j}.{k}" for j in range(self.mlp_depth) for k in ("weight", "bias")]

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        keys = self.expected_keys()
        missing = [k for k in keys if k not in sd]
        if missing:
            raise KeyError(f"projector: missing tensors {missing}")
        if strict and [k for k in sd if k not in keys]:
            raise KeyError(f"projector: unexpected tensors {[k for k in sd if k not in keys][:6]}")
        self.layers = []
        for j in range(self.mlp_depth):
            w = sd[keys[2 * j]].detach().to(self.device, torch.bfloat16).contiguous()
            self.layers.append((nat.pack_weight(w), sd[keys[2 * j + 1]].detach().to(self.device, torch.float32).contiguous(), w.shape[1]))
        self._loaded = True
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: tower features [b, t, l, d] -> [b, l, hidden] fp32 (the frame mean is taken here, as temporal_aggregator does)"""
        if not self._loaded:
            raise RuntimeError("projector: load_state_dict() first")
        if x.dim() != 4 or x.shape[-1] != self.encoder_hidden_size:
            raise ValueError(f"projector: expected [b, t, l, {self.encoder_hidden_size}], got {tuple(x.shape)}")
        B, T, L, D = (int(v) for v in x.shape)
        x = x.to(self.device)
        if x.dtype not in (torch.bfloat16, torch.float32, torch.float16):
            x = x.float()
        x = x.contiguous()
        dt = {torch.bfloat16: _lib.SM_DT_BF16, torch.float32: _lib.SM_DT_F32, torch.float16: _lib.SM_DT_F16}[x.dtype]
        mean = torch.empty(B, L * D, dtype=torch.float32, device=self.device)
        check(self.lib.sm_pool_rows(x.data_ptr(), dt, B, T, L * D, mean.data_ptr(), _st()), "sm_pool_rows")     # mean over t of [b][t][l*d]
        h = torch.empty(B * L, D, dtype=torch.bfloat16, device=self.device)
        check(self.lib.sm_add_act(mean.data_ptr(), None, B * L * D, _lib.SM_ACT_NONE, None, h.data_ptr(), _lib.SM_OP_BF16, _st()), "sm_add_act")
        for j, (w, b, k) in enumerate(self.layers):
            last = j == self.mlp_depth - 1
            h = nat.linear(h, w, self.hidden_size, k, bias=b, act=_lib.SM_ACT_NONE if last else _lib.SM_ACT_GELU,
                           out_dtype=torch.float32 if last else torch.bfloat16)
        return h.reshape(B, L, self.hidden_size)

    __call__ = forward


_TYPES = {"stc_connector": STCConnector, "stp_connector": STPConnector, "stc_connector_v35": STCConnectorV35,
          "spatial_conv": SpatialConv, "spatial_pool": SpatialPool}


def build_vision_projector(config, delay_load=False, **kwargs):
    """builder.py:119-158 for the STC family.  `mamba` (the StreamMind connector) lives inside the native model
    (`streammind_amd.model.builder.load_pretrained_model`); `identity` has no reader in temporal_aggregator upstream
    (videollama2_arch.py:293-321 raises for it) and is not built."""
    import re
    t = getattr(config, "mm_projector_type", "linear")
    if t in _TYPES:
        return _TYPES[t](config, **kwargs)
    m = re.match(r"^mlp(\d+)x_gelu$", t)
    if m or t == "linear":
        return MlpGeluProjector(config, mlp_depth=int(m.group(1)) if m else 1, **kwargs)
    if t == "mamba":
        raise ValueError("mm_projector_type 'mamba' is built into the native model: use streammind_amd.model.builder.load_pretrained_model")
    raise ValueError(f"Unknown projector type: {t}")

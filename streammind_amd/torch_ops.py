"""`torch.ops.streammind_hip.*`: the C ABI as PyTorch-ROCm custom ops ("on top: torch.library ops taking / returning
torch.Tensor so that CLIPVisionTower.forward, mm_projector(...), stream_generate_demo keep their Python signatures").

Every op is a thin binding: tensors in, tensors out, the arithmetic is one libstreammind_hip.so call on the current HIP
stream.  Model / stream handles travel as int64 (`NativeModel.h.value`, `NativeStream.h.value`).  Fake (meta) kernels give the
output shapes, so the ops can sit inside torch.compile / export graphs as opaque calls.  Registering needs no GPU; calling
does (the library has no CPU path).

    import streammind_amd.torch_ops                     # registers
    y = torch.ops.streammind_hip.linear(x, wp, N, K, bias, 1, None, torch.bfloat16)
    pooled = torch.ops.streammind_hip.vit_encode(model.h.value, frames_u8)
    st = streammind_amd.torch_ops.new_stream_state("cuda")
    logits, dec = torch.ops.streammind_hip.stream_push_frames(stream.h.value, frames_u8, st)"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check

NS = "streammind_hip"


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _h(handle: int) -> C.c_void_p:
    return C.c_void_p(handle)


def new_stream_state(device) -> Tensor:
    """the tensor that stands for one sm_stream's hidden state in the stateful ops (see the module docstring)"""
    return torch.zeros(1, dtype=torch.int64, device=device)


# ------------------------------------------------------------------------------------------------ operator level
@torch.library.custom_op(f"{NS}::linear", mutates_args=())
def linear(x: Tensor, w_packed: Tensor, N: int, K: int, bias: Optional[Tensor], act: int, residual: Optional[Tensor],
           out_dtype: torch.dtype) -> Tensor:
    """F.linear with the fused epilogue: act(x @ W^T + bias) + residual; W in the packed fragment-major layout"""
    from .native import linear as _linear
    return _linear(x.contiguous(), w_packed, N, K, bias=bias, act=act, residual=residual, out_dtype=out_dtype)


@linear.register_fake
def _(x, w_packed, N, K, bias, act, residual, out_dtype):
    return x.new_empty((x.shape[0], N), dtype=out_dtype)


@torch.library.custom_op(f"{NS}::pack_weight", mutates_args=())
def pack_weight(w: Tensor) -> Tensor:
    from .native import pack_weight as _pack
    return _pack(w.contiguous())


@pack_weight.register_fake
def _(w):
    n = ((w.shape[0] + 15) // 16) * ((w.shape[1] + 31) // 32) * 512
    return w.new_empty((n,))


@torch.library.custom_op(f"{NS}::norm", mutates_args=())
def norm(x: Tensor, gamma: Tensor, beta: Optional[Tensor], eps: float, out_dtype: torch.dtype) -> Tensor:
    """LayerNorm (beta given) / RMSNorm (beta None) over the rows of fp32 x [M, D]; fp32, bf16 or fp16 output"""
    lib = _lib.load()
    x = x.contiguous()
    M, D = x.shape
    out = torch.empty(M, D, dtype=out_dtype, device=x.device)
    of = out.data_ptr() if out_dtype == torch.float32 else None
    o16 = out.data_ptr() if out_dtype != torch.float32 else None
    check(lib.sm_norm_ex(x.data_ptr(), M, D, D, gamma.data_ptr(), None if beta is None else beta.data_ptr(), eps, 0, of, o16, D,
                         _lib.SM_OP_F16 if out_dtype == torch.float16 else _lib.SM_OP_BF16, _st()), "sm_norm")
    return out


@norm.register_fake
def _(x, gamma, beta, eps, out_dtype):
    return x.new_empty(x.shape, dtype=out_dtype)


@torch.library.custom_op(f"{NS}::vit_attention", mutates_args=())
def vit_attention(qkv: Tensor, B: int, S: int, H: int, dh: int) -> Tensor:
    """non-causal MHA over qkv [B*S, 3*H*dh] (bf16 or fp16) -> ctx [B*S, H*dh]"""
    lib = _lib.load()
    qkv = qkv.contiguous()
    ctx = torch.empty(B * S, H * dh, dtype=qkv.dtype, device=qkv.device)
    check(lib.sm_vit_attention(qkv.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0,
                               _lib.SM_OP_F16 if qkv.dtype == torch.float16 else _lib.SM_OP_BF16, _st()), "sm_vit_attention")
    return ctx


@vit_attention.register_fake
def _(qkv, B, S, H, dh):
    return qkv.new_empty((B * S, H * dh))


@torch.library.custom_op(f"{NS}::pool_rows", mutates_args=())
def pool_rows(feats: Tensor) -> Tensor:
    """builder.py:405: mean over the patch axis of [T, P, C] features -> fp32 [T, C]"""
    lib = _lib.load()
    feats = feats.contiguous()
    T, P, Cw = feats.shape
    out = torch.empty(T, Cw, dtype=torch.float32, device=feats.device)
    dt = {torch.bfloat16: _lib.SM_DT_BF16, torch.float32: _lib.SM_DT_F32, torch.float16: _lib.SM_DT_F16}[feats.dtype]
    check(lib.sm_pool_rows(feats.data_ptr(), dt, T, P, Cw, out.data_ptr(), _st()), "sm_pool_rows")
    return out


@pool_rows.register_fake
def _(feats):
    return feats.new_empty((feats.shape[0], feats.shape[2]), dtype=torch.float32)


@torch.library.custom_op(f"{NS}::ingest_frames", mutates_args=())
def ingest_frames(frames: Tensor, pad_square: bool, image_size: int) -> Tensor:
    """u8 frames [n, H, W, 3] of any size -> u8 [n, S, S, 3]: expand2square + PIL-exact bicubic resize + centre crop"""
    from .native import ingest_frames as _ing
    return _ing(frames.contiguous(), pad_square, image_size)


@ingest_frames.register_fake
def _(frames, pad_square, image_size):
    return frames.new_empty((frames.shape[0], image_size, image_size, 3))


# ------------------------------------------------------------------------------------------------ path level (handles as int64)
@torch.library.custom_op(f"{NS}::vit_encode", mutates_args=())
def vit_encode(model: int, frames_u8: Tensor, vit_hidden: int) -> Tensor:
    """CLIPVisionTower + patch mean: u8 frames [B, S, S, 3] -> pooled fp32 [B, vit_hidden]"""
    lib = _lib.load()
    f = frames_u8.contiguous()
    out = torch.empty(f.shape[0], vit_hidden, dtype=torch.float32, device=f.device)
    check(lib.sm_vit_encode(_h(model), f.data_ptr(), f.shape[0], out.data_ptr(), None, None, _st()), "sm_vit_encode")
    return out


@vit_encode.register_fake
def _(model, frames_u8, vit_hidden):
    return frames_u8.new_empty((frames_u8.shape[0], vit_hidden), dtype=torch.float32)


@torch.library.custom_op(f"{NS}::stream_push_frames", mutates_args=("state",))
def stream_push_frames(stream: int, frames_u8: Tensor, state: Tensor) -> Tuple[Tensor, Tensor]:
    """one streaming tick: ViT + connector step + gate for the new frames -> (gate logits [M, 2] fp32, decisions [M] int32)"""
    lib = _lib.load()
    f = frames_u8.contiguous()
    M = f.shape[0]
    lg = torch.empty(M, 2, dtype=torch.float32, device=f.device)
    dc = torch.empty(M, dtype=torch.int32, device=f.device)
    check(lib.sm_stream_push_frames(_h(stream), f.data_ptr(), M, lg.data_ptr(), dc.data_ptr(), _st()), "sm_stream_push_frames")
    state.add_(1)
    return lg, dc


@stream_push_frames.register_fake
def _(stream, frames_u8, state):
    M = frames_u8.shape[0]
    return frames_u8.new_empty((M, 2), dtype=torch.float32), frames_u8.new_empty((M,), dtype=torch.int32)


@torch.library.custom_op(f"{NS}::stream_push_pooled", mutates_args=("state",))
def stream_push_pooled(stream: int, pooled: Tensor, state: Tensor) -> Tuple[Tensor, Tensor]:
    lib = _lib.load()
    p = pooled.contiguous()
    M = p.shape[0]
    lg = torch.empty(M, 2, dtype=torch.float32, device=p.device)
    dc = torch.empty(M, dtype=torch.int32, device=p.device)
    check(lib.sm_stream_push_pooled(_h(stream), p.data_ptr(), M, lg.data_ptr(), dc.data_ptr(), _st()), "sm_stream_push_pooled")
    state.add_(1)
    return lg, dc


@stream_push_pooled.register_fake
def _(stream, pooled, state):
    M = pooled.shape[0]
    return pooled.new_empty((M, 2), dtype=torch.float32), pooled.new_empty((M,), dtype=torch.int32)


@torch.library.custom_op(f"{NS}::llm_prefill", mutates_args=("state",))
def llm_prefill(stream: int, ids: Tensor, state: Tensor) -> Tensor:
    """append the spliced positions (ids >= 0 text, < 0 frame token -(i+1)) to the KV cache; returns the greedy next token [1]"""
    lib = _lib.load()
    ids = ids.to(torch.int32).contiguous()
    check(lib.sm_llm_prefill(_h(stream), ids.data_ptr(), ids.numel(), _st()), "sm_llm_prefill")
    nt = torch.empty(1, dtype=torch.int32, device=ids.device)
    check(lib.sm_stream_read_logits(_h(stream), None, nt.data_ptr(), _st()), "sm_stream_read_logits")
    state.add_(1)
    return nt


@llm_prefill.register_fake
def _(stream, ids, state):
    return ids.new_empty((1,), dtype=torch.int32)


@torch.library.custom_op(f"{NS}::llm_decode", mutates_args=("state",))
def llm_decode(stream: int, n_steps: int, state: Tensor) -> Tensor:
    """n greedy steps continuing the stream"""
    lib = _lib.load()
    out = torch.empty(n_steps, dtype=torch.int32, device=state.device)
    check(lib.sm_llm_decode(_h(stream), n_steps, out.data_ptr(), _st()), "sm_llm_decode")
    state.add_(1)
    return out


@llm_decode.register_fake
def _(stream, n_steps, state):
    return state.new_empty((n_steps,), dtype=torch.int32)


OPS = ("linear", "pack_weight", "norm", "vit_attention", "pool_rows", "ingest_frames", "vit_encode", "stream_push_frames",
       "stream_push_pooled", "llm_prefill", "llm_decode")

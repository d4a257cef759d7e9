"""Torch-facing wrappers over the C ABI: device memory and streams come from PyTorch-ROCm, every
computation is a libstreammind_hip.so call.  No numerical work happens in this file."""
from __future__ import annotations

import os
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Sequence, Dict, Iterable, Optional, Tuple

import torch

from . import _lib
from ._lib import check, sm_config_t, sm_linear_t

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


@dataclass
class PathConfig:
    """Dimensions of the hot path; defaults = CLIP-ViT-L/14-336 + Video_Mamba_seq(4096) + 4-layer gate +
    Mistral-7B.  In a real deployment these come from the checkpoint's config.json files (SURVEY 8 intro)."""
    vit_image: int = 336
    vit_patch: int = 14
    vit_hidden: int = 1024
    vit_heads: int = 16
    vit_mlp: int = 4096
    vit_layers: int = 24
    vit_select_layer: int = -2
    vit_eps: float = 1e-5
    img_mean: Tuple[float, float, float] = CLIP_MEAN
    img_std: Tuple[float, float, float] = CLIP_STD
    conn_d_model: int = 4096
    conn_d_state: int = 16
    conn_d_conv: int = 4
    conn_expand: int = 2
    conn_eps: float = 1e-5
    gate_layers: int = 4
    gate_heads: int = 32
    gate_kv_heads: int = 8
    gate_mlp: int = 14336
    gate_eps: float = 1e-6
    llm_layers: int = 32
    llm_heads: int = 32
    llm_kv_heads: int = 8
    llm_mlp: int = 14336
    llm_vocab: int = 32000
    llm_eps: float = 1e-5
    llm_rope_theta: float = 1e6
    max_frames_per_call: int = 8
    gate_precise: bool = True
    weights_fp8: int = 0           # opt-in BASELINE config 5: 1 = fp8 gate + LLM weights (weight-only: bf16 activations); 2 = the same,
                                   # and calls with > 16 rows run fp8 x fp8 MFMA on per-row-quantised activations (gemm_fp8.hip)
    vit_fp16: bool = False         # vision-tower operands in IEEE fp16 (the reference demo's precision) instead of bf16
    llm_fp16: bool = False         # the same for the LLM (weights, embedding table, activations, q / KV caches, attention P)
    proj_fp16: bool = False        # the same for the connector + gate weights (activations as fp16 hi/lo pairs in precise mode)
    llm_sliding_window: int = 0    # Mistral `sliding_window`: a query at position p attends to keys (p - window, p]; 0 = full causal

    @property
    def vit_layers_run(self) -> int:
        return self.vit_layers + 1 + self.vit_select_layer if self.vit_select_layer < 0 else self.vit_select_layer

    @property
    def n_patches(self) -> int:
        return (self.vit_image // self.vit_patch) ** 2

    @property
    def conn_dt_rank(self) -> int:
        return math.ceil(self.conn_d_model / 16)

    def to_c(self) -> sm_config_t:
        c = sm_config_t()
        c.vit_image, c.vit_patch, c.vit_hidden, c.vit_heads, c.vit_mlp = (
            self.vit_image, self.vit_patch, self.vit_hidden, self.vit_heads, self.vit_mlp)
        c.vit_layers_run, c.vit_eps = self.vit_layers_run, self.vit_eps
        for i in range(3):
            c.img_mean[i] = self.img_mean[i]
            c.img_std[i] = self.img_std[i]
        c.conn_mm_hidden, c.conn_d_model, c.conn_d_state, c.conn_d_conv = (
            self.vit_hidden, self.conn_d_model, self.conn_d_state, self.conn_d_conv)
        c.conn_expand, c.conn_dt_rank, c.conn_eps = self.conn_expand, self.conn_dt_rank, self.conn_eps
        c.gate_hidden, c.gate_layers, c.gate_heads, c.gate_kv_heads, c.gate_mlp, c.gate_eps = (
            self.conn_d_model, self.gate_layers, self.gate_heads, self.gate_kv_heads, self.gate_mlp, self.gate_eps)
        c.llm_hidden, c.llm_layers, c.llm_heads, c.llm_kv_heads, c.llm_mlp, c.llm_vocab = (
            self.conn_d_model, self.llm_layers, self.llm_heads, self.llm_kv_heads, self.llm_mlp, self.llm_vocab)
        c.llm_eps, c.llm_rope_theta = self.llm_eps, self.llm_rope_theta
        c.max_frames_per_call, c.gate_precise = self.max_frames_per_call, int(self.gate_precise)
        c.weights_fp8 = int(self.weights_fp8)
        c.vit_fp16 = int(self.vit_fp16)
        c.llm_fp16 = int(self.llm_fp16)
        c.proj_fp16 = int(self.proj_fp16)
        c.llm_sliding_window = int(self.llm_sliding_window or 0)
        return c


class NativeModel:
    """sm_model: weights + workspaces on one GPU."""

    def __init__(self, cfg: PathConfig, device: str = "cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.StreamMindHipError("no HIP device visible: the StreamMind hot path runs only on a GPU")
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        ccfg = cfg.to_c()
        check(self.lib.sm_model_create(C.byref(ccfg), C.byref(h)), "sm_model_create")
        self.h = h
        self.finalized = False
        self.ignored = []

    def load_tensor(self, name: str, t: torch.Tensor) -> bool:
        """Hand one checkpoint tensor over (any device; bf16/fp16/fp32).  Returns False if the path ignores it."""
        if t.dtype not in (torch.bfloat16, torch.float32, torch.float16):
            t = t.float()
        t = t.to(self.device).contiguous()
        shape = (C.c_int64 * max(1, t.dim()))(*([int(s) for s in t.shape] or [1]))
        dt = {torch.bfloat16: _lib.SM_DT_BF16, torch.float32: _lib.SM_DT_F32, torch.float16: _lib.SM_DT_F16}[t.dtype]
        rc = check(self.lib.sm_model_load_tensor(self.h, name.encode(), t.data_ptr(), dt, max(1, t.dim()), shape, _stream()),
                   f"sm_model_load_tensor({name})")
        torch.cuda.current_stream().synchronize()      # `t` may be a temporary
        if rc == 1:
            self.ignored.append(name)
        return rc == 0

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        for k, v in sd.items():
            self.load_tensor(prefix + k, v)

    def missing(self) -> list:
        buf = C.create_string_buffer(1 << 16)
        self.lib.sm_model_missing(self.h, buf, len(buf))
        return [s for s in buf.value.decode().split("\n") if s]

    def set_fp8_mode(self, mode: int) -> None:
        """weights_fp8 models: 1 = weight-only fp8, 2 = fp8 x fp8 MFMA for calls with more than 16 rows (same weight images)"""
        check(self.lib.sm_model_set_fp8_mode(self.h, int(mode)), "sm_model_set_fp8_mode")

    def finalize(self) -> None:
        check(self.lib.sm_model_finalize(self.h, _stream()), "sm_model_finalize")
        self.finalized = True

    def vit_encode(self, frames_u8: torch.Tensor, return_feats: bool = False, return_pixels: bool = False):
        """frames u8 [B,H,W,3] on the GPU -> pooled fp32 [B, vit_hidden] (+ feats bf16 [B,P,D], pixel_values fp32)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.is_contiguous()
        B, H, W, _ = frames_u8.shape
        if (H, W, frames_u8.shape[3]) != (self.cfg.vit_image, self.cfg.vit_image, 3):
            raise ValueError(f"vit_encode: frames must be [n, {self.cfg.vit_image}, {self.cfg.vit_image}, 3] uint8 (other sizes go through "
                             f"native.ingest_frames first), got {tuple(frames_u8.shape)}")
        pooled = torch.empty(B, self.cfg.vit_hidden, dtype=torch.float32, device=self.device)
        feats = torch.empty(B, self.cfg.n_patches, self.cfg.vit_hidden, dtype=torch.bfloat16, device=self.device) if return_feats else None
        pix = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device) if return_pixels else None
        check(self.lib.sm_vit_encode(self.h, frames_u8.data_ptr(), B, pooled.data_ptr(), _p(feats), _p(pix), _stream()), "sm_vit_encode")
        out = (pooled,)
        if return_feats:
            out += (feats,)
        if return_pixels:
            out += (pix,)
        return out if len(out) > 1 else pooled

    def open_stream(self, max_frames: int = 4096, max_seq: int = 4096) -> "NativeStream":
        return NativeStream(self, max_frames, max_seq)

    def open_group(self, streams: Sequence["NativeStream"]) -> "NativeStreamGroup":
        return NativeStreamGroup(self, streams)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.sm_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeStream:
    """sm_stream: Mamba state, per-frame token store and KV cache of ONE video stream."""

    def __init__(self, model: NativeModel, max_frames: int, max_seq: int):
        self.model, self.lib = model, model.lib
        h = C.c_void_p()
        check(self.lib.sm_stream_open(model.h, max_frames, max_seq, C.byref(h)), "sm_stream_open")
        self.h = h
        self.dev = model.device

    def reset(self) -> None:
        check(self.lib.sm_stream_reset(self.h, _stream()), "sm_stream_reset")

    @property
    def num_frames(self) -> int:
        return self.lib.sm_stream_num_frames(self.h)

    @property
    def kv_len(self) -> int:
        return self.lib.sm_stream_kv_len(self.h)

    def set_kv_len(self, n: int) -> None:
        check(self.lib.sm_stream_set_kv_len(self.h, n), "sm_stream_set_kv_len")

    @property
    def kv_capacity(self) -> int:
        """tokens the K / V cache holds room for right now (grows with the context up to max_seq: sm_stream_kv_capacity)"""
        return self.lib.sm_stream_kv_capacity(self.h)

    def push_pooled(self, pooled: torch.Tensor):
        assert pooled.dtype == torch.float32 and pooled.is_cuda and pooled.is_contiguous()
        M = pooled.shape[0]
        logits = torch.empty(M, 2, dtype=torch.float32, device=self.dev)
        dec = torch.empty(M, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_stream_push_pooled(self.h, pooled.data_ptr(), M, logits.data_ptr(), dec.data_ptr(), _stream()), "sm_stream_push_pooled")
        return logits, dec

    def push_frames(self, frames_u8: torch.Tensor):
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.is_contiguous()
        side = self.model.cfg.vit_image
        if tuple(frames_u8.shape[1:]) != (side, side, 3):      # the C entry takes a bare pointer: a wrong size would be read as garbage
            raise ValueError(f"push_frames: frames must be [n, {side}, {side}, 3] uint8, got {tuple(frames_u8.shape)}")
        M = frames_u8.shape[0]
        logits = torch.empty(M, 2, dtype=torch.float32, device=self.dev)
        dec = torch.empty(M, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_stream_push_frames(self.h, frames_u8.data_ptr(), M, logits.data_ptr(), dec.data_ptr(), _stream()), "sm_stream_push_frames")
        return logits, dec

    def push_frames_pipelined(self, frames_u8: torch.Tensor):
        """push_frames with the connector + gate pass on the stream's side HIP stream (overlaps the next call's tower).  The
        returned tensors are complete after `join()` (or any other call on this stream object)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.is_contiguous()
        side = self.model.cfg.vit_image
        if tuple(frames_u8.shape[1:]) != (side, side, 3):
            raise ValueError(f"push_frames: frames must be [n, {side}, {side}, 3] uint8, got {tuple(frames_u8.shape)}")
        M = frames_u8.shape[0]
        logits = torch.empty(M, 2, dtype=torch.float32, device=self.dev)
        dec = torch.empty(M, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_stream_push_frames_pipelined(self.h, frames_u8.data_ptr(), M, logits.data_ptr(), dec.data_ptr(), _stream()),
              "sm_stream_push_frames_pipelined")
        # the side stream is unknown to torch's caching allocator: keep the outputs (and the frames) alive until joined
        self._inflight = getattr(self, "_inflight", [])[-4:] + [(logits, dec, frames_u8)]
        self.last_ticket = self.lib.sm_stream_pass_ticket(self.h)
        return logits, dec

    def join(self, ticket: Optional[int] = None) -> None:
        """order the CURRENT HIP stream behind the pipelined passes: all of them (ticket None), or only the pass of the call whose
        `last_ticket` is given -- a read-back of call i then does not wait for call i+1 (one batch of look-ahead)"""
        if ticket is None:
            check(self.lib.sm_stream_join(self.h, _stream()), "sm_stream_join")
        else:
            check(self.lib.sm_stream_join_ticket(self.h, ticket, _stream()), "sm_stream_join_ticket")

    def tokens(self, t0: int = 0, n: Optional[int] = None) -> torch.Tensor:
        n = self.num_frames - t0 if n is None else n
        out = torch.empty(n, self.model.cfg.conn_d_model, dtype=torch.float32, device=self.dev)
        check(self.lib.sm_stream_read_tokens(self.h, t0, n, out.data_ptr(), _stream()), "sm_stream_read_tokens")
        return out

    def state(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(conv_state [d_inner, d_conv], ssm_state [d_inner, d_state]) of the connector's recurrence after the frames pushed so far"""
        c = self.model.cfg
        di = c.conn_expand * c.conn_d_model
        conv = torch.empty(di, c.conn_d_conv, dtype=torch.float32, device=self.model.device)
        ssm = torch.empty(di, c.conn_d_state, dtype=torch.float32, device=self.model.device)
        check(self.lib.sm_stream_read_state(self.h, conv.data_ptr(), ssm.data_ptr(), _stream()), "sm_stream_read_state")
        return conv, ssm

    def write_tokens(self, t0: int, toks: torch.Tensor) -> None:
        assert toks.dtype == torch.float32 and toks.is_cuda and toks.is_contiguous()
        check(self.lib.sm_stream_write_tokens(self.h, t0, toks.shape[0], toks.data_ptr(), _stream()), "sm_stream_write_tokens")

    def prefill(self, ids: torch.Tensor) -> None:
        """ids int32 [n] on the GPU: >= 0 text token id, < 0 -> frame token index (-id - 1)."""
        assert ids.dtype == torch.int32 and ids.is_cuda and ids.is_contiguous() and ids.dim() == 1
        check(self.lib.sm_llm_prefill(self.h, ids.data_ptr(), ids.numel(), _stream()), "sm_llm_prefill")

    def forward_logits(self, ids: torch.Tensor) -> torch.Tensor:
        """teacher-forced forward (f1): like prefill, but returns the fp32 logits of EVERY new position [n, vocab]."""
        assert ids.dtype == torch.int32 and ids.is_cuda and ids.is_contiguous() and ids.dim() == 1
        out = torch.empty(ids.numel(), self.model.cfg.llm_vocab, dtype=torch.float32, device=self.dev)
        check(self.lib.sm_llm_forward_logits(self.h, ids.data_ptr(), ids.numel(), out.data_ptr(), _stream()), "sm_llm_forward_logits")
        return out

    def decode(self, n_steps: int) -> torch.Tensor:
        out = torch.empty(n_steps, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_llm_decode(self.h, n_steps, out.data_ptr(), _stream()), "sm_llm_decode")
        return out

    def set_next_token(self, tok: torch.Tensor) -> None:
        """tok: int32 [1] on the GPU -- replaces the pending greedy token (sampling)"""
        assert tok.dtype == torch.int32 and tok.is_cuda and tok.numel() == 1
        check(self.lib.sm_stream_set_next_token(self.h, tok.data_ptr(), _stream()), "sm_stream_set_next_token")

    def logits(self) -> Tuple[torch.Tensor, torch.Tensor]:
        lg = torch.empty(self.model.cfg.llm_vocab, dtype=torch.float32, device=self.dev)
        nt = torch.empty(1, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_stream_read_logits(self.h, lg.data_ptr(), nt.data_ptr(), _stream()), "sm_stream_read_logits")
        return lg, nt

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.sm_stream_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeStreamGroup:
    """sm_stream_group: one tick of S streams = ONE ViT batch + one connector/gate weight pass per <= 32 rows."""

    def __init__(self, model: NativeModel, streams: Sequence[NativeStream]):
        self.model, self.lib, self.streams = model, model.lib, list(streams)
        arr = (C.c_void_p * len(self.streams))(*[s.h for s in self.streams])
        h = C.c_void_p()
        check(self.lib.sm_group_create(arr, len(self.streams), C.byref(h)), "sm_group_create")
        self.h = h
        self.dev = model.device

    def __len__(self) -> int:
        return len(self.streams)

    def push_frames(self, frames_u8: torch.Tensor):
        """frames u8 [S, F, H, W, 3] (or [S, H, W, 3] for one frame per stream) -> (logits [S, F, 2], decisions [S, F])"""
        if frames_u8.dim() == 4:
            frames_u8 = frames_u8.unsqueeze(1)
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.is_contiguous()
        S, F = frames_u8.shape[:2]
        side = self.model.cfg.vit_image
        if S != len(self.streams) or tuple(frames_u8.shape[2:]) != (side, side, 3):
            raise ValueError(f"group push_frames: frames must be [{len(self.streams)}, F, {side}, {side}, 3] uint8, got {tuple(frames_u8.shape)}")
        logits = torch.empty(S, F, 2, dtype=torch.float32, device=self.dev)
        dec = torch.empty(S, F, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_group_push_frames(self.h, frames_u8.data_ptr(), F, logits.data_ptr(), dec.data_ptr(), _stream()), "sm_group_push_frames")
        return logits, dec

    def push_pooled(self, pooled: torch.Tensor):
        """pooled fp32 [S, F, vit_hidden]"""
        assert pooled.dtype == torch.float32 and pooled.is_cuda and pooled.is_contiguous() and pooled.dim() == 3
        S, F = pooled.shape[:2]
        assert S == len(self.streams)
        logits = torch.empty(S, F, 2, dtype=torch.float32, device=self.dev)
        dec = torch.empty(S, F, dtype=torch.int32, device=self.dev)
        check(self.lib.sm_group_push_pooled(self.h, pooled.data_ptr(), F, logits.data_ptr(), dec.data_ptr(), _stream()), "sm_group_push_pooled")
        return logits, dec

    def decode(self, n_steps: int, active: Optional[Sequence[bool]] = None) -> torch.Tensor:
        """batched greedy decode of the (active) streams: int32 [S, n_steps]; rows of inactive streams are -1"""
        S = len(self.streams)
        out = torch.full((S, n_steps), -1, dtype=torch.int32, device=self.dev)
        mask = None if active is None else (C.c_int32 * S)(*[int(bool(a)) for a in active])
        check(self.lib.sm_group_llm_decode(self.h, mask, n_steps, out.data_ptr(), _stream()), "sm_group_llm_decode")
        return out

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.sm_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ operator level
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N,K] bf16 (GPU) -> packed fragment-major image (flat bf16 tensor)."""
    lib = _lib.load()
    assert w.dtype in (torch.bfloat16, torch.float16) and w.is_cuda and w.dim() == 2 and w.is_contiguous()
    N, K = w.shape
    out = torch.empty(lib.sm_packed_elems(N, K), dtype=w.dtype, device=w.device)
    check(lib.sm_pack_weight(w.data_ptr(), N, K, K, out.data_ptr(), _stream()), "sm_pack_weight")
    return out


def pack_weight_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N,K] bf16 (GPU) -> (fp8 packed image as uint8, per-row scales fp32 [N])."""
    lib = _lib.load()
    assert w.dtype == torch.bfloat16 and w.is_cuda and w.dim() == 2 and w.is_contiguous()
    N, K = w.shape
    out = torch.empty(lib.sm_packed_fp8_bytes(N, K), dtype=torch.uint8, device=w.device)
    scale = torch.empty(N, dtype=torch.float32, device=w.device)
    check(lib.sm_quant_pack_weight_fp8(w.data_ptr(), N, K, K, out.data_ptr(), scale.data_ptr(), _stream()), "sm_quant_pack_weight_fp8")
    return out, scale


def ingest_frames(frames: torch.Tensor, pad_square: bool = True, image_size: int = 336,
                  pad_rgb: Sequence[int] = (122, 116, 104)) -> torch.Tensor:
    """f2: u8 frames [n, H, W, 3] of any size on the GPU -> u8 [n, image_size, image_size, 3]: optional expand2square with
    `pad_rgb` (int(CLIP mean * 255)), PIL-exact bicubic shortest-edge resize, centre crop."""
    lib = _lib.load()
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3 and frames.is_contiguous()
    n, H, W, _ = frames.shape
    out = torch.empty(n, image_size, image_size, 3, dtype=torch.uint8, device=frames.device)
    tmp = torch.empty(lib.sm_ingest_tmp_bytes(n, H, W, int(pad_square), image_size), dtype=torch.uint8, device=frames.device)
    rgb = (C.c_uint8 * 3)(*[int(v) for v in pad_rgb])
    check(lib.sm_ingest_frames(frames.data_ptr(), n, H, W, int(pad_square), rgb, image_size, out.data_ptr(), tmp.data_ptr(), _stream()),
          "sm_ingest_frames")
    return out


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> Tuple[torch.Tensor, torch.Tensor]:
    """rows of fp32 logits [n, V] scored against int32 labels [n] (already shifted by the caller):
    -> (nll fp32 [n], 0 where ignored; argmax int32 [n])."""
    lib = _lib.load()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    labels = labels.to(device=logits.device, dtype=torch.int32).contiguous()
    n, V = logits.shape
    assert labels.numel() == n
    nll = torch.empty(n, dtype=torch.float32, device=logits.device)
    am = torch.empty(n, dtype=torch.int32, device=logits.device)
    check(lib.sm_cross_entropy(logits.data_ptr(), n, V, logits.stride(0), labels.data_ptr(), ignore_index, nll.data_ptr(),
                               am.data_ptr(), _stream()), "sm_cross_entropy")
    return nll, am


def set_vit_ln_fold(mode: int) -> None:
    """tower LayerNorm folding, process-wide (sm_set_vit_ln_fold): -1 default (fp16 tower folds, bf16 does not), 0 off, 1 on for both, -2 back to SM_VIT_LN_FOLD"""
    check(_lib.load().sm_set_vit_ln_fold(int(mode)), "sm_set_vit_ln_fold")


def set_vit_frame_lanes(mode: int) -> None:
    """frame lanes of small calls, process-wide (sm_set_vit_frame_lanes): -1 the tile rule (8..10 / 15..20 full-size frames run as two concurrent halves), 1 never, 2..8 forced, -2 environment / default"""
    check(_lib.load().sm_set_vit_frame_lanes(int(mode)), "sm_set_vit_frame_lanes")


def set_prefill_attention_kernel(on: int) -> None:
    """which kernel runs the causal prefill at head_dim 128 (sm_set_prefill_attention_kernel): 1 the prefill kernel, 0 the general tile kernel, -1 environment / default"""
    check(_lib.load().sm_set_prefill_attention_kernel(int(on)), "sm_set_prefill_attention_kernel")


def cosine_rows(x: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """cos(x[t], ref) for every row of fp32 x [T, D] (sm_cosine_rows; torch.nn.functional.cosine_similarity's arithmetic) -> fp32 [T]"""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    ref = ref.to(device=x.device, dtype=torch.float32).contiguous().reshape(-1)
    T, D = x.shape
    assert ref.numel() == D
    out = torch.empty(T, dtype=torch.float32, device=x.device)
    check(lib.sm_cosine_rows(x.data_ptr(), T, D, x.stride(0), ref.data_ptr(), out.data_ptr(), _stream()), "sm_cosine_rows")
    return out


def linear(x: torch.Tensor, wp: torch.Tensor, N: int, K: int, *, w2p: Optional[torch.Tensor] = None,
           bias: Optional[torch.Tensor] = None, act: int = 0, residual: Optional[torch.Tensor] = None,
           out_dtype: torch.dtype = torch.float32, precise: bool = False, w_scale: Optional[torch.Tensor] = None,
           w2_scale: Optional[torch.Tensor] = None, norm_gamma: Optional[torch.Tensor] = None,
           norm_eps: float = 0.0, tile_hint: int = 0, remap: Optional[Tuple[int, int, int]] = None,
           out: Optional[torch.Tensor] = None, fp8_mfma: bool = False, out16: Optional[torch.Tensor] = None,
           post_ln: Optional[Tuple[torch.Tensor, Optional[torch.Tensor], float, torch.Tensor]] = None, post_ln_act: int = 0,
           x_rep: Optional[Tuple[int, int]] = None, fold_out: Optional[torch.Tensor] = None,
           fold_in: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, float]] = None) -> torch.Tensor:
    """Operator-level entry used by the parity tests: y = epilogue(x @ W^T); norm_gamma: RMSNorm of x fused in front.
    LayerNorm folding (sm_linear_t.fold_*): fold_out = stats [M, N / 256, 2] fp32 turns a post_ln call into the PRODUCER (post_ln's `out` receives
    16-bit(y * gamma), stats the per-tile row sums); fold_in = (stats [M, K / 256, 2], g [N], c [N], eps) makes the call the CONSUMER of such rows.
    post_ln = (gamma, beta | None, eps, out [M, N]): LayerNorm (beta given) / RMSNorm of the finished fp32 rows into `out` (16-bit or fp32) from
    the same call (sm_linear_t.post_ln_*), post_ln_act applied to the normalised value; x_rep = (rep, dh): x holds K / rep columns and column
    group j of width dh is read rep times (the gate's repeat_kv in front of o_proj).
    fp8_mfma (with w_scale): SM_W_FP8_MFMA -- above 16 rows the activations are quantised per row and the product is fp8 x fp8."""
    lib = _lib.load()
    assert x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32, torch.float16)
    if x.dtype == torch.float16:       # fp16 operands: `wp` must be the packed image of fp16 weights (pack_weight moves 16-bit words)
        a_f16 = True
        if out is None and out_dtype == torch.bfloat16:
            out_dtype = torch.float16
    else:
        a_f16 = False
    M = x.shape[0]
    a = sm_linear_t()
    a.w, a.w2, a.N, a.K = wp.data_ptr(), _p(w2p), N, K
    a.x, a.x_dtype = x.data_ptr(), (_lib.SM_X_F32 if x.dtype == torch.float32 else _lib.SM_X_BF16)
    a.precise, a.M, a.ldx = int(precise), M, x.shape[1]
    if x_rep is not None:
        a.x_rep, a.x_rep_dh = int(x_rep[0]), int(x_rep[1])
        assert x.shape[1] * a.x_rep == K
    a.bias, a.act = _p(bias), act
    if w_scale is not None:
        a.w_dtype, a.w_scale, a.w2_scale = (_lib.SM_W_FP8_MFMA if fp8_mfma else _lib.SM_W_FP8), w_scale.data_ptr(), _p(w2_scale)
    if residual is not None:
        a.residual, a.ldr = residual.data_ptr(), residual.shape[1]
    if norm_gamma is not None:
        a.norm_gamma, a.norm_eps = norm_gamma.data_ptr(), float(norm_eps)
    a.tile_hint = tile_hint
    a.op_dtype = _lib.SM_OP_F16 if a_f16 else _lib.SM_OP_BF16
    if remap is not None:       # (remap_in, remap_out, remap_off): patch-embed row scatter + broadcast residual rows
        a.remap_in, a.remap_out, a.remap_off = remap
        assert out is not None, "a remapped product writes into a caller-provided [rows][N] buffer"
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    out_dtype = out.dtype
    if out_dtype == torch.float32:
        a.out_f32, a.ldo = out.data_ptr(), N
        if out16 is not None:        # the same result also as a 16-bit copy (operand of the next product) from the same epilogue
            assert out16.shape == out.shape and out16.dtype == (torch.float16 if a_f16 else torch.bfloat16) and out16.is_contiguous()
            a.out_bf16, a.ldo_bf16 = out16.data_ptr(), N
    else:
        assert out16 is None
        a.out_bf16, a.ldo_bf16 = out.data_ptr(), out.shape[1]      # SM_ACT_SWIGLU_DUAL: the caller's [M][N / 2] buffer
    if post_ln is not None:
        g, b, eps, ln_out = post_ln
        assert out_dtype == torch.float32 and ln_out.shape == out.shape and ln_out.is_contiguous()
        a.post_ln_gamma, a.post_ln_beta, a.post_ln_eps, a.post_ln_ldo, a.post_ln_act = g.data_ptr(), _p(b), float(eps), N, int(post_ln_act)
        if ln_out.dtype == torch.float32:
            a.post_ln_out_f32 = ln_out.data_ptr()
        else:
            assert ln_out.dtype == (torch.float16 if a_f16 else torch.bfloat16)
            a.post_ln_out = ln_out.data_ptr()
    if fold_out is not None:
        assert post_ln is not None and fold_out.dtype == torch.float32 and fold_out.is_contiguous() and fold_out.numel() == M * (N // 256) * 2
        a.fold_stats_out = fold_out.data_ptr()
    if fold_in is not None:
        st_, g_, c_, eps_ = fold_in
        assert st_.dtype == torch.float32 and st_.is_contiguous() and st_.numel() == M * (K // 256) * 2 and g_.numel() == N and c_.numel() == N
        a.fold_stats_in, a.fold_g, a.fold_c, a.fold_eps = st_.data_ptr(), g_.data_ptr(), c_.data_ptr(), float(eps_)
    check(lib.sm_linear(C.byref(a), _stream()), "sm_linear")
    return out


class JpegDecoder:
    """Baseline JPEG frames -> RGB u8 [n, H, W, 3] in HBM through the C ABI (include/streammind_hip.h sm_jpeg_*, csrc/jpeg.hip): the
    bit-serial Huffman stage on host threads (ctypes releases the GIL: frames of a batch decode in parallel), one pinned upload of the
    coefficient images, inverse DCT + upsampling + colour conversion on the GPU with libjpeg's integer arithmetic -- byte for byte
    PIL's output.  Frames of one batch share a geometry (a Motion-JPEG clip does).  Raises StreamMindHipError for what the decoder
    does not implement (progressive, CMYK, ...): the caller keeps its host decoder for those."""

    def __init__(self, threads: int = 8, device: Optional[torch.device] = None):
        from concurrent.futures import ThreadPoolExecutor
        self.lib = _lib.load()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads))
        self._pinned = {}            # (n, coef_count) -> (coefs int16 pinned, qt int16 pinned)
        self.host_decode_s = 0.0     # accumulated wall time of the host (entropy) stage
        self.frames = 0
        self.gpu_entropy_frames = 0  # frames whose entropy-coded segment was decoded on the GPU
        self.crosscheck_batches = 0  # SM_JPEG_CROSSCHECK=1: batches cross-checked so far (the checked frame rotates through the batch)
        self.keep_sync_rounds = False  # diagnostic: after a self-synchronising decode, last_sync_rounds = rounds until each frame's states settled (a stream sync)
        self.last_sync_rounds = None

    def info(self, jpeg: bytes) -> "_lib.sm_jpeg_info_t":
        inf = self.lib.sm_jpeg_info.argtypes[2]._type_()
        check(self.lib.sm_jpeg_info(C.cast(C.c_char_p(jpeg), C.c_void_p), len(jpeg), C.byref(inf)), "sm_jpeg_info")
        return inf

    def _decode_gpu_entropy(self, jpegs, inf) -> Optional[torch.Tensor]:
        """Markers on the host (sm_jpeg_scan_prepare: no entropy decode), the files uploaded as they are, the Huffman decode on the GPU: one lane per
        restart interval when the frames carry DRI (sm_jpeg_entropy_decode), self-synchronising lanes over 1024-bit subsequences when they do not
        (sm_jpeg_entropy_decode_sync); then the same reconstruction.  None when a frame does not qualify (per-component scans, a batch that mixes the two
        kinds, ...) or the device reports a malformed segment / states that did not settle: the caller then takes the host path, whose error text names
        the reason."""
        n = len(jpegs)
        tot = sum((len(j) + 15) // 16 * 16 for j in jpegs)
        ssz = C.sizeof(self.lib.sm_jpeg_scan_prepare.argtypes[3]._type_)
        # pinned staging (grown on demand, reused): the files as they are, their offsets, the prepared scans
        st = getattr(self, "_ent_stage", None)
        if st is None or st[0].numel() < tot + 32 or st[1].numel() < n * ssz or st[2].numel() < n:
            st = (torch.empty(max(tot + 32, 1 << 20), dtype=torch.uint8).pin_memory(), torch.empty(max(n, 32) * ssz, dtype=torch.uint8).pin_memory(),
                  torch.empty(max(n, 32), dtype=torch.int32).pin_memory())
            self._ent_stage = st
        blob, sc_host, offs = st
        scans = (self.lib.sm_jpeg_scan_prepare.argtypes[3]._type_ * n).from_address(sc_host.data_ptr())
        pos = 0
        for i, j in enumerate(jpegs):
            if self.lib.sm_jpeg_scan_prepare(C.cast(C.c_char_p(j), C.c_void_p), len(j), C.byref(inf), C.byref(scans[i])) < 0:
                return None
            if (scans[i].restart > 0) != (scans[0].restart > 0):
                return None
            C.memmove(blob.data_ptr() + pos, j, len(j))
            offs[i] = pos
            pos += (len(j) + 15) // 16 * 16
        with torch.cuda.device(self.device):
            bd = blob[:tot + 32].to(self.device, non_blocking=True)
            sd = sc_host[:n * ssz].to(self.device, non_blocking=True)
            od = offs[:n].to(self.device, non_blocking=True)
            cd = torch.empty(n, inf.coef_count, dtype=torch.int16, device=self.device)
            qd = torch.empty(n, 3, 64, dtype=torch.int16, device=self.device)
            status = torch.empty(n, dtype=torch.int32, device=self.device)
            if scans[0].restart > 0:
                check(self.lib.sm_jpeg_entropy_decode(bd.data_ptr(), tot, od.data_ptr(), sd.data_ptr(), C.byref(inf), n, cd.data_ptr(), qd.data_ptr(), status.data_ptr(),
                                                      _stream()), "sm_jpeg_entropy_decode")
            else:
                check(self.lib.sm_jpeg_entropy_decode_sync(bd.data_ptr(), tot, max(len(j) for j in jpegs), od.data_ptr(), sd.data_ptr(), C.byref(inf), n, cd.data_ptr(),
                                                           qd.data_ptr(), status.data_ptr(), _stream()), "sm_jpeg_entropy_decode_sync")
                if self.keep_sync_rounds:
                    r = (C.c_int32 * n)()
                    check(self.lib.sm_jpeg_sync_rounds(_stream(), r, n), "sm_jpeg_sync_rounds")
                    self.last_sync_rounds = list(r)
            planes = torch.empty(self.lib.sm_jpeg_planes_bytes(C.byref(inf), n), dtype=torch.uint8, device=self.device)
            rgb = torch.empty(n, inf.height, inf.width, 3, dtype=torch.uint8, device=self.device)
            check(self.lib.sm_jpeg_reconstruct(cd.data_ptr(), qd.data_ptr(), C.byref(inf), n, planes.data_ptr(), rgb.data_ptr(), _stream()), "sm_jpeg_reconstruct")
            self.last_entropy_status = status.cpu()          # (also the sync that lets the pinned staging buffers be reused)
            if int(self.last_entropy_status.abs().max()) != 0:
                return None
        self.gpu_entropy_frames += n
        return rgb

    def decode(self, jpegs: Sequence[bytes], entropy: str = "auto") -> torch.Tensor:
        """entropy: "auto" = on the GPU when the batch qualifies (see _decode_gpu_entropy), else on host threads; "host" / "gpu" force one
        ("gpu" raises when a frame does not qualify)."""
        import time
        n = len(jpegs)
        assert n >= 1 and entropy in ("auto", "host", "gpu")
        jpegs = [bytes(j) for j in jpegs]
        inf = self.info(jpegs[0])
        if entropy != "host":
            rgb = self._decode_gpu_entropy(jpegs, inf)
            if rgb is not None:
                if os.environ.get("SM_JPEG_CROSSCHECK") == "1":
                    # soak mode (round-5 advisor): a wrong-but-status-0 device decode would not fall back by itself -- with this set, ONE frame of every batch is
                    # decoded again by the host Huffman path and compared byte for byte; a difference raises instead of travelling on
                    k = self.crosscheck_batches % n
                    self.crosscheck_batches += 1
                    ref = self.decode([jpegs[k]], entropy="host")[0]
                    if not torch.equal(ref, rgb[k]):
                        raise _lib.StreamMindHipError(f"SM_JPEG_CROSSCHECK: frame {k} of a {n}-frame batch decodes differently on the device and on the host")
                    self.frames -= 1                    # (the cross-check frame is not a delivered frame)
                self.frames += n
                return rgb
            if entropy == "gpu":
                sc = self.lib.sm_jpeg_scan_prepare.argtypes[3]._type_()
                for i, j in enumerate(jpegs):
                    check(self.lib.sm_jpeg_scan_prepare(C.cast(C.c_char_p(j), C.c_void_p), len(j), C.byref(inf), C.byref(sc)), f"sm_jpeg_scan_prepare(frame {i})")
                raise _lib.StreamMindHipError("GPU entropy decode: the batch mixes frames with and without restart intervals, or the device reported a malformed "
                                              f"entropy-coded segment / decoder states that did not settle (status {getattr(self, 'last_entropy_status', None)})")
        key = (n, inf.coef_count)
        if key not in self._pinned:
            if len(self._pinned) > 4:
                self._pinned.clear()
            self._pinned[key] = (torch.empty(n, inf.coef_count, dtype=torch.int16).pin_memory(), torch.empty(n, 3, 64, dtype=torch.int16).pin_memory())
        coefs, qt = self._pinned[key]
        cp, qp = coefs.data_ptr(), qt.data_ptr()

        def one(i):
            return self.lib.sm_jpeg_decode_coefs(C.cast(C.c_char_p(jpegs[i]), C.c_void_p), len(jpegs[i]), C.byref(inf), cp + i * inf.coef_count * 2, qp + i * 3 * 64 * 2)

        t0 = time.perf_counter()
        rcs = list(self.pool.map(one, range(n))) if n > 1 else [one(0)]
        self.host_decode_s += time.perf_counter() - t0
        self.frames += n
        for i, rc in enumerate(rcs):
            if rc < 0:        # thread-local error text lives on the worker: decode that frame again here for the message
                check(one(i), f"sm_jpeg_decode_coefs(frame {i})")
        with torch.cuda.device(self.device):
            cd = coefs.to(self.device, non_blocking=True)
            qd = qt.to(self.device, non_blocking=True)
            planes = torch.empty(self.lib.sm_jpeg_planes_bytes(C.byref(inf), n), dtype=torch.uint8, device=self.device)
            rgb = torch.empty(n, inf.height, inf.width, 3, dtype=torch.uint8, device=self.device)
            check(self.lib.sm_jpeg_reconstruct(cd.data_ptr(), qd.data_ptr(), C.byref(inf), n, planes.data_ptr(), rgb.data_ptr(), _stream()), "sm_jpeg_reconstruct")
            for t in (cd, qd, planes):
                t.record_stream(torch.cuda.current_stream(self.device))
        # the pinned buffers are reused by the next call: the upload has to be done before they are overwritten
        torch.cuda.current_stream(self.device).synchronize()
        return rgb

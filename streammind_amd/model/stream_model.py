from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import check
from ..constants import IGNORE_INDEX, MMODAL_TOKEN_INDEX
from .. import native as _native
from ..native import NativeModel, NativeStream, PathConfig, _stream

_DT = {torch.bfloat16: _lib.SM_DT_BF16, torch.float32: _lib.SM_DT_F32, torch.float16: _lib.SM_DT_F16}


class CLIPVisionTower:
    """drop-in for clip_encoder.py:7-84: images [N,3,H,W] (normalised pixel_values, any float dtype) or a list of [3,H,W]
    -> hidden_states[select_layer][:, 1:] as [N, num_patches, hidden] in the input dtype."""

    def __init__(self, vision_tower, args=None, delay_load=False, select_feature: Optional[str] = None):
        """clip_encoder.py:9-29 signature.  `vision_tower`: the tower checkpoint -- a local CLIP directory (config.json +
        weights; the reference passes the same string to CLIPVisionModel.from_pretrained, hub ids need a network) -- or, inside
        this build, the NativeModel that already holds the tower.  `args`: the config namespace the reference reads
        mm_vision_select_layer / mm_vision_select_feature from (:15-16).  delay_load=True defers reading the directory to
        load_model() as upstream (:18-22)."""
        self.vision_tower_name = vision_tower if isinstance(vision_tower, str) else None
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = select_feature or getattr(args, "mm_vision_select_feature", "patch")
        self._fp16 = bool(getattr(args, "vit_fp16", False))
        self._max_frames = int(getattr(args, "max_frames_per_call", 8))
        self.is_loaded = False
        self.native = None
        if isinstance(vision_tower, NativeModel):
            self.native = vision_tower
            self.select_layer = vision_tower.cfg.vit_select_layer
            self.is_loaded = True
        elif not delay_load:
            self.load_model()

    def load_model(self):
        """clip_encoder.py:24-29: read the tower's own checkpoint directory into a tower-only native model (no connector, no gate,
        no LLM: sm_config_t.conn_d_state = gate_layers = llm_layers = 0)."""
        import json
        import os
        from .builder import _TOWER_ROOTS, _checkpoint_tensors
        d = self.vision_tower_name
        if not d or not os.path.isdir(d):
            raise FileNotFoundError(f"vision tower {d!r} is not a local directory (hub ids cannot be resolved here)")
        vj = json.load(open(os.path.join(d, "config.json")))
        vj = vj.get("vision_config", vj)
        over = {}
        pp = os.path.join(d, "preprocessor_config.json")
        if os.path.exists(pp):
            pj = json.load(open(pp))
            if pj.get("image_mean") and pj.get("image_std"):
                over.update(img_mean=tuple(pj["image_mean"]), img_std=tuple(pj["image_std"]))
        cfg = PathConfig(vit_image=vj["image_size"], vit_patch=vj["patch_size"], vit_hidden=vj["hidden_size"],
                         vit_heads=vj["num_attention_heads"], vit_mlp=vj["intermediate_size"], vit_layers=vj["num_hidden_layers"],
                         vit_select_layer=self.select_layer, vit_eps=vj.get("layer_norm_eps", 1e-5), conn_d_state=0, gate_layers=0,
                         llm_layers=0, max_frames_per_call=self._max_frames, vit_fp16=self._fp16, **over)
        nat = NativeModel(cfg)
        for k, v in _checkpoint_tensors(d):
            if k.startswith("vision_model."):
                nat.load_tensor("model.vision_tower.vision_tower." + k, v)
            elif k.startswith(_TOWER_ROOTS):
                nat.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v)
        if nat.missing():
            raise ValueError(f"vision tower checkpoint incomplete, missing: {nat.missing()[:8]}")
        nat.finalize()
        self.native, self.is_loaded = nat, True

    def feature_select(self, image_forward_outs) -> torch.Tensor:
        """clip_encoder.py:31-39.  The native tower emits hidden_states[select_layer] with the CLS row already dropped, so the
        argument here is that tensor; an HF-style output object (with .hidden_states) is sliced as upstream."""
        feats = image_forward_outs
        if hasattr(image_forward_outs, "hidden_states"):
            feats = image_forward_outs.hidden_states[self.select_layer][:, 1:]
        if self.select_feature == "patch":
            return feats
        raise ValueError(f"Unexpected select feature: {self.select_feature}")   # clip_encoder.py:38 ('cls_patch' needs the CLS row)

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:
            return [self.forward(im.unsqueeze(0)) for im in images]
        cfg, nat = self.native.cfg, self.native
        x = images.to(self.device)
        if x.dtype not in _DT:
            x = x.float()
        x = x.contiguous()
        N = x.shape[0]
        assert x.dim() == 4 and x.shape[1] == 3 and x.shape[2] == cfg.vit_image and x.shape[3] == cfg.vit_image
        out = torch.empty(N, cfg.n_patches, cfg.vit_hidden, dtype=torch.bfloat16, device=self.device)
        pooled = torch.empty(N, cfg.vit_hidden, dtype=torch.float32, device=self.device)
        B = cfg.max_frames_per_call
        for i in range(0, N, B):
            n = min(B, N - i)
            check(nat.lib.sm_vit_encode_pixels(nat.h, x[i:i + n].data_ptr(), _DT[x.dtype], n, pooled[i:i + n].data_ptr(),
                                              out[i:i + n].data_ptr(), _stream()), "sm_vit_encode_pixels")
        return self.feature_select(out).to(images.dtype)

    __call__ = forward

    @property
    def dtype(self):
        return torch.bfloat16

    @property
    def device(self):
        return self.native.device

    @property
    def config(self):
        c = self.native.cfg
        return SimpleNamespace(hidden_size=c.vit_hidden, image_size=c.vit_image, patch_size=c.vit_patch,
                               num_hidden_layers=c.vit_layers, num_attention_heads=c.vit_heads, intermediate_size=c.vit_mlp)

    @property
    def hidden_size(self):
        return self.native.cfg.vit_hidden

    @property
    def num_patches(self):
        return self.native.cfg.n_patches

    @property
    def num_patches_per_side(self):
        return self.native.cfg.vit_image // self.native.cfg.vit_patch

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)


class Video_Mamba_seq:
    """drop-in for the inference branches of builder.py:403-414,547-564: frames_features [1,t,P,C] -> tokens [1,t,d]
    (and, with cls_demo=True, the gate logits of the LAST frame, fp32 [2]).  Like the reference it recomputes all t
    frames on every call; the O(1)-per-frame form is StreamingSession / stream_generate_demo."""

    def __init__(self, native: NativeModel):
        self.native = native

    @torch.no_grad()
    def forward(self, x: torch.Tensor, cls_inference=False, cls_training=False, cls_demo=False, frames_features_shape=[],
                prompt_time_input_ids=None, prompt_time_lable=None):
        if (cls_inference or cls_training) and prompt_time_input_ids is not None and prompt_time_input_ids.numel() > 1:
            # builder.py:424-494: needs gate token ids 32000/32001 that its own 2-word ClsNet does not have
            raise NotImplementedError("the prompt-conditioned gate branch is debug-broken in the reference (builder.py:422-494)")
        b, t, l, d = x.shape
        assert b == 1, "only support batch size 1"          # eval/inference_video_score_stream_ddp.py:325
        nat = self.native
        x = x.to(nat.device)
        if x.dtype not in _DT:
            x = x.float()
        x = x.contiguous()
        pooled = torch.empty(t, d, dtype=torch.float32, device=nat.device)
        check(nat.lib.sm_pool_rows(x.data_ptr(), _DT[x.dtype], t, l, d, pooled.data_ptr(), _stream()), "sm_pool_rows")
        s = nat.open_stream(max_frames=t, max_seq=64)
        logits = None
        cap = 32
        all_logits = []
        for i in range(0, t, cap):
            logits, _ = s.push_pooled(pooled[i:i + cap].contiguous())
            all_logits.append(logits)
        self._all_logits = torch.cat(all_logits)
        tokens = s.tokens().unsqueeze(0)
        s.close()
        if cls_inference or cls_training:
            out, lab = gate_eval_outputs(self._all_logits, frames_features_shape)
            return out if cls_training else (out, lab)
        if cls_demo:
            return tokens, logits[-1]
        return tokens

    __call__ = forward


GATE_CLASS_WEIGHT = (0.15, 0.85)                                   # CrossEntropyLoss(weight=...) of builder.py:345-349


def gate_eval_outputs(gate_logits: torch.Tensor, frames_features_shape: Sequence[int]):
    """builder.py:496-545 (prompt-free branch): per frame the 2-token sequence [frame token, embed(target)], labels
    [IGNORE, target], target 1 for the last frame of a clip else 0, at most 4000 frames; the class-weighted shifted CE.
    -> (SimpleNamespace(logits [T,2,2], loss), labels [T,2]).  Only position 0 is ever scored (its label sits at
    position 1); it is the streaming gate's logit pair, computed by the same HIP step.  Position 1 of `logits` would need
    the 2-token attention the gate path never runs: it is returned as NaN."""
    T = gate_logits.shape[0]
    starts = [0] + list(frames_features_shape[:-1])
    tgt = torch.zeros(T, dtype=torch.int32)
    for k, end in enumerate(frames_features_shape):
        if end > starts[k]:
            tgt[end - 1] = 1
    n = min(T, 4000)
    lg, tgt = gate_logits[:n], tgt[:n]
    nll, _ = _native.cross_entropy(lg.contiguous(), tgt)
    w = torch.tensor(GATE_CLASS_WEIGHT, device=nll.device)[tgt.to(nll.device).long()]
    loss = (nll * w).sum() / w.sum()
    logits = torch.full((n, 2, 2), float("nan"), dtype=torch.float32, device=lg.device)
    logits[:, 0] = lg
    labels = torch.stack([torch.full((n,), IGNORE_INDEX, dtype=torch.long), tgt.long()], dim=1)
    return SimpleNamespace(logits=logits, loss=loss), labels


def exponential_sampling_indices(n: int, percentage: float = 0.6) -> List[int]:
    """videollama2_arch.py:595-601 ("log" sampling; the live code is linear spacing)."""
    num = 1 if int(percentage * n) == 0 else int(percentage * n)
    return torch.linspace(0, n - 1, num).int().tolist()


def similarity_sampling_indices(tokens: torch.Tensor, percentage: float = 0.6) -> List[int]:
    """videollama2_arch.py:603-611: the top `percentage` frames by cosine similarity to the last one, in time order."""
    # the T similarities are a library call (sm_cosine_rows); ranking T numbers is host bookkeeping
    sim = _native.cosine_rows(tokens.float().contiguous(), tokens[-1])
    order = torch.argsort(sim, descending=True)
    return sorted(order[:max(int(percentage * len(order)), 1)].tolist())


class Videollama2MistralForCausalLM:
    """The streaming surface of language_model/videollama2_mistral.py:146-449.  Per-stream state lives on the object, as in
    the reference (frame history -> here the native sm_stream, interval_id_list): one stream per instance, not re-entrant."""

    def __init__(self, native: NativeModel, max_frames: int = 4096, max_seq: int = 4096, eos_token_id: Optional[int] = 2):
        self.native = native
        self.config = native.cfg                       # load_pretrained_model replaces it by the checkpoint's config.json namespace
        self.vision_tower = CLIPVisionTower(native)
        # conn_d_state == 0: a stock VideoLLaMA2 checkpoint -- load_pretrained_model puts the STC-family connector here
        self.mm_projector = Video_Mamba_seq(native) if native.cfg.conn_d_state > 0 else None
        self.stream: NativeStream = native.open_stream(max_frames=max_frames, max_seq=max_seq)
        self.max_seq = max_seq
        self.interval_id_list: List[int] = []          # videollama2_mistral.py:162 (never reset by the demo loop)
        self.eos_token_id = eos_token_id
        self._kv_ids: List[int] = []                   # ids whose K/V currently sit in the cache (prefix reuse)
        self.decode_chunk = 16
        self.last_gate_logits: Optional[torch.Tensor] = None

    def get_vision_tower(self):
        return self.vision_tower

    def get_model(self):
        return self

    @property
    def device(self):
        return self.native.device

    @property
    def frame_feature(self):          # the reference exposes the raw patch history; here the history is the token store
        return None if self.stream.num_frames == 0 else self.stream.tokens()

    @frame_feature.setter
    def frame_feature(self, v):       # `model.frame_feature = None` is how callers reset a stream (inference_..._ddp.py:373)
        if v is not None:
            raise ValueError("frame_feature can only be reset to None")
        self.stream.reset()
        self._kv_ids = []

    # ---- a3/a5-a9: one call = the new frame(s) of this tick
    @torch.no_grad()
    def _perceive(self, images_or_videos: torch.Tensor) -> Tuple[torch.Tensor, int]:
        if self.native.cfg.conn_d_state == 0:
            raise NotImplementedError("this checkpoint has a stock VideoLLaMA2 projector (no Mamba connector, no event gate): only the offline "
                                      "generate() path exists for it; the streaming calls need mm_projector_type 'mamba'")
        x = images_or_videos
        assert x.dim() == 4, "expected [n,3,H,W] pixel_values or [n,H,W,3] uint8 frames"      # videollama2_arch.py:179 (5-D after unsqueeze)
        if x.shape[0] > 600:
            x = x[-600:]                                                                        # videollama2_arch.py:186-187
        x = x.to(self.device)
        nat, cfg = self.native, self.native.cfg
        logits = None
        tick_logits = []
        step = min(32, cfg.max_frames_per_call)
        for i in range(0, x.shape[0], step):
            xi = x[i:i + step].contiguous()
            if xi.dtype == torch.uint8 and xi.shape[-1] == 3:
                logits, dec = self.stream.push_frames(xi)
            else:
                if xi.dtype not in _DT:
                    xi = xi.float()
                n = xi.shape[0]
                pooled = torch.empty(n, cfg.vit_hidden, dtype=torch.float32, device=self.device)
                check(nat.lib.sm_vit_encode_pixels(nat.h, xi.data_ptr(), _DT[xi.dtype], n, pooled.data_ptr(), None, _stream()), "sm_vit_encode_pixels")
                logits, dec = self.stream.push_pooled(pooled)
            tick_logits.append(logits)
        self.last_gate_logits = logits[-1]
        self._tick_logits = torch.cat(tick_logits)
        return logits[-1], int(dec[-1].item())        # the per-tick device->host read (videollama2_arch.py:941 .item())

    @torch.no_grad()
    def _perceive_features(self, feats: torch.Tensor) -> None:
        """pre-extracted tower features [t, P, C] (the on-disk cache of row a15, or CLIPVisionTower.forward output): patch-mean
        -> connector + gate, appended to this stream like frames would be (mamba_encode_images_or_videos_score,
        videollama2_arch.py:206-207 -> temporal_aggregator)"""
        nat, cfg = self.native, self.native.cfg
        if feats.dim() == 4 and feats.shape[0] == 1:
            feats = feats[0]
        if feats.dim() != 3 or feats.shape[-1] != cfg.vit_hidden:
            raise ValueError(f"expected features [t, patches, {cfg.vit_hidden}], got {tuple(feats.shape)}")
        x = feats.to(self.device)
        if x.dtype not in _DT:
            x = x.float()
        x = x.contiguous()
        t, P, Cw = x.shape
        pooled = torch.empty(t, Cw, dtype=torch.float32, device=self.device)
        check(nat.lib.sm_pool_rows(x.data_ptr(), _DT[x.dtype], t, P, Cw, pooled.data_ptr(), _stream()), "sm_pool_rows")
        cap = 32
        logits = [self.stream.push_pooled(pooled[i:i + cap].contiguous())[0] for i in range(0, t, cap)]
        self._tick_logits = torch.cat(logits)
        self.last_gate_logits = self._tick_logits[-1]

    # ---- a10: sentinel expansion (videollama2_arch.py:948-984)
    def _expand(self, input_ids: Sequence[int]) -> List[int]:
        starts = [0] + self.interval_id_list[:-1]
        seq: List[int] = []
        k = 0
        for t in input_ids:
            if t == MMODAL_TOKEN_INDEX["VIDEO"]:
                seq.extend(-(f + 1) for f in range(starts[k], self.interval_id_list[k]))
                k += 1
            else:
                seq.append(int(t))
        return seq

    # ---- a12: generate from the spliced context with KV prefix reuse (greedy; or sampled, serve/model_worker.py:247-282)
    def _sample(self, temperature: float, top_p: float, generator=None) -> torch.Tensor:
        """HF's sampling rule on the pending position's logits (TemperatureLogitsWarper, then TopPLogitsWarper, then
        multinomial): plain torch on the [vocab] logits vector -- token choice, not a kernel of the path."""
        lg, _ = self.stream.logits()
        lg = lg / max(float(temperature), 1e-6)
        if top_p < 1.0:
            srt, idx = torch.sort(lg, descending=False)
            cum = torch.softmax(srt, dim=-1).cumsum(dim=-1)
            drop = cum <= (1.0 - top_p)
            drop[-1] = False                                  # min_tokens_to_keep = 1
            lg = lg.masked_fill(torch.zeros_like(drop).scatter(0, idx, drop), float("-inf"))
        return torch.multinomial(torch.softmax(lg, dim=-1), 1, generator=generator).to(torch.int32)

    def _check_ids(self, seq: Sequence[int]) -> None:
        """text ids must index the checkpoint's embedding table, frame sentinels the frames pushed so far.  The loader ADDS
        <im_patch> / <im_start> / <im_end> to the tokenizer (model/builder.py; reference builder.py:186-191 then grows the table by
        fresh rows); a prompt that spells one of them out tokenises to an id behind the table -- rejected here like torch's
        embedding raises IndexError, instead of indexing past the device buffer (the kernel reads such rows as zeros)."""
        vocab, T = self.native.cfg.llm_vocab, self.stream.num_frames
        for t in seq:
            if t >= vocab:
                raise IndexError(f"token id {t} is outside the checkpoint's embedding table ({vocab} rows)")
            if t < 0 and -t - 1 >= T:
                raise IndexError(f"frame token {-t - 1} referenced, the stream holds {T}")

    def _begin_generate(self, seq: List[int], max_new_tokens: int) -> int:
        """prefill `seq` behind the longest cached prefix (KV prefix reuse); returns the number of new tokens that still fit"""
        self._check_ids(seq)
        if len(seq) + max_new_tokens > self.max_seq:
            max_new_tokens = self.max_seq - len(seq)
            if max_new_tokens <= 0:
                raise ValueError(f"context of {len(seq)} tokens exceeds max_seq={self.max_seq}")
        lcp = 0
        for a, b in zip(seq, self._kv_ids):
            if a != b:
                break
            lcp += 1
        lcp = min(lcp, len(seq) - 1, self.stream.kv_len)
        self.stream.set_kv_len(lcp)
        self.stream.prefill(torch.tensor(seq[lcp:], dtype=torch.int32, device=self.device))
        self._kv_ids = list(seq)
        return max_new_tokens

    def _accept_tokens(self, out: List[int], ids: Sequence[int], seq_len: int, stopping_criteria=None) -> Tuple[List[int], bool]:
        """take the freshly decoded `ids` one by one up to the first stop (EOS / stopping criteria): the accepted ones join
        `out`; the K/V of speculative tokens after a stop is dropped.  -> (accepted ids, stopped?)"""
        fresh: List[int] = []
        for tok in ids:
            out.append(tok)
            fresh.append(tok)
            self._kv_ids.append(tok)
            done = self.eos_token_id is not None and tok == self.eos_token_id
            if not done and stopping_criteria is not None:
                t = torch.tensor([out], dtype=torch.long)
                done = all(c(t, None) for c in stopping_criteria)
            if done:
                self._kv_ids = self._kv_ids[:seq_len + len(out)]
                self.stream.set_kv_len(len(self._kv_ids))
                return fresh, True
        return fresh, False

    @torch.no_grad()
    def _generate_iter(self, seq: List[int], max_new_tokens: int, stopping_criteria=None, do_sample: bool = False,
                       temperature: float = 1.0, top_p: float = 1.0, generator=None):
        """yields the new ids in chunks (what a TextIteratorStreamer consumer sees); the concatenation is `_generate`'s list"""
        max_new_tokens = self._begin_generate(seq, max_new_tokens)
        out: List[int] = []
        done = False
        while not done and len(out) < max_new_tokens:
            if do_sample:                                     # one step at a time: the choice needs this position's logits
                self.stream.set_next_token(self._sample(temperature, top_p, generator))
                n = 1
            else:
                n = min(self.decode_chunk, max_new_tokens - len(out))
            ids = self.stream.decode(n).cpu().tolist()          # n speculative greedy steps, one host sync
            fresh, done = self._accept_tokens(out, ids, len(seq), stopping_criteria)
            yield fresh

    def _generate(self, seq: List[int], max_new_tokens: int, stopping_criteria=None, **sample_kw) -> List[int]:
        return [t for chunk in self._generate_iter(seq, max_new_tokens, stopping_criteria, **sample_kw) for t in chunk]

    # ---- f1: teacher-forced evaluation forward (videollama2_mistral.py:173-259, videollama2_arch.py:613-753), batch 1
    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor = None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels: Optional[torch.Tensor] = None, use_cache=None, output_attentions=None,
                output_hidden_states=None, images=None, return_dict=None, cls_output=None, **kwargs):
        """model(input_ids, labels=..., images=[clips, ["video"]], timestamp=..., llm_eval=True) -> (output, labels) with
        output.logits fp32 [1, S, vocab] and output.loss (HF shifted CE); model_type="cls" -> the batch gate evaluation
        (data_type "train" -> output, else (output, labels)).  The stream state of this object is reset and reused."""
        if inputs_embeds is not None:
            raise NotImplementedError("`inputs_embeds` is not supported")
        if "timestamp" not in kwargs:
            raise NotImplementedError("only the stream path (timestamp=...) of forward() is part of this build")
        if input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise NotImplementedError("teacher-forced forward: batch size 1 (eval/inference_video_ego4d_stream_parallel_new.py:186)")
        model_type, data_type = kwargs.pop("model_type", None), kwargs.pop("data_type", None)
        Xs, keys = images
        sample_type, sample_per = getattr(self, "sample_type", "all"), getattr(self, "sample_per", 0.5)
        ids, lab_in = input_ids[0].tolist(), (labels[0].tolist() if labels is not None else None)
        if model_type == "cls":
            sample_type = "all"
        seq, is_frame, gate_lg, feature_idx = self._splice_clips(ids, Xs, keys, sample_type, sample_per)
        if model_type == "cls":
            out, lab = gate_eval_outputs(gate_lg, feature_idx)
            return out if data_type == "train" else (out, lab)
        if self.native.cfg.llm_layers == 0:
            raise RuntimeError("perception-only model: no LLM loaded")
        new_labels: List[int] = []
        if lab_in is not None:
            sentinels = [MMODAL_TOKEN_INDEX[key.upper()] for key in keys]
            text_labels = iter(l for t, l in zip(ids, lab_in) if t not in sentinels)
            new_labels = [IGNORE_INDEX if f else next(text_labels) for f in is_frame]
        if len(seq) > self.max_seq:
            raise ValueError(f"spliced sequence of {len(seq)} tokens exceeds max_seq={self.max_seq}")
        self.stream.set_kv_len(0)
        self._kv_ids = []
        self._check_ids(seq)
        logits = self.stream.forward_logits(torch.tensor(seq, dtype=torch.int32, device=self.device))
        self._kv_ids = list(seq)
        loss = None
        out_labels = None
        if lab_in is not None:
            out_labels = torch.tensor([new_labels], dtype=torch.long)
            shifted = torch.tensor(new_labels[1:] + [IGNORE_INDEX], dtype=torch.int32)
            nll, _ = _native.cross_entropy(logits, shifted)
            n_scored = int((shifted != IGNORE_INDEX).sum())
            loss = nll.sum() / max(n_scored, 1)
        output = SimpleNamespace(loss=loss, logits=logits.unsqueeze(0), past_key_values=None, hidden_states=None, attentions=None)
        if kwargs.pop("llm_eval", None):
            return output, out_labels
        return output

    __call__ = forward

    # ---- f4: offline generate (videollama2_mistral.py:261-318, non-score branch), greedy
    def _splice_clips(self, ids: Sequence[int], clips, keys, sample_type: str = "all", sample_per: float = 0.5, features: bool = False):
        """all clips -> ViT -> one connector pass (stream reset first); sentinels -> (optionally sub-sampled) frame indices.
        -> (sequence of ids with frame positions as -(index+1), per-position 'is frame' flags, gate logits of every frame)."""
        self.frame_feature = None
        if self.native.cfg.conn_d_state == 0:
            return self._splice_clips_stc(ids, clips, keys, features)
        counts, all_logits = [], []
        for clip in clips:
            if clip.shape[0] > 600 and not features:
                clip = clip[-600:]                                        # videollama2_arch.py:150-151
            if features:
                self._perceive_features(clip)
                if clip.dim() == 4:
                    clip = clip[0]
            else:
                self._perceive(clip)
            all_logits.append(self._tick_logits)
            counts.append(int(clip.shape[0]))
        feature_idx = [sum(counts[:i + 1]) for i in range(len(counts))]
        starts = [0] + feature_idx[:-1]
        sentinels = [MMODAL_TOKEN_INDEX[key.upper()] for key in keys]
        seq: List[int] = []
        is_frame: List[bool] = []
        k = 0
        for t in ids:
            if t in sentinels:
                n_clip = feature_idx[k] - starts[k]
                if sample_type == "log":
                    sel = exponential_sampling_indices(n_clip, sample_per)
                elif sample_type == "similarity":
                    sel = similarity_sampling_indices(self.stream.tokens(starts[k], n_clip), sample_per)
                else:
                    sel = list(range(n_clip))
                seq.extend(-(starts[k] + f + 1) for f in sel)
                is_frame.extend([True] * len(sel))
                k += 1
            else:
                seq.append(int(t))
                is_frame.append(False)
        return seq, is_frame, torch.cat(all_logits), feature_idx

    def _splice_clips_stc(self, ids: Sequence[int], clips, keys, features: bool):
        """stock VideoLLaMA2 path (videollama2_arch.py:113-133 encode_images_or_videos -> temporal_aggregator :303-309): every clip
        [t, 3, H, W] (or u8 [t, H, W, 3], or with `features` the tower's [t, 576, C]) -> tower -> [1, t, 576, C] -> STC-family
        connector -> tokens [n, hidden], appended to the token store and spliced at the clip's sentinel."""
        counts = []
        for clip in clips:
            clip = clip.to(self.device)
            if features:
                feats = clip[0] if clip.dim() == 4 else clip
            elif clip.dtype == torch.uint8 and clip.shape[-1] == 3:
                B = self.native.cfg.max_frames_per_call
                feats = torch.cat([self.native.vit_encode(clip[i:i + B].contiguous(), return_feats=True)[1] for i in range(0, clip.shape[0], B)])
            else:
                feats = self.vision_tower(clip)
            toks = self.mm_projector(feats.unsqueeze(0))[0].contiguous()
            self.stream.write_tokens(self.stream.num_frames, toks)
            counts.append(int(toks.shape[0]))
        ends = [sum(counts[:i + 1]) for i in range(len(counts))]
        starts = [0] + ends[:-1]
        sentinels = [MMODAL_TOKEN_INDEX[key.upper()] for key in keys]
        seq: List[int] = []
        is_frame: List[bool] = []
        k = 0
        for t in ids:
            if t in sentinels:
                seq.extend(-(f + 1) for f in range(starts[k], ends[k]))
                is_frame.extend([True] * counts[k])
                k += 1
            else:
                seq.append(int(t))
                is_frame.append(False)
        return seq, is_frame, torch.empty(0, 2), ends

    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, images_or_videos=None, modal_list=None, **kwargs):
        """model.generate(input_ids, images_or_videos=[clip, ...], modal_list=["video"], do_sample=False, max_new_tokens=..,
        stopping_criteria=[...]) -> LongTensor [1, n_new] (new ids only: the inputs were embeddings).  As in the reference,
        generate() splices EVERY frame token (it does not forward sample_type / sample_per)."""
        kwargs.pop("position_ids", None)
        kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        # score_video=True: the reference routes to prepare_inputs_labels_for_multimodal_score (videollama2_mistral.py:276-289),
        # whose `encode_images_or_videos_score` does not exist in the tree (videollama2_arch.py:502: AttributeError on every
        # call).  The evident intent -- clips given as PRE-EXTRACTED tower features [t, 576, 1024] (the feature cache of row
        # a15), connector only (mamba_encode_images_or_videos_score, :206-207) -- is what this build does for that flag.
        score_video = bool(kwargs.pop("score_video", None))
        if inputs.dim() != 2 or inputs.shape[0] != 1:
            raise NotImplementedError("generate: batch size 1")
        if self.native.cfg.llm_layers == 0:
            raise RuntimeError("perception-only model: no LLM loaded")
        ids = inputs[0].tolist()
        if images_or_videos is not None and len(images_or_videos):
            seq, _, _, _ = self._splice_clips(ids, images_or_videos, modal_list or ["video"], features=score_video)
        else:
            seq = [int(t) for t in ids]
        do_sample = bool(kwargs.get("do_sample", False))
        sample_kw = dict(do_sample=do_sample, temperature=float(kwargs.get("temperature") or 1.0), top_p=float(kwargs.get("top_p") or 1.0),
                         generator=kwargs.get("generator")) if do_sample else {}
        streamer = kwargs.get("streamer")                 # HF streamer protocol: put(ids) per step, end() -- model_worker.py:263
        new_ids: List[int] = []
        if streamer is not None:                          # HF hands the (here empty: the inputs are embeddings) prompt ids over first
            streamer.put(torch.empty(1, 0, dtype=torch.long))
        for chunk in self._generate_iter(seq, int(kwargs.get("max_new_tokens", 1024)), kwargs.get("stopping_criteria"), **sample_kw):
            new_ids.extend(chunk)
            if streamer is not None and chunk:
                streamer.put(torch.tensor(chunk, dtype=torch.long))
        if streamer is not None:
            streamer.end()
        return torch.tensor([new_ids], dtype=torch.long)

    @torch.no_grad()
    def generate_iter(self, inputs: torch.Tensor, images_or_videos=None, modal_list=None, max_new_tokens: int = 1024, stopping_criteria=None,
                      do_sample: bool = False, temperature: float = 1.0, top_p: float = 1.0, generator=None):
        """`generate` as a generator of id chunks, for callers that forward text while it is produced (the serving worker): the
        native decode loop runs `decode_chunk` greedy steps per host sync and hands each accepted chunk over at once -- no second
        thread, no queue.  The concatenation of the chunks is `generate`'s result."""
        if inputs.dim() != 2 or inputs.shape[0] != 1:
            raise NotImplementedError("generate_iter: batch size 1")
        if self.native.cfg.llm_layers == 0:
            raise RuntimeError("perception-only model: no LLM loaded")
        ids = inputs[0].tolist()
        if images_or_videos is not None and len(images_or_videos):
            seq, _, _, _ = self._splice_clips(ids, images_or_videos, modal_list or ["video"], features=False)
        else:
            seq = [int(t) for t in ids]
        sample_kw = dict(do_sample=True, temperature=float(temperature or 1.0), top_p=float(top_p or 1.0), generator=generator) if do_sample else {}
        yield from self._generate_iter(seq, int(max_new_tokens), stopping_criteria, **sample_kw)

    @torch.no_grad()
    def stream_generate_demo(self, inputs: Optional[torch.Tensor] = None, images_or_videos: Optional[torch.Tensor] = None,
                             modal_list=None, **kwargs):
        """-> (decoded str | None, cls_pred).  Mirrors videollama2_mistral.py:385-439 incl. its error behaviour."""
        kwargs.pop("position_ids", None)
        kwargs.pop("attention_mask", None)
        kwargs.pop("score_video", None)
        tokenizer = kwargs.pop("tokenizer", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        _, cls_pred = self._perceive(images_or_videos)
        if cls_pred == 0:
            return None, cls_pred
        self.interval_id_list.append(self.stream.num_frames)
        if self.native.cfg.llm_layers == 0:
            raise RuntimeError("perception-only model: no LLM loaded")
        ids = inputs[0].tolist() if inputs.dim() == 2 else inputs.tolist()
        seq = self._expand(ids)
        sample_kw = dict(do_sample=True, temperature=float(kwargs.get("temperature") or 1.0), top_p=float(kwargs.get("top_p") or 1.0),
                         generator=kwargs.get("generator")) if kwargs.get("do_sample", False) else {}
        new_ids = self._generate(seq, int(kwargs.get("max_new_tokens", 1024)), kwargs.get("stopping_criteria"), **sample_kw)
        self.last_new_ids = new_ids
        output = tokenizer.batch_decode([new_ids], skip_special_tokens=True)[0].strip()
        return output, cls_pred


# ------------------------------------------------------------------------------------------------ loading
def build_from_state_dicts(cfg: PathConfig, vision_sd: Dict[str, torch.Tensor], projector_sd: Dict[str, torch.Tensor],
                           lm_sd: Optional[Dict[str, torch.Tensor]] = None, device: str = "cuda:0", **kw) -> Videollama2MistralForCausalLM:
    """weights given as the reference's state_dict pieces (names relative to vision_model / mm_projector / the LM root)."""
    nat = NativeModel(cfg, device)
    for k, v in vision_sd.items():
        nat.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v)
    for k, v in projector_sd.items():
        nat.load_tensor("model.mm_projector." + k, v)
    if lm_sd is not None:
        for k, v in lm_sd.items():
            nat.load_tensor(k, v)
    miss = nat.missing()
    if miss:
        raise ValueError(f"checkpoint is missing tensors for: {miss[:8]}{' ...' if len(miss) > 8 else ''}")
    nat.finalize()
    return Videollama2MistralForCausalLM(nat, **kw)

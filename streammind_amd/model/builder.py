"""Checkpoint loader of the drop-in: `load_pretrained_model` with the signature and return value of
streammind/model/builder.py:30-210, for a merged (non-LoRA) Mistral StreamMind checkpoint directory:

    <model_path>/config.json                      HF Mistral keys + mm_projector_type / mm_vision_tower / mm_hidden_size /
                                                  mm_vision_select_layer / mm_vision_select_feature (videollama2_arch.py:69-73)
    <model_path>/*.safetensors | pytorch_model*.bin   LM weights under HF names, `model.mm_projector.*`, and -- when the
                                                  checkpoint was saved with its tower attached -- `model.vision_tower.vision_tower.*`
    <model_path>/mm_projector.bin                 optional separate projector weights (load_mm_projector, builder.py:66-85)
    <config.mm_vision_tower>/                     the CLIP tower directory, delay-loaded when the LM checkpoint does not carry
                                                  the tower (clip_encoder.py:18-29): config.json, weights, preprocessor_config.json
    tokenizer files                               AutoTokenizer.from_pretrained(model_path)

Every tensor goes to the native model under its checkpoint name (fp16 as stored, builder.py:54; the library converts);
nothing is computed here."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, Iterator, Optional, Tuple

import torch

from ..constants import DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN
from ..native import NativeModel, PathConfig
from .stream_model import Videollama2MistralForCausalLM

# ClsNet is MistralConfig(vocab_size=2, num_hidden_layers=4) with every other field at its default
# (multimodal_projector/builder.py:373-378): the gate's shape is hard-wired in the reference, not stored in config.json
_GATE_DEFAULTS = dict(num_hidden_layers=4, num_attention_heads=32, num_key_value_heads=8, intermediate_size=14336, rms_norm_eps=1e-6)
_TOWER_ROOTS = ("embeddings.", "pre_layrnorm.", "encoder.", "post_layernorm.")


def _checkpoint_tensors(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """(name, tensor) of every weight file in a HF checkpoint directory: safetensors shards (with or without an index),
    else pytorch_model*.bin shards."""
    names = sorted(os.listdir(path))
    st = [f for f in names if f.endswith(".safetensors")]
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        st = sorted(set(json.load(open(idx))["weight_map"].values()))
    if st:
        from safetensors import safe_open
        for f in st:
            with safe_open(os.path.join(path, f), framework="pt") as sf:
                for k in sf.keys():
                    yield k, sf.get_tensor(k)
        return
    bins = [f for f in names if f.startswith("pytorch_model") and f.endswith(".bin")]
    if not bins:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    for f in bins:
        for k, v in torch.load(os.path.join(path, f), map_location="cpu", weights_only=True).items():
            yield k, v


def path_config_from_checkpoint(cfgj: dict, vision_cfg: dict, **overrides) -> PathConfig:
    """config.json of the LM (+ mm_* keys) and of the CLIP tower -> dimensions of the native path.  `mm_gate_config` is not a
    reference key: it lets shrunken test checkpoints state a gate smaller than the reference's hard-wired one."""
    vj = vision_cfg.get("vision_config", vision_cfg)
    g = dict(_GATE_DEFAULTS)
    g.update(cfgj.get("mm_gate_config") or {})
    if cfgj.get("mm_hidden_size", vj["hidden_size"]) != vj["hidden_size"]:
        raise ValueError(f"config.mm_hidden_size={cfgj['mm_hidden_size']} does not match the vision tower width {vj['hidden_size']}")
    if cfgj.get("mm_vision_select_feature", "patch") != "patch":
        raise ValueError(f"Unexpected select feature: {cfgj['mm_vision_select_feature']}")          # clip_encoder.py:38
    kw = dict(
        vit_image=vj["image_size"], vit_patch=vj["patch_size"], vit_hidden=vj["hidden_size"], vit_heads=vj["num_attention_heads"],
        vit_mlp=vj["intermediate_size"], vit_layers=vj["num_hidden_layers"], vit_select_layer=cfgj.get("mm_vision_select_layer", -2),
        vit_eps=vj.get("layer_norm_eps", 1e-5), conn_d_model=cfgj["hidden_size"],
        gate_layers=g["num_hidden_layers"], gate_heads=g["num_attention_heads"], gate_kv_heads=g["num_key_value_heads"],
        gate_mlp=g["intermediate_size"], gate_eps=g["rms_norm_eps"],
        llm_layers=cfgj["num_hidden_layers"], llm_heads=cfgj["num_attention_heads"],
        llm_kv_heads=cfgj.get("num_key_value_heads", cfgj["num_attention_heads"]), llm_mlp=cfgj["intermediate_size"],
        llm_vocab=cfgj["vocab_size"], llm_eps=cfgj.get("rms_norm_eps", 1e-6),
        # transformers 4.x writes `rope_theta`, 5.x nests it under `rope_parameters`
        llm_rope_theta=cfgj.get("rope_theta") or (cfgj.get("rope_parameters") or {}).get("rope_theta", 1e4))
    if cfgj.get("head_dim") not in (None, cfgj["hidden_size"] // cfgj["num_attention_heads"]):
        raise ValueError(f"head_dim={cfgj['head_dim']} != hidden_size / num_attention_heads is not supported by the attention kernels")
    # what the kernels hard-wire (Mistral / CLIP as the reference uses them): anything else must fail here, not compute something else
    if vj.get("hidden_act", "quick_gelu") != "quick_gelu":
        raise NotImplementedError(f"vision tower hidden_act={vj['hidden_act']!r}: the tower's MLP epilogue is CLIP's quick_gelu")
    if cfgj.get("hidden_act", "silu") != "silu":
        raise NotImplementedError(f"LLM hidden_act={cfgj['hidden_act']!r}: the gate / up kernels fuse SiLU (Mistral)")
    if cfgj.get("tie_word_embeddings", False):
        raise NotImplementedError("tie_word_embeddings=true: lm_head and embed_tokens are separate tensors on this path (Mistral checkpoints)")
    if cfgj.get("attention_bias", False) or cfgj.get("mlp_bias", False):
        raise NotImplementedError("attention_bias / mlp_bias: the LLM's projections are bias-free on this path (Mistral)")
    rs = cfgj.get("rope_scaling") or {k: v for k, v in (cfgj.get("rope_parameters") or {}).items() if k not in ("rope_theta", "rope_type")}
    rtype = (cfgj.get("rope_parameters") or {}).get("rope_type", (cfgj.get("rope_scaling") or {}).get("type", "default"))
    if rtype not in ("default", None) or (cfgj.get("rope_scaling") or None):
        raise NotImplementedError(f"rope scaling ({rtype!r}, {rs!r}): only the plain rotary embedding with rope_theta is built")
    kw.update(overrides)
    return PathConfig(**kw)


def load_pretrained_model(model_path, model_base=None, model_name="VideoLLaMA2-7B", load_8bit=False, load_4bit=False,
                          device_map="auto", device="cuda", use_flash_attn=False, **kwargs):
    """-> (tokenizer, model, image_processor, context_len).  Extra keyword arguments of this build: max_frames_per_call
    (ViT batch capacity, default 8), max_seq (KV capacity per stream), max_frames (token store), weights_fp8, torch_dtype
    (torch.float16 = the reference's loading precision, default; torch.bfloat16 = its evaluation precision)."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantisation is not part of the MI355X path")
    if model_base is not None or "lora" in model_name.lower():
        raise NotImplementedError("LoRA / base-model merging is a training-side feature (out of scope): pass a merged checkpoint")
    from transformers import AutoTokenizer, CLIPImageProcessor
    if not os.path.isdir(model_path):
        raise FileNotFoundError(f"{model_path}: hub ids cannot be resolved here (no network): pass a local checkpoint directory")
    cfgj = json.load(open(os.path.join(model_path, "config.json")))
    ptype = cfgj.get("mm_projector_type") or ""
    from .stc_connector import _TYPES as _STC_TYPES
    import re
    # stock VideoLLaMA2 projectors (builder.py:119-154): tower + host-side projector + LLM, no event gate
    stc = ptype in _STC_TYPES or ptype == "linear" or re.match(r"^mlp(\d+)x_gelu$", ptype) is not None
    if "mamba" not in ptype and not stc:
        raise ValueError(f"Unsupported projector type {cfgj.get('mm_projector_type')}!!!")      # videollama2_arch.py:321
    tower_dir = cfgj.get("mm_vision_tower")
    if not tower_dir or not os.path.isdir(tower_dir):
        raise FileNotFoundError(f"config.mm_vision_tower={tower_dir!r} is not a local directory (hub ids cannot be resolved here)")
    vj = json.load(open(os.path.join(tower_dir, "config.json")))
    # precision of the vision tower's operands: the reference loads EVERYTHING as fp16 (builder.py:54, tower :201) and its
    # teacher-forced evaluation casts to bf16 (eval/inference_video_ego4d_stream_parallel_new.py:160).  fp16 is the default here
    # too (gate logits 1e-4 from fp32 at full size; the LLM keeps the checkpoint's fp16 weights bit for bit and computes with fp16
    # operands -- llm_fp16; the connector / gate likewise -- proj_fp16, with their fp32 activations carried as fp16 hi/lo pairs);
    # torch_dtype=torch.bfloat16 selects bf16 operands throughout.  fp8 weights imply the bf16 gate + LLM.
    tdt = kwargs.pop("torch_dtype", torch.float16)
    if tdt not in (torch.float16, torch.bfloat16):
        raise ValueError(f"torch_dtype must be torch.float16 or torch.bfloat16, got {tdt}")
    w8 = bool(kwargs.pop("weights_fp8", False))
    over = dict(max_frames_per_call=kwargs.pop("max_frames_per_call", 8), weights_fp8=w8, llm_fp16=(tdt == torch.float16 and not w8), proj_fp16=(tdt == torch.float16 and not w8),
                vit_fp16=(tdt == torch.float16 and vj.get("vision_config", vj)["hidden_size"] // vj.get("vision_config", vj)["num_attention_heads"] == 64))
    ppath = os.path.join(tower_dir, "preprocessor_config.json")
    if os.path.exists(ppath):                               # the normalisation constants belong to the tower checkpoint
        pj = json.load(open(ppath))
        if pj.get("image_mean") and pj.get("image_std"):
            over.update(img_mean=tuple(pj["image_mean"]), img_std=tuple(pj["image_std"]))
    if stc:
        over.update(conn_d_state=0, gate_layers=0)          # native model without the Mamba connector / gate (sm_config_t.conn_d_state)
    # builder.py:150-152, 186-196: the tokenizer comes first because it decides the size of the embedding table -- the patch / start / end
    # tokens are ADDED to it and the reference then calls resize_token_embeddings(len(tokenizer)), growing embed_tokens and lm_head by
    # freshly initialised rows (transformers 4.44: normal(0, initializer_range), i.e. unspecified values).  Same here: the native table has
    # len(tokenizer) rows, the checkpoint fills the rows it has, the new rows get seeded normal(0, initializer_range) values.  A checkpoint
    # SAVED after the resize already holds its (trained) rows and config.vocab_size says so: then nothing is grown.
    try:
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False)
    except Exception:                                       # no slow-tokenizer files: builder.py:150-152 falls back to the fast one too
        tokenizer = AutoTokenizer.from_pretrained(model_path, model_max_length=2048, padding_side="right", use_fast=True)
    if cfgj.get("mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if cfgj.get("mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    if cfgj.get("sliding_window"):
        over.update(llm_sliding_window=int(cfgj["sliding_window"]))
    ckpt_vocab = int(cfgj["vocab_size"])
    grown_vocab = max(ckpt_vocab, len(tokenizer))
    if grown_vocab > ckpt_vocab:
        over.update(llm_vocab=grown_vocab)
    cfg = path_config_from_checkpoint(cfgj, vj, **over)
    # Mistral's sliding window (4096 for Mistral-7B-v0.1, null for v0.2): the attention kernels mask it like HF's MistralModel (a query at
    # position p sees keys (p - window, p]; sm_config_t.llm_sliding_window), so the KV capacity may exceed it.  Default capacity 4096.
    window = cfgj.get("sliding_window")
    max_seq = kwargs.pop("max_seq", None)
    if max_seq is None:
        max_seq = 4096
    max_seq -= max_seq % 64
    dev = "cuda:0" if device == "cuda" else device
    nat = NativeModel(cfg, dev)
    proj_sd = {}                                            # STC family: the projector's tensors go to the host-side connector class
    def grown(k, v):            # resize_token_embeddings on the way in: rows [ckpt_vocab, len(tokenizer)) of the two vocabulary-sized tensors
        if grown_vocab > v.shape[0] and v.dim() == 2 and (k.endswith("embed_tokens.weight") or k.endswith("lm_head.weight")) and "cls_net" not in k:
            g = torch.Generator().manual_seed(20240 + (1 if k.endswith("lm_head.weight") else 0))
            extra = torch.randn(grown_vocab - v.shape[0], v.shape[1], generator=g) * float(cfgj.get("initializer_range", 0.02))
            return torch.cat([v, extra.to(v.dtype)], dim=0)
        return v
    for k, v in _checkpoint_tensors(model_path):
        if stc and "mm_projector." in k:
            proj_sd[k.split("mm_projector.", 1)[1]] = v
        else:
            nat.load_tensor(k, grown(k, v))
    pbin = os.path.join(model_path, "mm_projector.bin")
    if os.path.exists(pbin):                                # builder.py:141-142 -> load_mm_projector
        for k, v in torch.load(pbin, map_location="cpu", weights_only=True).items():
            if stc:
                proj_sd[k.split("mm_projector.", 1)[1] if "mm_projector." in k else k] = v
            else:
                nat.load_tensor(k if "mm_projector." in k else "model.mm_projector." + k, v)
    if any(m.startswith("vit.") for m in nat.missing()):    # the tower is delay-loaded from its own checkpoint (clip_encoder.py:18-29)
        for k, v in _checkpoint_tensors(tower_dir):
            # a full CLIP directory also holds text_model.*, visual_projection, text_projection, logit_scale: never read
            if k.startswith("vision_model."):
                nat.load_tensor("model.vision_tower.vision_tower." + k, v)
            elif k.startswith(_TOWER_ROOTS):                # CLIPVisionModel saved by transformers >= 5: no vision_model. prefix
                nat.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v)
    if nat.missing():
        raise ValueError(f"checkpoint incomplete, missing: {nat.missing()[:8]}")
    nat.finalize()
    image_processor = CLIPImageProcessor.from_pretrained(tower_dir)
    context_len = cfgj.get("max_sequence_length", 2048)                      # builder.py:205-208
    model = Videollama2MistralForCausalLM(nat, max_frames=kwargs.pop("max_frames", 4096), max_seq=max_seq,
                                          eos_token_id=tokenizer.eos_token_id)
    model.config = SimpleNamespace(**cfgj)
    if stc:
        from .stc_connector import build_vision_projector
        pc = SimpleNamespace(mm_projector_type=ptype, mm_hidden_size=cfgj.get("mm_hidden_size", cfg.vit_hidden), hidden_size=cfgj["hidden_size"])
        model.mm_projector = build_vision_projector(pc, device=dev).load_state_dict(proj_sd)
    return tokenizer, model, image_processor, context_len

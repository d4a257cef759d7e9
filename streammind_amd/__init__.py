"""streammind_amd -- MI355X-native drop-in for the StreamMind streaming hot path.

Public API mirrors the reference's package surface (streammind/__init__.py:14-103): `model_init`, the offline `infer` -> str and
`x_infer`.  The streaming tick (eval/video_score_stream_demo.py:66-125) is `streammind_amd.eval.video_score_stream_demo.infer`,
re-exported as `stream_infer`.  All arithmetic runs in libstreammind_hip.so."""
from __future__ import annotations

from functools import partial

import torch

from .constants import DEFAULT_MMODAL_TOKEN, MMODAL_TOKEN_INDEX, NUM_FRAMES
from .conversation import SeparatorStyle, conv_templates
from .mm_utils import KeywordsStoppingCriteria, process_video, tokenizer_MMODAL_token


def model_init(model_path=None, model_name="VideoLLaMA2-7B", model_base=None):
    """streammind/__init__.py:14-35 (positional order of the package API; the evaluation scripts' local variant,
    eval/video_score_stream_demo.py:42-63, passes model_base by position 2 -- use the keyword there)."""
    from .model import load_pretrained_model
    from .mm_utils import get_model_name_from_path
    if model_path is None:
        raise FileNotFoundError("model_init: the reference's default is the hub id DAMO-NLP-SG/VideoLLaMA2-7B; there is no network here, pass a local checkpoint directory")
    model_name = get_model_name_from_path(model_path) if model_name is None else model_name
    tokenizer, model, processor, context_len = load_pretrained_model(model_path, model_base, model_name)
    if tokenizer.unk_token is not None:
        tokenizer.pad_token = tokenizer.unk_token
    num_frames = getattr(model.config, "num_frames", NUM_FRAMES)
    version = "v1" if "vicuna" in model_name.lower() else "qwen" if "qwen" in model_name.lower() else "llama_2"
    return model, partial(process_video, aspect_ratio=None, processor=processor, num_frames=num_frames), tokenizer, version


def _generate_reply(model, tokenizer, prompt: str, clips, stop_str: str, do_sample: bool, max_new_tokens: int) -> str:
    """tokenise `prompt` around its <video> sentinel, greedy-generate with the keyword stop, decode the new ids"""
    ids = tokenizer_MMODAL_token(prompt, tokenizer, MMODAL_TOKEN_INDEX["VIDEO"], return_tensors="pt").unsqueeze(0)
    pad = -1 if tokenizer.pad_token_id is None else tokenizer.pad_token_id
    new_ids = model.generate(ids, attention_mask=ids.ne(pad).long(), images_or_videos=clips, modal_list=["video"],
                             do_sample=do_sample, temperature=0.2 if do_sample else 0.0, max_new_tokens=max_new_tokens,
                             use_cache=True, stopping_criteria=[KeywordsStoppingCriteria([stop_str], tokenizer, ids)],
                             pad_token_id=tokenizer.eos_token_id)
    return tokenizer.batch_decode(new_ids, skip_special_tokens=True)[0].strip()


def infer(model, video, instruct, tokenizer, do_sample=False, version="llama_2", max_new_tokens=1024):
    """streammind/__init__.py:38-91 -> str: offline whole-clip question answering -- "<video>\n" + instruct in the `version`
    conversation template, every frame of `video` [T, C, H, W] through the ViT and the connector, ONE generate, the decoded reply.
    `max_new_tokens` (hard-coded 1024 upstream, :84) is the one added keyword, last and defaulted.  The STREAMING tick of the same
    name lives where the reference keeps it, `eval/video_score_stream_demo.py:66-125` (`streammind_amd.eval.video_score_stream_demo.infer`,
    exported here as `stream_infer`)."""
    conv = conv_templates[version].copy()
    conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\n" + instruct)
    conv.append_message(conv.roles[1], None)
    stop = conv.sep if conv.sep_style in [SeparatorStyle.SINGLE] else conv.sep2
    clip = video if video.dtype == torch.uint8 else video.half()
    return _generate_reply(model, tokenizer, conv.get_prompt(), [clip], stop, do_sample, max_new_tokens)


_X_INFER_SUFFIX = {
    "vanilla": "",
    "mcqa": "\nAnswer with the option's letter from the given choices directly and only give the best option.",
    "openend": "\nAnswer the question using a single word or a short phrase with multiple words.",
}


def x_infer(video, question, model, tokenizer, mode="vanilla", do_sample=False, version="llama_2"):
    """streammind/__init__.py:94-103: the three question styles of the offline benchmarks"""
    if mode not in _X_INFER_SUFFIX:
        raise ValueError(f"x_infer: unknown mode {mode!r}")
    return infer(model, video, question + _X_INFER_SUFFIX[mode], tokenizer, do_sample=do_sample, version=version)


def stream_infer(model, video, instruct, tokenizer, do_sample=False, version="mistral_instruct", score_video=None, prompt=None,
                 max_new_tokens=1024):
    """the streaming tick under a name of its own: eval/video_score_stream_demo.py:66-125 -> (reply | None, prompt)"""
    from .eval.video_score_stream_demo import infer as _tick
    return _tick(model, video, instruct, tokenizer, do_sample=do_sample, version=version, score_video=score_video, prompt=prompt,
                 max_new_tokens=max_new_tokens)


infer_offline = infer          # round <= 3 name of the offline API

"""The streaming caller: `eval/video_score_stream_demo.py` of the reference (its twin `eval/video_test_stream_demo.py` has the same
`model_init` / `infer`), so that `from videollama2.eval.video_score_stream_demo import infer, model_init` of a reference caller
resolves to the MI355X path through the alias packages.

    model_init(model_path, model_base=None, model_name=...)   eval/video_score_stream_demo.py:42-63  (model_base by position 2)
    infer(model, video, instruct, tokenizer, do_sample, version, score_video, prompt) -> (reply | None, prompt)      :66-125
    get_index_stream / read_video_stream                                                                             :208-225
    run_stream: the per-frame loop of run_inference_time_metric                                                       :258-299
"""
from __future__ import annotations

from functools import partial

import torch

from ..constants import DEFAULT_MMODAL_TOKEN, MMODAL_TOKEN_INDEX, NUM_FRAMES
from ..conversation import SeparatorStyle, conv_templates
from ..mm_utils import KeywordsStoppingCriteria, process_video, tokenizer_MMODAL_token
from ..video_io import get_index_stream, read_video_stream          # noqa: F401  (reference-named re-exports)


def model_init(model_path, model_base=None, model_name="VideoLLaMA2-7B"):
    """eval/video_score_stream_demo.py:42-63: the scripts' local variant of the package `model_init` -- model_base is
    positional argument 2 here and model_path has no hub default."""
    from ..model import load_pretrained_model
    tokenizer, model, processor, context_len = load_pretrained_model(model_path, model_base, model_name)
    if tokenizer.unk_token is not None:
        tokenizer.pad_token = tokenizer.unk_token
    num_frames = getattr(model.config, "num_frames", NUM_FRAMES)
    version = "v1" if "vicuna" in model_name.lower() else "qwen" if "qwen" in model_name.lower() else "llama_2"
    return model, partial(process_video, aspect_ratio=None, processor=processor, num_frames=num_frames), tokenizer, version


def infer(model, video, instruct, tokenizer, do_sample=False, version="mistral_instruct", score_video=None, prompt=None,
          max_new_tokens=1024):
    """One streaming tick: the new frame(s) in `video`, the running `prompt` (None on the first call) -> (reply text | None,
    prompt).  eval/video_score_stream_demo.py:66-125 in control flow: `version` is overwritten with 'mistral_instruct' (:83) and
    `instruct` with '<video>\\n' (:92) exactly as there; on a fire the prompt grows by " " + reply + " </s>[INST] <video>\\n [/INST]"
    (:123-124).  `max_new_tokens` (the reference hard-codes 1024, :116) is the one added keyword, last and defaulted."""
    modal_index = MMODAL_TOKEN_INDEX["VIDEO"]
    conv = conv_templates["mistral_instruct"].copy()
    tensor = video if video.dtype == torch.uint8 else video.half()
    tensor = tensor.to(model.device)
    if prompt is None:
        conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\n")
        conv.append_message(conv.roles[1], None)
        prompt = conv.get_prompt()
    input_ids = tokenizer_MMODAL_token(prompt, tokenizer, modal_index, return_tensors="pt").unsqueeze(0)
    pad = tokenizer.pad_token_id if tokenizer.pad_token_id is not None else -1
    attention_masks = input_ids.ne(pad).long()
    stop_str = conv.sep if conv.sep_style in [SeparatorStyle.SINGLE] else conv.sep2
    stopping_criteria = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
    outputs, cls_pred = model.stream_generate_demo(
        input_ids, attention_mask=attention_masks, images_or_videos=tensor, modal_list=["video"], do_sample=do_sample,
        temperature=0.2 if do_sample else 0.0, max_new_tokens=max_new_tokens, use_cache=True,
        stopping_criteria=[stopping_criteria], pad_token_id=tokenizer.eos_token_id, score_video=score_video, tokenizer=tokenizer)
    if cls_pred == 1:
        prompt += " " + outputs + " </s>[INST] <video>\n [/INST]"
    return outputs, prompt


def run_stream(model, processor, tokenizer, video_path, cur_fps=2, instruct="", version="llama_2", max_new_tokens=1024,
               on_reply=None):
    """The loop of run_inference_time_metric (eval/video_score_stream_demo.py:271-299) for one video: sample frame indices at
    `cur_fps`, hand every sampled frame to `infer` alone, carry the prompt.  (The reference's `cur_min / cur_sec` filter at
    :282 is always true -- both stay -1 -- and is therefore absent.)  Returns [(frame_id, reply)] for the frames the gate fired on."""
    from PIL import Image
    frame_ids, vr = read_video_stream(video_path, cur_fps)
    prompt, replies = None, []
    for frame_id in frame_ids:
        fr = vr[int(frame_id)]
        img = Image.fromarray(fr.asnumpy() if hasattr(fr, "asnumpy") else fr)
        video_frame = processor([img], num_frames=1)
        pred, prompt = infer(model=model, video=video_frame, instruct=instruct, tokenizer=tokenizer, do_sample=False,
                             version=version, score_video=True, prompt=prompt, max_new_tokens=max_new_tokens)
        if pred is not None and pred != "":
            replies.append((int(frame_id), pred))
            if on_reply is not None:
                on_reply(int(frame_id), pred)
    return replies

"""Host-side mirrors of the reference's text / bookkeeping interface against the goldens (CPU only)."""
import json

import numpy as np
import torch

from streammind_amd import conversation, mm_utils
from streammind_amd.constants import MMODAL_TOKEN_INDEX


def test_prompt_template_matches_reference(gold):
    g = gold("g5_prompt")
    conv = conversation.conv_templates["mistral_instruct"].copy()
    conv.append_message(conv.roles[0], "<video>\n")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == str(g["prompt0"])
    assert conv.sep2 == "</s>" and conv.sep_style == conversation.SeparatorStyle.LLAMA_2


def test_tokenizer_mmodal_token_matches_reference(gold, tiny_tokenizer):
    g = gold("g5_prompt")
    p = str(g["prompt0"])
    prompts = [p]
    for reply in ["the person picks up a knife", "someone washes a plate", "a player kicks the ball"]:
        prompts.append(prompts[-1] + " " + reply + " </s>[INST] <video>\n [/INST]")
    for i, pr in enumerate(prompts):
        ids = mm_utils.tokenizer_MMODAL_token(pr, tiny_tokenizer, MMODAL_TOKEN_INDEX["VIDEO"], return_tensors="pt")
        assert ids.dtype == torch.long and ids.tolist() == g[f"ids{i}"].tolist()
    try:
        mm_utils.tokenizer_MMODAL_token(p, tiny_tokenizer, -201, return_tensors="np")
        raise AssertionError
    except ValueError:
        pass


def test_keywords_stopping_matches_reference(gold, tiny_tokenizer):
    g = gold("g5_prompt")
    inp = torch.tensor([g["ids1"].tolist()])
    crit = mm_utils.KeywordsStoppingCriteria(["</s>"], tiny_tokenizer, inp)
    for cs, want in zip(g["stop_cases"], g["stop_results"]):
        assert crit(torch.tensor([json.loads(str(cs))]), None) == bool(want)


def test_process_video_keeps_u8_frames():
    fr = [np.full((336, 336, 3), i, dtype=np.uint8) for i in range(3)]
    out = mm_utils.process_video(fr, None, aspect_ratio=None, num_frames=3)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (3, 336, 336, 3) and out[2, 0, 0, 0] == 2
    try:
        mm_utils.process_video("/some/video.mp4", None)
        raise AssertionError
    except NotImplementedError:
        pass


def test_sentinel_expansion_and_errors():
    from streammind_amd.model.stream_model import Videollama2MistralForCausalLM as M
    obj = M.__new__(M)
    obj.interval_id_list = [3, 7]
    seq = obj._expand([1, 5, -201, 6, -201, 7])
    assert seq == [1, 5, -1, -2, -3, 6, -4, -5, -6, -7, 7]
    try:
        M.stream_generate_demo(obj, None, None, inputs_embeds=torch.zeros(1))
        raise AssertionError
    except NotImplementedError as e:
        assert "inputs_embeds" in str(e)


def test_feature_cache_stride_and_names(tmp_path):
    from streammind_amd import feature_cache as fc
    from oracle import streammind_oracle as O
    x = torch.arange(1 * 500 * 2 * 3, dtype=torch.float32).reshape(1, 500, 2, 3).to(torch.bfloat16)
    d = tmp_path / "features_video_encode_ddp" / "vid"
    d.mkdir(parents=True)
    p = d / fc.chunk_name("vid", 0, 500)
    torch.save(x, p)
    out = fc.process_file(str(p))
    assert out == O.stride_output_path(str(p)) and "features_video_encode_ddp_fps" in out
    y = torch.load(out)
    assert tuple(y.shape) == (1, 42, 2, 3) and torch.equal(y, O.feature_stride(x))
    assert fc.chunk_name("v", 500, 1000) == "v_encode_feature_frame_500_1000.pt"

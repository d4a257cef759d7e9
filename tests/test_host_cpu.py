"""Host-side mirrors of the reference's text / bookkeeping interface against the goldens (CPU only)."""
import json
import os

import numpy as np
import pytest
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
    with pytest.raises(FileNotFoundError):
        mm_utils.process_video("/some/video.mp4", None)


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


def test_feature_cache_plumbing_vs_reference_golden(tmp_path, gold):
    """config-1 plumbing (golden g13 = the reference's own process_clip_encoder.process_file run on a 31-frame chunk,
    and encode_all_videos_score's chunking / naming statements evaluated for a 1234-frame video): the drop-in's stride writes the
    same file under the same relative path with the same frames; chunk boundaries, names and the rank slicing agree."""
    import json
    from streammind_amd import feature_cache as fc
    from oracle import streammind_oracle as O
    g = gold("g13_feature_cache_plumbing")
    n, P, C = int(g["n"]), int(g["P"]), int(g["C"])
    x = torch.arange(n * P * C, dtype=torch.float32).reshape(1, n, P, C).to(torch.bfloat16)
    src = tmp_path / str(g["src_rel"])
    src.parent.mkdir(parents=True)
    torch.save(x, src)
    out = fc.process_file(str(src))
    assert os.path.relpath(out, tmp_path) == str(g["out_rel"]) == os.path.relpath(O.stride_output_path(str(src)), tmp_path)
    y = torch.load(out)
    assert list(y.shape) == g["out_shape"].tolist() and fc.SEGMENT == int(g["segment"])
    assert y[0, :, 0, 0].float().tolist() == g["kept_frames"].tolist() and torch.equal(y, O.feature_stride(x))
    # chunking + naming of the bulk encoder
    vp, dur = str(g["video_path"]), int(g["duration"])
    want_paths, want_lens = json.loads(str(g["chunk_paths"])), g["chunk_lens"].tolist()
    assert [c[0] for c in O.feature_cache_chunks(vp, dur)] == want_paths
    d, half = fc.output_dir(vp)
    starts = list(range(0, dur, fc.CHUNK))
    assert [os.path.join(d, fc.chunk_name(half, s)) for s in starts] == want_paths
    assert [min(s + fc.CHUNK, dur) - s for s in starts] == want_lens
    vids = [f"v{i}" for i in range(10)]
    assert fc.rank_slice(vids, 1, 4) == ["v2", "v3"] and sum(len(fc.rank_slice(vids, r, 4)) for r in range(4)) == 8   # remainder dropped, as in the reference


def test_reference_import_names_resolve_to_the_dropin():
    """SURVEY fact 0.3 / 8b last line: the reference's callers import `videollama2.*` (and the directory is `streammind/`);
    both names resolve to the SAME module objects as streammind_amd -- the import block of eval/video_score_stream_demo.py:19-38
    works unedited."""
    import streammind_amd
    from videollama2.constants import NUM_FRAMES
    from videollama2.model import Videollama2LlamaForCausalLM, Videollama2MistralForCausalLM, Videollama2MixtralForCausalLM
    from videollama2.model.builder import load_pretrained_model
    from videollama2.conversation import conv_templates, SeparatorStyle
    from videollama2.mm_utils import process_video, tokenizer_MMODAL_token, get_model_name_from_path, KeywordsStoppingCriteria
    from videollama2.constants import NUM_FRAMES, DEFAULT_MMODAL_TOKEN, DEFAULT_MMODAL_START_TOKEN, DEFAULT_MMODAL_END_TOKEN, MMODAL_TOKEN_INDEX
    from videollama2.mm_utils import tokenizer_MMODAL_token, tokenizer_image_token, expand2square, process_video, process_image
    from videollama2 import model_init, x_infer
    from videollama2 import conversation as conversation_lib
    import videollama2, streammind
    import streammind.model.builder as b2
    import streammind_amd.model.builder as b3
    assert videollama2 is streammind_amd and streammind is streammind_amd and b2 is b3
    assert load_pretrained_model is b3.load_pretrained_model and model_init is streammind_amd.model_init
    assert Videollama2MistralForCausalLM is streammind_amd.model.Videollama2MistralForCausalLM
    assert NUM_FRAMES == 8 and MMODAL_TOKEN_INDEX["VIDEO"] == -201 and "llama_2" in conv_templates
    assert get_model_name_from_path("/a/b/run1/checkpoint-500/") == "run1_checkpoint-500" and get_model_name_from_path("x/VideoLLaMA2-7B") == "VideoLLaMA2-7B"
    with pytest.raises(NotImplementedError):
        Videollama2LlamaForCausalLM()
    img = np.zeros((4, 10, 3), np.uint8) + 9
    sq = expand2square(img, (1, 2, 3))
    assert sq.shape == (10, 10, 3) and sq[0, 0].tolist() == [1, 2, 3] and sq[3, 0].tolist() == [9, 9, 9] and (sq[7:] == [1, 2, 3]).all()


def test_checkpoint_config_to_path_dims(gold):
    """config.json as the REFERENCE's model object writes it (golden g12) -> the native path's dimensions; rope_theta under
    either transformers spelling; the gate defaults are the reference's hard-wired MistralConfig(vocab_size=2, layers=4)."""
    import json
    from streammind_amd.model.builder import path_config_from_checkpoint
    g = gold("g12_checkpoint_layout")
    cfgj, vj = json.loads(str(g["config_json"])), json.loads(str(g["tower_config_json"]))
    c = path_config_from_checkpoint(cfgj, vj)
    assert (c.vit_image, c.vit_patch, c.vit_hidden, c.vit_heads, c.vit_mlp, c.vit_layers, c.vit_layers_run) == (56, 14, 128, 2, 256, 3, 2)
    assert (c.conn_d_model, c.llm_layers, c.llm_heads, c.llm_kv_heads, c.llm_mlp, c.llm_vocab) == (256, 2, 2, 1, 512, 384)
    assert c.llm_rope_theta == 1e6 and abs(c.llm_eps - 1e-5) < 1e-12
    assert (c.gate_layers, c.gate_heads, c.gate_kv_heads, c.gate_mlp, c.gate_eps) == (4, 32, 8, 14336, 1e-6)
    old = dict(cfgj); old.pop("rope_parameters"); old["rope_theta"] = 5e5
    old["mm_gate_config"] = {"num_attention_heads": 2, "num_key_value_heads": 1, "intermediate_size": 512}
    c2 = path_config_from_checkpoint(old, vj)
    assert c2.llm_rope_theta == 5e5 and (c2.gate_heads, c2.gate_kv_heads, c2.gate_mlp, c2.gate_layers) == (2, 1, 512, 4)
    with pytest.raises(ValueError, match="select feature"):
        path_config_from_checkpoint(dict(cfgj, mm_vision_select_feature="cls_patch"), vj)


def _marked_frames(n, side=336):
    return np.stack([np.full((side, side, 3), i % 256, np.uint8) for i in range(n)])


def test_read_video_stream_and_process_video_paths_vs_reference_golden(tmp_path, gold):
    """f2 host side against golden g14 (the reference's own read_video_stream / process_video path branch run with stub
    readers): the drop-in samples the same frame ids for the streaming loop and for offline clips ("uniform" / "fps", .gif
    rule), from every source the decoder adaptor serves in this image (.npy, .npz with fps, a directory of stills, .gif)."""
    from types import SimpleNamespace
    from PIL import Image
    from streammind_amd import video_io
    from streammind_amd.mm_utils import process_video
    g = gold("g14_video_sampling")
    for i, (n, fps, cur) in enumerate(g["stream_cases"]):
        n = int(n)
        p = tmp_path / f"s{i}.npz"
        np.savez(p, frames=_marked_frames(n, 8), fps=fps)
        ids, vr = video_io.read_video_stream(str(p), cur)
        assert ids.tolist() == g[f"stream_ids{i}"].tolist() and len(vr) == n and vr.get_avg_fps() == fps
        if len(ids):
            assert int(vr[ids[-1]].asnumpy()[0, 0, 0]) == int(ids[-1]) % 256
    got = [(fid, int(fr[0, 0, 0])) for fid, fr in video_io.stream_frames(str(tmp_path / "s3.npz"), 30)]
    assert [a for a, _ in got] == g["stream_ids3"].tolist() and all(a % 256 == b for a, b in got)
    proc = SimpleNamespace(crop_size={"height": 336, "width": 336}, image_mean=[0.5, 0.5, 0.5])
    for i, (n, fps, nf, sc) in enumerate(g["clip_cases"]):
        n, nf = int(n), int(nf)
        d = tmp_path / f"clip{i}"
        d.mkdir()
        (d / "fps.txt").write_text(str(fps))
        for j in range(n):
            Image.fromarray(np.full((336, 336, 3), j, np.uint8)).save(d / f"f{j:05d}.png")
        out = process_video(str(d), proc, aspect_ratio=None, num_frames=nf, sample_scheme="uniform" if sc == 0 else "fps")
        assert out.dtype == torch.uint8 and out[:, 0, 0, 0].tolist() == g[f"clip_ids{i}"].tolist()
    gif = tmp_path / "a.gif"
    fr = [Image.fromarray(np.full((336, 336, 3), (j, 255 - j, j), np.uint8)) for j in range(int(g["gif_n"]))]
    fr[0].save(gif, save_all=True, append_images=fr[1:], duration=100, loop=0)
    assert len(video_io.open_video(str(gif))) == int(g["gif_n"]) and video_io.open_video(str(gif)).get_avg_fps() == 10.0
    from oracle import streammind_oracle as O
    for i, (nf, sc) in enumerate(g["gif_cases"]):
        out = process_video(str(gif), proc, aspect_ratio=None, num_frames=int(nf), sample_scheme="uniform" if sc == 0 else "fps")
        want = g[f"gif_ids{i}"].tolist()
        assert out.shape[0] == len(want) == len(O.clip_frame_indices(int(g["gif_n"]), 10, int(nf), "uniform" if sc == 0 else "fps", gif=True))
    npy = tmp_path / "v.npy"
    np.save(npy, _marked_frames(40))
    assert process_video(str(npy), proc, num_frames=8)[:, 0, 0, 0].tolist() == O.clip_frame_indices(40, 30, 8)
    (tmp_path / "movie.mp4").write_bytes(b"not a video")
    with pytest.raises(ImportError, match="no decoder"):
        video_io.open_video(str(tmp_path / "movie.mp4"))


def _write_mjpeg_avi(path, jpegs, w, h, rate, scale, handler=b"MJPG"):
    """a minimal RIFF AVI: hdrl (avih + one vids stream) and a movi list of 00dc chunks, word aligned, with an idx1 behind it"""
    import struct

    def chunk(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    def lst(kind, body):
        return b"LIST" + struct.pack("<I", len(body) + 4) + kind + body

    avih = struct.pack("<14I", int(1e6 * scale / rate), 0, 0, 0x10, len(jpegs), 0, 1, 0, w, h, 0, 0, 0, 0)
    strh = b"vids" + handler + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, scale, rate, 0, len(jpegs), 0, 0, 0) + struct.pack("<4h", 0, 0, w, h)
    strf = struct.pack("<IiiHH", 40, w, h, 1, 24) + handler + struct.pack("<IiiII", w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi = lst(b"movi", b"".join(chunk(b"00dc", j) for j in jpegs))
    idx1 = chunk(b"idx1", b"".join(struct.pack("<4sIII", b"00dc", 0x10, 0, len(j)) for j in jpegs))
    body = b"AVI " + hdrl + movi + idx1
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_mjpeg_avi_is_read_without_a_codec_library(tmp_path):
    """f2: a Motion-JPEG AVI (the one real container this image can decode: RIFF walk here, baseline JPEG through PIL) behind
    the decord surface the streaming loop uses -- frame count, fps = dwRate / dwScale, random access, and the sampler on top;
    frames written WITHOUT Huffman tables (the AVI1 convention) decode to the same pixels as with them; an AVI holding
    another codec falls through to the installed decoders (none here: ImportError naming the reason)."""
    import io
    from PIL import Image
    from streammind_amd import video_io
    n, w, h = 13, 48, 32
    rng = np.random.default_rng(5)
    base = np.kron(rng.integers(0, 255, (4, 6, 3)), np.ones((8, 8, 1))).astype(np.uint8)          # blocky: survives JPEG closely
    frames, jpegs, bare = [], [], []
    for j in range(n):
        fr = np.clip(base.astype(int) + 9 * j, 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(fr).save(buf, "JPEG", quality=95, optimize=False, subsampling=0)
        b = buf.getvalue()
        frames.append(fr); jpegs.append(b)
        out, pos = b[:2], 2                                   # the same frame with its DHT segments stripped
        while b[pos + 1] != 0xDA:
            seg = 2 + int.from_bytes(b[pos + 2:pos + 4], "big")
            if b[pos + 1] != 0xC4:
                out += b[pos:pos + seg]
            pos += seg
        bare.append(out + b[pos:])
        assert b"\xff\xc4" not in bare[-1][:bare[-1].find(b"\xff\xda")]
    _write_mjpeg_avi(tmp_path / "cam.avi", jpegs, w, h, 30000, 1001)
    _write_mjpeg_avi(tmp_path / "cam_bare.avi", bare, w, h, 25, 1)
    vr = video_io.open_video(str(tmp_path / "cam.avi"))
    assert isinstance(vr, video_io.MjpegAviVideo) and len(vr) == n and abs(vr.get_avg_fps() - 30000 / 1001) < 1e-9
    ref = [np.asarray(Image.open(io.BytesIO(b)).convert("RGB")) for b in jpegs]
    for j in (0, 7, n - 1, 3):
        got = vr[j].asnumpy()
        assert got.shape == (h, w, 3) and np.array_equal(got, ref[j]) and np.abs(got.astype(int) - frames[j]).max() <= 6
    assert np.array_equal(vr.get_batch([1, 5, 9]).asnumpy(), np.stack([ref[1], ref[5], ref[9]]))
    vb = video_io.open_video(str(tmp_path / "cam_bare.avi"))
    assert len(vb) == n and vb.get_avg_fps() == 25.0
    for j in range(n):
        assert np.array_equal(vb[j].asnumpy(), ref[j])
    ids, _ = video_io.read_video_stream(str(tmp_path / "cam_bare.avi"), 5)
    assert ids.tolist() == list(range(0, n - 1, 5))
    assert [fid for fid, _ in video_io.stream_frames(str(tmp_path / "cam_bare.avi"), 5)] == ids.tolist()
    _write_mjpeg_avi(tmp_path / "h264.avi", jpegs, w, h, 25, 1, handler=b"H264")
    with pytest.raises(ImportError, match="not Motion-JPEG"):
        video_io.open_video(str(tmp_path / "h264.avi"))


def test_mjpeg_avi_opendml_segments_and_dropped_frames(tmp_path):
    """OpenDML AVI (what ffmpeg / capture tools write past ~1 GiB): frames continue in top-level `RIFF....AVIX` segments behind the
    first `RIFF....AVI `; a zero-length `00dc` chunk is a dropped frame and repeats its predecessor, so frame index keeps tracking
    time.  Every frame of every segment is returned, in order."""
    import io
    import struct
    from PIL import Image
    from streammind_amd import video_io
    w, h = 32, 16
    jpegs = []
    for j in range(7):
        buf = io.BytesIO()
        Image.fromarray(np.full((h, w, 3), 20 + 30 * j, np.uint8)).save(buf, "JPEG", quality=90)
        jpegs.append(buf.getvalue())
    _write_mjpeg_avi(tmp_path / "seg.avi", jpegs[:3], w, h, 30, 1)

    def chunk(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    def avix(js):
        movi = b"movi" + b"".join(chunk(b"00dc", j) for j in js)
        body = b"AVIX" + b"LIST" + struct.pack("<I", len(movi)) + movi
        return b"RIFF" + struct.pack("<I", len(body)) + body

    with open(tmp_path / "seg.avi", "ab") as f:
        f.write(avix([jpegs[3], b"", jpegs[4]]))              # a dropped frame inside the second segment
        f.write(avix(jpegs[5:]))
    vr = video_io.open_video(str(tmp_path / "seg.avi"))
    assert isinstance(vr, video_io.MjpegAviVideo) and len(vr) == 8
    order = [0, 1, 2, 3, 3, 4, 5, 6]
    for i, j in enumerate(order):
        assert np.array_equal(vr[i].asnumpy(), np.asarray(Image.open(io.BytesIO(jpegs[j])).convert("RGB"))), i


def test_loader_rejects_configs_the_kernels_do_not_implement():
    """path_config_from_checkpoint: a config.json that asks for something the kernels hard-wire differently (activation, tied
    embeddings, projection biases, rope scaling, a foreign head_dim) must raise, not load and compute something else."""
    from streammind_amd.model.builder import path_config_from_checkpoint
    lm = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512, vocab_size=384,
              rms_norm_eps=1e-5, rope_theta=1e6, mm_gate_config=dict(num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512))
    vis = dict(image_size=56, patch_size=14, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=3)
    cfg = path_config_from_checkpoint(dict(lm, mm_hidden_size=128), vis)
    assert cfg.llm_layers == 2 and cfg.vit_hidden == 128 and cfg.llm_rope_theta == 1e6
    assert path_config_from_checkpoint(dict(lm, rope_theta=None, rope_parameters=dict(rope_theta=5e5, rope_type="default")), vis).llm_rope_theta == 5e5
    for bad_lm, bad_vis in ((dict(hidden_act="gelu"), {}), (dict(tie_word_embeddings=True), {}), (dict(attention_bias=True), {}),
                            (dict(rope_scaling=dict(type="linear", factor=2.0)), {}), (dict(rope_parameters=dict(rope_theta=1e6, rope_type="yarn", factor=4.0)), {}),
                            ({}, dict(hidden_act="gelu"))):
        with pytest.raises(NotImplementedError):
            path_config_from_checkpoint(dict(lm, **bad_lm), dict(vis, **bad_vis))
    with pytest.raises(ValueError, match="head_dim"):
        path_config_from_checkpoint(dict(lm, head_dim=128), vis)
    with pytest.raises(ValueError, match="select feature"):
        path_config_from_checkpoint(dict(lm, mm_vision_select_feature="cls_patch"), vis)


def test_video_test_stream_demo_main_loop(tmp_path, monkeypatch):
    """eval/video_test_stream_demo.py:147-182 (`run_inference_time_metric`): ONE video, the fixed instruction, every sampled frame handed
    to `infer` alone with the carried prompt, and the reference's printed line (mm:ss from frame_id // 25, as upstream) for every frame the
    gate fired on.  `infer` is a stub here (the tick itself is pinned by golden g6 on the GPU); what is checked is the loop: which frames
    are fed (read_video_stream at cur_fps, last frame excluded), the argument set of each tick, prompt carry-over, the line format."""
    import types
    from streammind_amd.eval import video_test_stream_demo as demo
    frames = np.zeros((200, 8, 8, 3), np.uint8)
    frames[:, 0, 0, 0] = np.arange(200) % 256
    np.savez(tmp_path / "v.npz", frames=frames, fps=np.float64(30.0))
    calls = []

    def fake_infer(model, video, instruct, tokenizer, do_sample=False, version="mistral_instruct", score_video=None, prompt=None, **kw):
        calls.append(dict(first=int(video[0]), instruct=instruct, do_sample=do_sample, version=version, score_video=score_video, prompt=prompt, kw=kw))
        fired = len(calls) % 3 == 0
        return ("reply %d" % len(calls) if fired else None), (prompt or "P") + ("+" if fired else "")

    monkeypatch.setattr(demo, "infer", fake_infer)
    processor = lambda imgs, num_frames: [int(np.asarray(imgs[0])[0, 0, 0])]
    args = types.SimpleNamespace(model=("M", processor, "T", "llama_2"), video_path=str(tmp_path / "v.npz"), cur_fps=2, model_path=None, model_base=None, model_name=None)
    lines = []
    out = demo.run_inference_time_metric(args, on_reply=lines.append)
    want_ids = list(range(0, 199, 15))                       # 30 fps source at 2 fps: every 15th frame, the LAST frame (199) never sampled
    assert [c["first"] for c in calls] == want_ids
    assert all(c["instruct"] == demo.INSTRUCT and c["do_sample"] is False and c["version"] == "llama_2" and c["score_video"] is True and c["kw"] == {} for c in calls)
    assert calls[0]["prompt"] is None and calls[1]["prompt"] == "P" and calls[3]["prompt"] == "P+"      # the prompt is carried and grows on a fire
    fired = [want_ids[i] for i in range(len(want_ids)) if (i + 1) % 3 == 0]
    assert [f for f, _ in out] == fired and lines == [l for _, l in out]
    f = fired[-1]
    assert out[-1][1] == "The content of the video until {}:{}  is: reply {}:".format(f // 25 // 60, f // 25 % 60, want_ids.index(f) + 1)
    # the reference's filter (:163) with its never-advanced cur_min / cur_sec = -1 passes every frame; advanced, it skips the past
    assert demo._frame_passes(0, -1, -1) and demo._frame_passes(1799, -1, -1)
    assert not demo._frame_passes(30 * 59, 0, 59) and demo._frame_passes(30 * 60, 0, 59) and demo._frame_passes(30 * 30, 0, 29) and not demo._frame_passes(30 * 29, 0, 29)

"""The weights-in boundary (streammind/model/builder.py:30-210) on a real MI355X: checkpoint directories laid
out as the REFERENCE's own objects write them (golden g12: names, shapes, dtypes and config files recorded from the reference
model's `save_pretrained` and HF `CLIPVisionModel.save_pretrained`; the weights are regenerated from the goldens' seeds), loaded
through `from videollama2 import model_init`, then driven through the reference's streaming loop and compared with golden g6
(the reference's own stream_generate_demo trace on the same weights)."""
import json
import os
import shutil

import pytest
import torch

from oracle import streammind_oracle as O
from oracle.make_golden import TINY_V as TV, TINY_C as TC, TINY_G as TG, TINY_L as TL
from tests.util_models import check_stream_against_g6, conn_gate_weights

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _weights():
    return O.make_vit_weights(TV, 41), conn_gate_weights(TC, TG, 86), O.make_lm_weights(TL, 44)


def _tensor(k, shape, Wv, Wc, Wl):
    if "vision_tower.vision_tower." in k:
        t = Wv[k.split("vision_tower.vision_tower.")[1]]
    elif k.startswith("model.mm_projector."):
        t = Wc[k[len("model.mm_projector."):]]
    else:
        t = Wl[k]
    return t.reshape(shape).to(torch.float16).contiguous()          # fp16 on disk: builder.py:54


def _write_tokenizer(d):
    shutil.copy(os.path.join(GOLD, "tiny_tokenizer", "tokenizer.json"), os.path.join(d, "tokenizer.json"))
    json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>",
               "model_max_length": 2048, "padding_side": "right"}, open(os.path.join(d, "tokenizer_config.json"), "w"))


def _write_checkpoint(tmp_path, g, layout):
    from safetensors.torch import save_file
    Wv, Wc, Wl = _weights()
    ck, tower = tmp_path / "StreamMind-tiny", tmp_path / "clip-tiny"
    ck.mkdir(); tower.mkdir()
    cfg = json.loads(str(g["config_json"]))
    cfg["mm_vision_tower"] = str(tower)
    cfg["_name_or_path"] = "mistralai/Mistral-7B-Instruct-v0.2"
    # the reference hard-wires the gate (MistralConfig defaults, 4096 wide); the goldens shrank it by patching that constructor
    # (oracle/make_golden.py _ref_stream_model) -- the checkpoint states the shrunken gate through the build's own key
    cfg["mm_gate_config"] = {"num_attention_heads": TG.heads, "num_key_value_heads": TG.kv_heads, "intermediate_size": TG.mlp}
    lm_keys = json.loads(str(g["lm_keys"]))
    tensors = {k: _tensor(k, shp, Wv, Wc, Wl) for k, (shp, _) in lm_keys.items()}
    assert all(dt == "float16" for _, dt in lm_keys.values())
    vcfg = json.loads(str(g["tower_config_json"]))
    if layout == "as_saved_by_the_reference":
        # ONE model.safetensors with LM + projector + tower, exactly the reference object's save_pretrained (transformers 5 names)
        save_file(tensors, str(ck / "model.safetensors"))
        json.dump(vcfg, open(tower / "config.json", "w"))
    else:
        # transformers-4.44-era layout: sharded LM safetensors + index WITHOUT tower and projector, projector in mm_projector.bin,
        # rope_theta at top level; the tower directory is a FULL CLIP checkpoint (pytorch_model.bin with vision_model.*, text_model.*,
        # visual_projection, text_projection, logit_scale; config.json with a nested vision_config)
        cfg["rope_theta"] = cfg.pop("rope_parameters")["rope_theta"]
        cfg["sliding_window"] = 4096
        lm = {k: v for k, v in tensors.items() if "vision_tower" not in k and "mm_projector" not in k}
        names = sorted(lm)
        shards = {"model-00001-of-00002.safetensors": names[:len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
        for f, ks in shards.items():
            save_file({k: lm[k] for k in ks}, str(ck / f))
        json.dump({"metadata": {}, "weight_map": {k: f for f, ks in shards.items() for k in ks}}, open(ck / "model.safetensors.index.json", "w"))
        torch.save({k: v for k, v in tensors.items() if "mm_projector" in k}, str(ck / "mm_projector.bin"))
        full = {"vision_model." + k.split("vision_tower.vision_tower.")[1]: v.float() for k, v in tensors.items() if "vision_tower" in k}
        full.update({"text_model.embeddings.token_embedding.weight": torch.zeros(8, 16), "visual_projection.weight": torch.zeros(16, TV.hidden),
                     "text_projection.weight": torch.zeros(16, 16), "logit_scale": torch.tensor(2.6)})
        torch.save(full, str(tower / "pytorch_model.bin"))
        json.dump({"model_type": "clip", "projection_dim": 16, "text_config": {}, "vision_config": vcfg}, open(tower / "config.json", "w"))
    json.dump(cfg, open(ck / "config.json", "w"))
    open(tower / "preprocessor_config.json", "w").write(str(g["preprocessor_config_json"]))
    _write_tokenizer(str(ck))
    return str(ck), (Wv, Wc, Wl)


@pytest.mark.parametrize("layout", ["as_saved_by_the_reference", "hf444_sharded_projector_bin_full_clip_tower"])
def test_model_init_from_checkpoint_then_reference_stream_loop(tmp_path, gold, layout):
    from videollama2 import model_init                 # the reference's import name
    ck, (Wv, Wc, Wl) = _write_checkpoint(tmp_path, gold("g12_checkpoint_layout"), layout)
    model, processor, tokenizer, version = model_init(ck, "VideoLLaMA2-7B")
    assert version == "llama_2" and tokenizer.pad_token == tokenizer.unk_token
    assert model.config.mm_projector_type == "mamba" and model.native.cfg.vit_layers_run == TV.layers - 1
    assert sorted(model.native.ignored) != [] or layout.startswith("as_saved")      # q/k of the gate, post_layernorm: never read
    if layout.startswith("hf444"):
        assert model.max_seq == 4096 and model.native.cfg.llm_sliding_window == 4096      # the window is the attention kernels' mask now, not a capacity cap
    # resize_token_embeddings (builder.py:186-196): <im_patch> was added to the tokenizer and the native table grew with it -- the id indexes
    # a real (freshly initialised) row, as in the reference, instead of being refused
    pid = tokenizer.convert_tokens_to_ids("<im_patch>")
    assert pid == len(tokenizer) - 1 and model.native.cfg.llm_vocab == len(tokenizer) == json.load(open(os.path.join(ck, "config.json")))["vocab_size"] + 1
    model._check_ids([1, pid, 5])

    def to_video(frame_u8):                           # eval/video_score_stream_demo.py:285-287: processor([img], num_frames=1)
        return processor([frame_u8[0].numpy()], num_frames=1)
    check_stream_against_g6(model, tokenizer, gold("g6_stream_tiny"), Wv, Wc, Wl, (TV, TC, TG, TL), to_video)


def test_loader_errors(tmp_path, gold):
    from streammind_amd.model.builder import load_pretrained_model
    g = gold("g12_checkpoint_layout")
    ck, _ = _write_checkpoint(tmp_path, g, "as_saved_by_the_reference")
    cfgp = os.path.join(ck, "config.json")
    cfg = json.load(open(cfgp))
    with pytest.raises(NotImplementedError):
        load_pretrained_model(ck, None, "VideoLLaMA2-7B", load_4bit=True)
    with pytest.raises(NotImplementedError):
        load_pretrained_model(ck, "base", "VideoLLaMA2-7B")
    json.dump(dict(cfg, mm_projector_type="identity"), open(cfgp, "w"))             # no reader in temporal_aggregator upstream either
    with pytest.raises(ValueError, match="Unsupported projector type"):
        load_pretrained_model(ck, None, "VideoLLaMA2-7B")
    json.dump(dict(cfg, mm_projector_type="mlp2x_gelu"), open(cfgp, "w"))
    with pytest.raises(KeyError, match="projector: missing tensors"):
        load_pretrained_model(ck, None, "VideoLLaMA2-7B")
    json.dump(dict(cfg, mm_projector_type="stc_connector"), open(cfgp, "w"))       # a Mamba checkpoint labelled STC: the connector's tensors are not there
    with pytest.raises(KeyError, match="STC connector: missing tensors"):
        load_pretrained_model(ck, None, "VideoLLaMA2-7B")
    json.dump(dict(cfg, mm_vision_tower="openai/clip-vit-large-patch14-336"), open(cfgp, "w"))
    with pytest.raises(FileNotFoundError, match="not a local directory"):
        load_pretrained_model(ck, None, "VideoLLaMA2-7B")
    json.dump(dict(cfg, sliding_window=128), open(cfgp, "w"))      # a KV capacity beyond the window is legal: the kernels mask the window
    _, mw, _, _ = load_pretrained_model(ck, None, "VideoLLaMA2-7B", max_seq=256)
    assert mw.max_seq == 256 and mw.native.cfg.llm_sliding_window == 128
    del mw
    # a checkpoint that lost one half of a fused pair must not finalize -- even when the other half arrives twice (here:
    # gate_proj of the gate is in model.safetensors AND in mm_projector.bin, its up_proj nowhere)
    from safetensors.torch import load_file, save_file
    json.dump(cfg, open(cfgp, "w"))
    sd = load_file(os.path.join(ck, "model.safetensors"))
    sd.pop("model.mm_projector.cls_net.cls_model.model.layers.1.mlp.up_proj.weight")
    save_file(sd, os.path.join(ck, "model.safetensors"))
    k = "model.mm_projector.cls_net.cls_model.model.layers.1.mlp.gate_proj.weight"
    torch.save({k: sd[k]}, os.path.join(ck, "mm_projector.bin"))
    with pytest.raises(ValueError, match="incomplete.*layers.1.gu"):
        load_pretrained_model(ck, None, "VideoLLaMA2-7B")


def test_stock_videollama2_checkpoint_with_stc_connector_loads_and_generates(tmp_path, gold, tiny_tokenizer):
    """SURVEY 8f row f4, the tail: a checkpoint laid out as the reference saves it but with `mm_projector_type: stc_connector`
    (builder.py:139-146: the stock VideoLLaMA2 projector, timm state-dict names, an unused cls_net riding along).  The loader
    builds the native model WITHOUT the Mamba connector and gate (sm_config_t.conn_d_state = 0), hands the projector's tensors
    to the host-side STCConnector, and `model.generate(ids, images_or_videos=[clip], modal_list=["video"])` runs tower -> STC ->
    splice of all 27 tokens -> greedy decode (videollama2_arch.py:113-133,303-309).  Against the oracle: spliced tokens at the
    connector's bf16 noise level (tests/test_gpu_stc.py), prefill logits, ids wherever the margin allows; the streaming calls
    refuse (no event gate in such a checkpoint)."""
    from safetensors.torch import load_file, save_file
    from streammind_amd.model.stc_connector import STCConnector
    ck, (Wv, Wc, Wl) = _write_checkpoint(tmp_path, gold("g12_checkpoint_layout"), "as_saved_by_the_reference")
    scfg = O.StcCfg(mm_hidden=TV.hidden, hidden=TL.hidden, depth=4)
    Ws = O.make_stc_weights(scfg, 77)
    t = {k: v for k, v in load_file(os.path.join(ck, "model.safetensors")).items() if "mm_projector." not in k}
    t.update({"model.mm_projector." + k: v.to(torch.float16).contiguous() for k, v in Ws.items()})
    t["model.mm_projector.cls_net.cls_model.lm_head.weight"] = torch.zeros(2, TL.hidden, dtype=torch.float16)
    save_file(t, os.path.join(ck, "model.safetensors"))
    cfg = json.load(open(os.path.join(ck, "config.json")))
    cfg["mm_projector_type"] = "stc_connector"
    cfg.pop("mm_gate_config", None)
    json.dump(cfg, open(os.path.join(ck, "config.json"), "w"))
    from videollama2.model.builder import load_pretrained_model
    tokenizer, model, processor, _ = load_pretrained_model(ck, None, "VideoLLaMA2-7B", torch_dtype=torch.bfloat16)      # bf16 operands: the oracle's mode
    assert isinstance(model.mm_projector, STCConnector) and model.native.cfg.conn_d_state == 0 and model.native.cfg.gate_layers == 0
    frames = O.synthetic_frames(4, TV.image_size, seed=31, scene_len=2)
    pix = O.preprocess_frames(frames, TV.image_size)
    ids = [1, 5, 9, -201, 11, 12, 13]
    out = model.generate(torch.tensor([ids]), images_or_videos=[pix], modal_list=["video"], do_sample=False, max_new_tokens=6)
    toks = model.stream.tokens().cpu()
    assert toks.shape == (27, TL.hidden)                                     # (4 + 2 - 2) // 2 + 1 = 3 frames x 3 x 3
    # oracle: tower (bf16 mode) -> STC -> splice -> greedy decode
    Ws16 = {k: v.to(torch.float16).float() for k, v in Ws.items()}
    feats = O.vit_features(pix, Wv, TV, O.MIXED)
    ref_tok = O.stc_forward(O.bf16_round(feats)[None], Ws16, scfg, O.MIXED)[0]
    noise = ((ref_tok - O.stc_forward(O.bf16_round(feats)[None], Ws16, scfg, O.FP32)[0]).abs().max() / ref_tok.abs().max()).item()
    err = ((toks - ref_tok).abs().max() / ref_tok.abs().max()).item()
    print(f"stc tokens through the loader: {err:.2e} of the largest token (oracle bf16 vs fp32: {noise:.2e})")
    assert err < 3 * noise + 1e-3
    table = Wl["model.embed_tokens.weight"]
    emb = torch.cat([table[[1, 5, 9]], toks, table[[11, 12, 13]]])           # the GPU's own tokens: isolates the LLM side of the comparison
    ref_ids, trace = O.greedy_generate(emb, Wl, TL, 6, eos_token_id=tokenizer.eos_token_id, prec=O.MIXED, return_logits=True)
    got = out[0].tolist()
    for j, (a, b) in enumerate(zip(got, ref_ids)):
        if a != b:
            assert float(torch.topk(trace[j], 2).values.diff().abs()) < 6e-2, (j, got, ref_ids)
            break
    # u8 frames and pre-extracted tower features take the same route
    out_u8 = model.generate(torch.tensor([ids]), images_or_videos=[frames], modal_list=["video"], do_sample=False, max_new_tokens=6)
    assert model.stream.tokens().shape == (27, TL.hidden) and out_u8.shape[0] == 1
    with pytest.raises(NotImplementedError, match="event gate"):
        model.stream_generate_demo(torch.tensor([ids]), images_or_videos=frames[:1], modal_list=["video"], tokenizer=tokenizer)
    with pytest.raises(Exception, match="without the Mamba connector"):
        model.stream.push_frames(frames[:1].cuda())


def test_stock_videollama2_checkpoint_with_mlp2x_gelu_projector(tmp_path, gold):
    """the pooled projector of plain VideoLLaMA2 checkpoints (`mm_projector_type: mlp2x_gelu`, builder.py:121-128; frame mean in
    temporal_aggregator, videollama2_arch.py:293-294) through the same loader: nn.Sequential names ("0.weight", "2.weight") in the
    checkpoint, 16 tokens (one per patch position) spliced for a 4-frame clip.  Tokens 2e-3 from the oracle on the oracle's
    bf16-mode tower features chain (tower differences included: 1e-2 budget), ids as in the STC test."""
    from safetensors.torch import load_file, save_file
    from streammind_amd.model.stc_connector import MlpGeluProjector
    from videollama2.model.builder import load_pretrained_model
    ck, (Wv, Wc, Wl) = _write_checkpoint(tmp_path, gold("g12_checkpoint_layout"), "as_saved_by_the_reference")
    Wp = O.make_mlp_projector_weights(TV.hidden, TL.hidden, 2, 79)
    t = {k: v for k, v in load_file(os.path.join(ck, "model.safetensors")).items() if "mm_projector." not in k}
    t.update({"model.mm_projector." + k: v.to(torch.float16).contiguous() for k, v in Wp.items()})
    save_file(t, os.path.join(ck, "model.safetensors"))
    cfg = json.load(open(os.path.join(ck, "config.json")))
    cfg["mm_projector_type"] = "mlp2x_gelu"
    cfg.pop("mm_gate_config", None)
    json.dump(cfg, open(os.path.join(ck, "config.json"), "w"))
    tokenizer, model, processor, _ = load_pretrained_model(ck, None, "VideoLLaMA2-7B", torch_dtype=torch.bfloat16)
    assert isinstance(model.mm_projector, MlpGeluProjector) and model.native.cfg.conn_d_state == 0
    frames = O.synthetic_frames(4, TV.image_size, seed=33, scene_len=2)
    pix = O.preprocess_frames(frames, TV.image_size)
    ids = [1, 5, 9, -201, 11, 12]
    out = model.generate(torch.tensor([ids]), images_or_videos=[pix], modal_list=["video"], do_sample=False, max_new_tokens=5)
    toks = model.stream.tokens().cpu()
    assert toks.shape == (16, TL.hidden)
    feats = O.vit_features(pix, Wv, TV, O.MIXED)
    ref_tok = O.mlp_projector_forward(O.bf16_round(feats)[None], Wp, 2, True, O.MIXED)[0]
    err = ((toks - ref_tok).abs().max() / ref_tok.abs().max()).item()
    print(f"mlp2x_gelu tokens through the loader: {err:.2e} of the largest token")
    assert err < 1e-2
    table = Wl["model.embed_tokens.weight"]
    emb = torch.cat([table[[1, 5, 9]], toks, table[[11, 12]]])
    ref_ids, trace = O.greedy_generate(emb, Wl, TL, 5, eos_token_id=tokenizer.eos_token_id, prec=O.MIXED, return_logits=True)
    for j, (a, b) in enumerate(zip(out[0].tolist(), ref_ids)):
        if a != b:
            assert float(torch.topk(trace[j], 2).values.diff().abs()) < 6e-2, (j, out[0].tolist(), ref_ids)
            break

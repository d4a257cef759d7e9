"""Test helpers: build a NativeModel (GPU) and the matching oracle weight dicts from the same seeds."""
import torch

from oracle import streammind_oracle as O


def path_config(vcfg: O.VitCfg, ccfg: O.ConnCfg, gcfg: O.LmCfg, lcfg=None, max_frames_per_call=8, precise=True, weights_fp8=False, vit_fp16=False, llm_fp16=False, proj_fp16=False):
    from streammind_amd.native import PathConfig
    kw = dict(vit_image=vcfg.image_size, vit_patch=vcfg.patch, vit_hidden=vcfg.hidden, vit_heads=vcfg.heads,
              vit_mlp=vcfg.mlp, vit_layers=vcfg.layers, vit_select_layer=vcfg.select_layer, vit_eps=vcfg.eps,
              conn_d_model=ccfg.d_model, conn_d_state=ccfg.d_state, conn_d_conv=ccfg.d_conv, conn_expand=ccfg.expand,
              conn_eps=ccfg.ln_eps, gate_layers=gcfg.layers, gate_heads=gcfg.heads, gate_kv_heads=gcfg.kv_heads,
              gate_mlp=gcfg.mlp, gate_eps=gcfg.eps, max_frames_per_call=max_frames_per_call, gate_precise=precise,
              weights_fp8=weights_fp8, vit_fp16=vit_fp16, llm_fp16=llm_fp16, proj_fp16=proj_fp16)
    if lcfg is None:
        kw.update(llm_layers=0)
    else:
        kw.update(llm_layers=lcfg.layers, llm_heads=lcfg.heads, llm_kv_heads=lcfg.kv_heads, llm_mlp=lcfg.mlp,
                  llm_vocab=lcfg.vocab, llm_eps=lcfg.eps, llm_rope_theta=lcfg.rope_theta, llm_sliding_window=int(lcfg.sliding_window or 0))
    return PathConfig(**kw)


def build_native(vcfg, ccfg, gcfg, Wv, Wc, lcfg=None, Wl=None, **kw):
    """Load the oracle's seeded weights into a NativeModel under the REFERENCE's checkpoint names."""
    from streammind_amd.native import NativeModel
    m = NativeModel(path_config(vcfg, ccfg, gcfg, lcfg, **kw))
    for k, v in Wv.items():
        m.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v.to(torch.bfloat16) if v.dim() >= 2 else v)
    for k, v in Wc.items():
        m.load_tensor("model.mm_projector." + k, v.to(torch.bfloat16) if v.dim() >= 2 and "conv1d" not in k and "A_log" not in k else v)
    if Wl is not None:
        for k, v in Wl.items():
            m.load_tensor(k, v.to(torch.bfloat16) if v.dim() >= 2 else v)
    assert m.missing() == [], m.missing()
    m.finalize()
    return m


def conn_gate_weights(ccfg, gcfg, seed):
    Wc = O.make_conn_weights(ccfg, seed)
    Wc.update(O.make_lm_weights(gcfg, seed + 1, prefix="cls_net.cls_model."))
    return Wc


def fp8_view(W: dict, skip=("embed_tokens",)) -> dict:
    """the weights an fp8-mode model effectively uses: every 2-D linear weight replaced by its fp8 dequantisation."""
    out = {}
    for k, v in W.items():
        if v.dim() == 2 and not any(s in k for s in skip) and ("proj.weight" in k or "lm_head" in k):
            out[k] = O.fp8_quantize_rows(v)[0]
        else:
            out[k] = v
    return out


LOGIT_TOL = 3e-2          # bf16-activation budget on the tiny LLM's O(1) logits (stated in test_llm_tiny_prefill_decode)


def check_stream_against_g6(model, tokenizer, g, Wv, Wc, Wl, cfgs, to_video=lambda fr: fr, gate_tol=5e-3, logit_tol=LOGIT_TOL):
    """Drive `streammind_amd.stream_infer` frame by frame exactly like eval/video_score_stream_demo.py:283-299 and compare with golden
    g6 (the reference's own stream_generate_demo trace): gate logits of every frame (5e-3: bf16 ViT in front), decisions, fire
    positions, and -- with the reference's prompt teacher-forced after every fire -- the generated ids, which must equal the
    reference's wherever the oracle's top-2 logit margin exceeds TWICE the stated logit tolerance (2 x 3e-2)."""
    import streammind_amd
    TV, TC, TG, TL = cfgs
    n = int(g["n_frames"])
    frames = O.synthetic_frames(n, TV.image_size, seed=int(g["seed_frames"]), scene_len=int(g["scene_len"]))
    prompt, fires = None, 0
    st = O.StreamOracleState()
    for i in range(n):
        golden_prompt_before = prompt
        text, prompt = streammind_amd.stream_infer(model, to_video(frames[i:i + 1]), "", tokenizer, prompt=prompt, max_new_tokens=int(g["max_new"]))
        d = (model.last_gate_logits.float().cpu() - torch.as_tensor(g["gate_logits"][i])).abs().max().item()
        assert d < gate_tol, (i, d)
        pred = int(g["preds"][i])
        assert (text is not None) == bool(pred), (i, text, pred)
        r = O.stream_frame(frames[i], st, Wv, Wc, Wl, TV, TC, TG, TL, tokenizer, max_new_tokens=int(g["max_new"]))
        if pred:
            want = g[f"new_ids{fires}"].tolist()
            assert r.new_ids == want
            ids_in = O.tokenize_with_video(golden_prompt_before or O.initial_prompt(), tokenizer)
            emb = O.splice_embeds(ids_in, O.connector_scan(O.pool_patches(st.feats), Wc, TC), st.interval_ids, Wl["model.embed_tokens.weight"])
            _, trace = O.greedy_generate(emb, Wl, TL, len(want), tokenizer.eos_token_id, return_logits=True)
            for j, (a, b) in enumerate(zip(model.last_new_ids, want)):
                margin = float(torch.topk(trace[j], 2).values.diff().abs())
                if a != b:
                    assert margin < 2 * logit_tol, (i, j, model.last_new_ids, want, margin)
                    break
            fires += 1
            prompt = st.prompt                       # teacher-force the reference's prompt for the next ticks
    assert model.interval_id_list == g["interval_ids"].tolist()
    assert fires == int(g["n_fires"])
    assert st.prompt == str(g["final_prompt"])

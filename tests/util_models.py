"""Test helpers: build a NativeModel (GPU) and the matching oracle weight dicts from the same seeds."""
import torch

from oracle import streammind_oracle as O


def path_config(vcfg: O.VitCfg, ccfg: O.ConnCfg, gcfg: O.LmCfg, lcfg=None, max_frames_per_call=8, precise=True, weights_fp8=False):
    from streammind_amd.native import PathConfig
    kw = dict(vit_image=vcfg.image_size, vit_patch=vcfg.patch, vit_hidden=vcfg.hidden, vit_heads=vcfg.heads,
              vit_mlp=vcfg.mlp, vit_layers=vcfg.layers, vit_select_layer=vcfg.select_layer, vit_eps=vcfg.eps,
              conn_d_model=ccfg.d_model, conn_d_state=ccfg.d_state, conn_d_conv=ccfg.d_conv, conn_expand=ccfg.expand,
              conn_eps=ccfg.ln_eps, gate_layers=gcfg.layers, gate_heads=gcfg.heads, gate_kv_heads=gcfg.kv_heads,
              gate_mlp=gcfg.mlp, gate_eps=gcfg.eps, max_frames_per_call=max_frames_per_call, gate_precise=precise,
              weights_fp8=weights_fp8)
    if lcfg is None:
        kw.update(llm_layers=0)
    else:
        kw.update(llm_layers=lcfg.layers, llm_heads=lcfg.heads, llm_kv_heads=lcfg.kv_heads, llm_mlp=lcfg.mlp,
                  llm_vocab=lcfg.vocab, llm_eps=lcfg.eps, llm_rope_theta=lcfg.rope_theta)
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

"""every Python call signature of the drop-in boundary against golden g18 (the reference's signatures, read from its
sources with `ast` by oracle/make_golden.py::g18_signatures).  Rule per surface: the reference's parameters are a PREFIX of the
build's (same names, same order, same kinds; where the reference states a default, the same default); whatever the build adds
comes after them and has a default, so every call that is valid against the reference -- positional or keyword -- binds
identically.  A default where the reference REQUIRES the argument is allowed (a superset of the valid calls).  CPU only: nothing here touches the GPU library's compute."""
import importlib
import inspect
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g18_signatures.json")

# reference file -> module of the build, through the alias package a reference caller would import
MODULE_OF = {
    "streammind/__init__.py": "videollama2",
    "streammind/eval/video_score_stream_demo.py": "videollama2.eval.video_score_stream_demo",
    "streammind/eval/video_test_stream_demo.py": "videollama2.eval.video_test_stream_demo",
    "streammind/model/builder.py": "videollama2.model.builder",
    "streammind/model/multimodal_encoder/clip_encoder.py": "videollama2.model.multimodal_encoder.clip_encoder",
    "streammind/model/multimodal_projector/builder.py": "videollama2.model.multimodal_projector.builder",
    "streammind/model/language_model/videollama2_mistral.py": "videollama2.model.language_model.videollama2_mistral",
    "streammind/mm_utils.py": "videollama2.mm_utils",
    "streammind/dist.py": "videollama2.dist",
    "streammind/serve/model_worker.py": "videollama2.serve.model_worker",
    "streammind/serve/controller.py": "videollama2.serve.controller",
    "process_clip_encoder.py": "streammind_amd.feature_cache",
}
# the reference spells defaults as source text; names it uses resolve to these values
DEFAULT_NAMES = {"NUM_FRAMES": 8, "IMAGE_TOKEN_INDEX": -200}
# surfaces the build deliberately does not mirror 1:1, with the reason a reader needs
WAIVED = {
    # the reference's method is the HF-streamer body of /worker_generate_stream; same name and (self, params) here
}


def _gold():
    with open(GOLD) as f:
        return json.load(f)


def _resolve(modname, qual):
    obj = importlib.import_module(modname)
    for part in qual.split("."):
        obj = getattr(obj, part)
    return obj


def _ours(obj):
    out = []
    for p in inspect.signature(obj).parameters.values():
        kind = {p.POSITIONAL_ONLY: "pos", p.POSITIONAL_OR_KEYWORD: "pos", p.VAR_POSITIONAL: "var", p.KEYWORD_ONLY: "kw",
                p.VAR_KEYWORD: "varkw"}[p.kind]
        out.append((p.name, kind, p.default))
    return out


def _default_value(src):
    return DEFAULT_NAMES[src] if src in DEFAULT_NAMES else eval(src, {"__builtins__": {}}, {})


@pytest.mark.parametrize("surface", sorted(_gold()["signatures"].keys()))
def test_signature_is_the_references(surface):
    ref = _gold()["signatures"][surface]
    path, qual = surface.split("::")
    if surface in WAIVED:
        pytest.skip(WAIVED[surface])
    ours = _ours(_resolve(MODULE_OF[path], qual))
    ref_fixed = [r for r in ref if r[1] in ("pos", "kw")]
    assert len(ours) >= len(ref_fixed), (surface, ours, ref)
    for i, (name, kind, dsrc) in enumerate(ref_fixed):
        oname, okind, odef = ours[i]
        assert oname == name, f"{surface}: parameter {i} is {oname!r}, the reference's is {name!r}"
        assert okind == kind, (surface, name, okind, kind)
        if dsrc is not None:
            assert odef is not inspect.Parameter.empty and odef == _default_value(dsrc), (surface, name, odef, dsrc)
    for oname, okind, odef in ours[len(ref_fixed):]:
        assert okind in ("var", "varkw") or odef is not inspect.Parameter.empty, f"{surface}: added parameter {oname} needs a default"
    if any(r[1] == "varkw" for r in ref):
        assert any(k == "varkw" for _, k, _ in ours), f"{surface}: the reference accepts **kwargs"


def test_names_a_caller_addresses():
    names = _gold()["names"]
    from videollama2 import constants, conversation
    assert dict(constants.MMODAL_TOKEN_INDEX) == names["MMODAL_TOKEN_INDEX"]
    assert dict(constants.DEFAULT_MMODAL_TOKEN) == names["DEFAULT_MMODAL_TOKEN"]
    assert constants.NUM_FRAMES == names["NUM_FRAMES"] and constants.MAX_FRAMES == names["MAX_FRAMES"]
    # the templates the streaming path and the package API name (conversation.py:383-393, __init__.py:38)
    for t in ("mistral_instruct", "llama_2"):
        assert t in names["conv_templates"] and t in conversation.conv_templates


def test_package_infer_is_the_offline_api_and_the_tick_lives_in_eval():
    """VERDICT r3 missing #1: `from videollama2 import infer` is streammind/__init__.py:38-91 (-> str); the streaming tick is
    eval/video_score_stream_demo.py:66-125 (-> (reply | None, prompt))."""
    import streammind
    import videollama2
    from videollama2.eval.video_score_stream_demo import infer as tick
    assert videollama2.infer is streammind.infer
    assert "score_video" not in inspect.signature(videollama2.infer).parameters
    assert list(inspect.signature(tick).parameters)[:8] == ["model", "video", "instruct", "tokenizer", "do_sample", "version", "score_video", "prompt"]
    assert videollama2.stream_infer is not videollama2.infer

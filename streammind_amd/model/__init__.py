"""Host-side mirror of the reference's model surface for the streaming path:

    CLIPVisionTower.forward                      streammind/model/multimodal_encoder/clip_encoder.py:41-84
    Video_Mamba_seq.__call__ (mm_projector)       streammind/model/multimodal_projector/builder.py:390-414,547-564
    Videollama2MistralForCausalLM.stream_generate_demo   language_model/videollama2_mistral.py:385-439
    load_pretrained_model                         streammind/model/builder.py:30-210

Same names, argument meaning and error behaviour; every computation is a libstreammind_hip.so call."""
from .stream_model import CLIPVisionTower, Video_Mamba_seq, Videollama2MistralForCausalLM, build_from_state_dicts
from .builder import load_pretrained_model


class _OtherFamily:
    """Llama / Mixtral wrappers of the reference (language_model/videollama2_{llama,mixtral}.py): the streaming path is
    Mistral-only (out of scope).  The names exist so that `from videollama2.model import ...` of the
    reference's callers resolves; using them raises."""

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: only the Mistral StreamMind path is built for MI355X")

    @classmethod
    def from_pretrained(cls, *a, **k):
        raise NotImplementedError(f"{cls.__name__}: only the Mistral StreamMind path is built for MI355X")


class Videollama2LlamaForCausalLM(_OtherFamily):
    pass


class Videollama2MixtralForCausalLM(_OtherFamily):
    pass


__all__ = ["CLIPVisionTower", "Video_Mamba_seq", "Videollama2MistralForCausalLM", "Videollama2LlamaForCausalLM",
           "Videollama2MixtralForCausalLM", "load_pretrained_model", "build_from_state_dicts"]

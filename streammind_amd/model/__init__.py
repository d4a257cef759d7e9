"""Host-side mirror of the reference's model surface for the streaming path:

    CLIPVisionTower.forward                      streammind/model/multimodal_encoder/clip_encoder.py:41-84
    Video_Mamba_seq.__call__ (mm_projector)       streammind/model/multimodal_projector/builder.py:390-414,547-564
    Videollama2MistralForCausalLM.stream_generate_demo   language_model/videollama2_mistral.py:385-439
    load_pretrained_model                         streammind/model/builder.py:30-210

Same names, argument meaning and error behaviour; every computation is a libstreammind_hip.so call."""
from .stream_model import (CLIPVisionTower, Video_Mamba_seq, Videollama2MistralForCausalLM, load_pretrained_model,
                           build_from_state_dicts)

__all__ = ["CLIPVisionTower", "Video_Mamba_seq", "Videollama2MistralForCausalLM", "load_pretrained_model",
           "build_from_state_dicts"]

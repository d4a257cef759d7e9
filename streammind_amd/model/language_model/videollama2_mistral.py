"""streammind/model/language_model/videollama2_mistral.py: the module path reference callers import
`Videollama2MistralForCausalLM` from; the class itself is streammind_amd/model/stream_model.py."""
from ..stream_model import Videollama2MistralForCausalLM  # noqa: F401

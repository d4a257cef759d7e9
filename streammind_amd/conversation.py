"""Prompt templates of the streaming path (mirror of streammind/conversation.py: the LLAMA_2 separator style with the
hard-wired instruction sentence, :78-98, and the `mistral_instruct` template, :383-393).  Only the templates the hot path
uses are provided; the byte-exact prompt strings are pinned by tests/golden/g5_prompt.npz."""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    LLAMA_2 = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: tuple
    messages: list
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.LLAMA_2
    sep: str = ""
    sep2: Optional[str] = None
    version: str = "llama_v2"

    def get_prompt(self) -> str:
        if self.sep_style != SeparatorStyle.LLAMA_2:
            raise NotImplementedError("only the LLAMA_2 style is on the streaming path")
        ret = ""
        for i, (role, message) in enumerate(self.messages):
            if i == 0:
                assert message, "first message should not be none"
                assert role == self.roles[0], "first message should come from user"
            if message:
                if type(message) is tuple:
                    message = message[0]
                if i == 0:   # conversation.py:90 -- the instruction sentence is hard-wired into the first user turn
                    message = (f"<<SYS>>\n{self.system}\n<</SYS>>\n\n"
                               "Please describe the video content in detail based on the provided information." + message)
                if i % 2 == 0:
                    ret += self.sep + f"[INST] {message} [/INST]"
                else:
                    ret += " " + message + " " + self.sep2
        return ret.lstrip(self.sep) if self.sep else ret

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[x, y] for x, y in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)


conv_mistral_instruct = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=("USER", "ASSISTANT"), version="llama_v2", messages=[], offset=0,
    sep_style=SeparatorStyle.LLAMA_2, sep="", sep2="</s>")

# conversation.py:452-463: the template the package-level offline `infer` / `x_infer` select for Mistral checkpoints
# (model_init returns version "llama_2", streammind/__init__.py:27-33); same LLAMA_2 style, other system text, "<s>" separator
conv_llama_2 = Conversation(
    system="You are a helpful, respectful and honest assistant. Always answer as helpfully as possible, while being safe.  "
           "Your answers should not include any harmful, unethical, racist, sexist, toxic, dangerous, or illegal content. "
           "Please ensure that your responses are socially unbiased and positive in nature.\n\n"
           "If a question does not make any sense, or is not factually coherent, explain why instead of answering something "
           "not correct. If you don't know the answer to a question, please don't share false information.",
    roles=("USER", "ASSISTANT"), version="llama_v2", messages=[], offset=0,
    sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")

default_conversation = conv_mistral_instruct
conv_templates = {"mistral_instruct": conv_mistral_instruct, "llama_2": conv_llama_2}

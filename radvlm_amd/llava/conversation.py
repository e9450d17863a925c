"""Prompt templates of the training path (the literal strings of reference finetuning/llava/conversation.py:
conv_vicuna_v1 :345-354, conv_qwen :443-452, conv_llava_plain :456-463; get_prompt :47-134).

Unlike the reference module this one has no import-time side effect (the reference performs a hub tokenizer lookup
at import, conversation.py:380-395).
"""
import copy
import dataclasses
from enum import Enum, auto
from typing import List, Optional, Tuple


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    CHATML = auto()
    LLAMA_2 = auto()
    LLAMA_3 = auto()
    QWEN = auto()
    GEMMA = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Tuple[str, str]
    messages: List[List[str]]
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        style = self.sep_style
        if style == SeparatorStyle.TWO:
            seps = (self.sep, self.sep2)
            out = self.system + seps[0]
            for i, (role, msg) in enumerate(self.messages):
                out += f"{role}: {msg}{seps[i % 2]}" if msg else f"{role}:"
            return out
        if style == SeparatorStyle.SINGLE:
            out = self.system + self.sep
            for role, msg in self.messages:
                out += f"{role}: {msg}{self.sep}" if msg else f"{role}:"
            return out
        if style == SeparatorStyle.CHATML:
            out = "" if self.system == "" else self.system + self.sep + "\n"
            for role, msg in self.messages:
                out += f"{role}\n{msg}{self.sep}\n" if msg else f"{role}\n"
            return out
        if style == SeparatorStyle.PLAIN:
            seps = (self.sep, self.sep2)
            out = self.system
            for i, (_, msg) in enumerate(self.messages):
                out += (msg + seps[i % 2]) if msg else ""
            return out
        raise ValueError(f"Invalid style: {style}")

    def copy(self):
        return copy.deepcopy(self)


conv_vicuna_v1 = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=("USER", "ASSISTANT"), version="v1", messages=[], offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

conv_qwen = Conversation(
    system="<|im_start|>system\nYou are a helpful assistant.",
    roles=("<|im_start|>user", "<|im_start|>assistant"), version="qwen", messages=[], offset=0,
    sep_style=SeparatorStyle.CHATML, sep="<|im_end|>")

conv_llava_plain = Conversation(system="", roles=("", ""), messages=[], offset=0, sep_style=SeparatorStyle.PLAIN, sep="\n")

conv_templates = {
    "default": conv_vicuna_v1,
    "v1": conv_vicuna_v1,
    "vicuna_v1": conv_vicuna_v1,
    "llava_v1": conv_vicuna_v1,
    "plain": conv_llava_plain,
    "v0_plain": conv_llava_plain,
    "qwen_1_5": conv_qwen,
    "qwen_2": conv_qwen,
}
default_conversation = conv_vicuna_v1

"""Prompt templates for the grounding path (host side of the kept API).

Same surface as revisionllm/conversation.py (``Conversation``, ``SeparatorStyle``, ``conv_templates``) for the
templates the inference path can select; ``conv_templates["v1"]`` (Vicuna v1: sep " ", sep2 "</s>", style TWO,
conversation.py:253-263) is the one ``inference()`` uses (inference.py:31).
"""
import dataclasses
from enum import Enum, auto
from typing import List, Optional


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: tuple
    messages: List[list]
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def _text(self, message):
        return message[0] if isinstance(message, tuple) else message

    def get_prompt(self) -> str:
        """Render the dialogue (conversation.py:29-104).  A ``None`` message leaves the role open ("ROLE:")."""
        st = self.sep_style
        if st in (SeparatorStyle.SINGLE, SeparatorStyle.TWO):
            seps = [self.sep, self.sep] if st == SeparatorStyle.SINGLE else [self.sep, self.sep2]
            out = self.system + seps[0]
            for i, (role, message) in enumerate(self.messages):
                out += f"{role}: {self._text(message)}{seps[i % 2]}" if message else f"{role}:"
            return out
        if st == SeparatorStyle.MPT:
            out = self.system + self.sep
            for role, message in self.messages:
                out += role + self._text(message) + self.sep if message else role
            return out
        if st == SeparatorStyle.PLAIN:
            seps = [self.sep, self.sep2]
            out = self.system
            for i, (_, message) in enumerate(self.messages):
                if message:
                    out += self._text(message) + seps[i % 2]
            return out
        if st == SeparatorStyle.LLAMA_2:
            out = ""
            for i, (role, message) in enumerate(self.messages):
                if not message:
                    continue
                message = self._text(message)
                if i == 0:
                    message = f"<<SYS>>\n{self.system}\n<</SYS>>\n\n" + message
                out += (self.sep + f"[INST] {message} [/INST]") if i % 2 == 0 else (" " + message + " " + self.sep2)
            return out.lstrip(self.sep)
        raise ValueError(f"Invalid style: {self.sep_style}")

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self):
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)

    def dict(self):
        return {"system": self.system, "roles": self.roles, "messages": self.messages, "offset": self.offset,
                "sep": self.sep, "sep2": self.sep2}


conv_vicuna_v1 = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=("USER", "ASSISTANT"), version="v1", messages=[], offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

conv_llama_2 = Conversation(
    system="You are a helpful, respectful and honest assistant. Always answer as helpfully as possible, while being safe.",
    roles=("USER", "ASSISTANT"), version="llama_v2", messages=[], offset=0, sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")

conv_plain = Conversation(system="", roles=("", ""), messages=[], offset=0, sep_style=SeparatorStyle.PLAIN, sep="\n")

default_conversation = conv_vicuna_v1
conv_templates = {"default": conv_vicuna_v1, "v1": conv_vicuna_v1, "vicuna_v1": conv_vicuna_v1, "llama_2": conv_llama_2,
                  "plain": conv_plain}

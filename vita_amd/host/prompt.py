"""Host-side prompt assembly and tokenisation with multimodal placeholders.

Behavioural mirror (not a copy) of the reference's vita/conversation.py:30-105,229-298 and
vita/util/mm_utils.py:45-155 for the pieces video_audio_demo.py uses:
  conv_templates["mixtral_two"].copy() -> append_message() -> get_prompt(modality)
  tokenizer_image_token / tokenizer_image_audio_token  (sentinels -200 / -500)
  KeywordsStoppingCriteria, get_model_name_from_path
The three system prompts are model-facing data and must match the reference byte for byte."""
import copy
import re
from enum import Enum, auto

import torch

from .constants import AUDIO_TOKEN_INDEX, IMAGE_TOKEN_INDEX


class SeparatorStyle(Enum):
    TWO = auto()
    PLAIN = auto()
    MixtralZh = auto()
    MixtralTwo = auto()


_SYS_COMMON = ("You are an AI robot and your name is VITA. \n- You are a multimodal large language model developed by "
               "the open source community. Your aim is to be helpful, honest and harmless. \n- You support the ability "
               "to communicate fluently and answer user questions in multiple languages of the user's choice. \n- If "
               "the user corrects the wrong answer you generated, you will apologize and discuss the correct answer "
               "with the user.")
_SYS_MEDIA = (" \n- You must answer the question strictly according to the content of the {0} given by the user, and it "
              "is strictly forbidden to answer the question without the content of the {0}. Please note that you are "
              "seeing the {0}, not the {1}.")
SYSTEM_PROMPTS = [_SYS_COMMON + _SYS_MEDIA.format("image", "video"), _SYS_COMMON + _SYS_MEDIA.format("video", "image"),
                  _SYS_COMMON]


class Conversation:
    def __init__(self, system, roles, sep_style, sep="###", sep2=None, version="Unknown", messages=(), offset=0):
        self.system, self.roles, self.sep_style = system, tuple(roles), sep_style
        self.sep, self.sep2, self.version = sep, sep2, version
        self.messages, self.offset = [list(m) for m in messages], offset

    def copy(self):
        return copy.deepcopy(self)

    def append_message(self, role, message):
        self.messages.append([role, message])

    @staticmethod
    def _text(message):
        return message[0] if isinstance(message, tuple) else message

    def get_prompt(self, modality=None):
        seps = [self.sep, self.sep2]
        msgs = self.messages
        if self.sep_style == SeparatorStyle.MixtralTwo:
            has_image = any(m and "<image>" in self._text(m) for _, m in msgs)
            if has_image:
                if modality not in ("image", "video"):
                    raise AssertionError("prompt holds <image> but modality is not image/video")
                self.system = self.system[0 if modality == "image" else 1]
            else:
                if modality != "lang":
                    raise AssertionError("text-only prompt needs modality='lang'")
                self.system = self.system[2]
            out = "system:" + self.system + seps[0]
            for i, (role, m) in enumerate(msgs):
                out += "\n" + role + ":" + (self._text(m) + seps[i % 2] if m else "")
            return out
        if self.sep_style == SeparatorStyle.MixtralZh:
            out = "system:" + self.system + seps[0]
            for i, (role, m) in enumerate(msgs):
                out += "\n" + role + ":" + (self._text(m) + seps[i % 2] if m else "")
            return out
        if self.sep_style == SeparatorStyle.TWO:
            out = self.system + seps[0]
            for i, (role, m) in enumerate(msgs):
                out += role + ": " + self._text(m) + seps[i % 2] if m else role + ":"
            return out
        if self.sep_style == SeparatorStyle.PLAIN:
            return self.system + "".join(self._text(m) + seps[i % 2] for i, (_, m) in enumerate(msgs) if m)
        raise ValueError(f"Invalid style: {self.sep_style}")


conv_mixtral_two = Conversation(system=SYSTEM_PROMPTS, roles=("user", "bot"), version="mixtral_two",
                                sep_style=SeparatorStyle.MixtralTwo, sep="</s>", sep2="</s>")
conv_mixtral_zh = Conversation(system=_SYS_COMMON, roles=("user", "bot"), version="mixtral_zh",
                               sep_style=SeparatorStyle.MixtralZh, sep="</s>", sep2="</s>")
conv_plain = Conversation(system="", roles=("", ""), sep_style=SeparatorStyle.PLAIN, sep="\n")
default_conversation = conv_mixtral_two
conv_templates = {"default": conv_mixtral_two, "mixtral_two": conv_mixtral_two, "mixtral_zh": conv_mixtral_zh,
                  "plain": conv_plain}


# ---- tokenisation with placeholders -----------------------------------------------------------
def _splice_chunks(chunks, tokenizer, sentinels):
    """chunks: list of either a sentinel int or a tokenised id list.  A BOS at the head of the first
    text chunk is kept once; every later text chunk drops its own BOS (mm_utils.py:57-66,91-104)."""
    ids, drop = [], 0
    first = chunks[0] if chunks else []
    if isinstance(first, list) and first and first[0] == tokenizer.bos_token_id:
        drop = 1
        ids.append(first[0])
    for c in chunks:
        if isinstance(c, list):
            ids.extend(c[drop:])
        else:
            ids.append(c)
    return ids


def _ret(ids, return_tensors):
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    parts = prompt.split("<image>")
    chunks = []
    for i, p in enumerate(parts):
        if i:
            chunks.append(image_token_index)
        chunks.append(list(tokenizer(p).input_ids))
    return _ret(_splice_chunks(chunks, tokenizer, (image_token_index,)), return_tensors)


def tokenizer_image_audio_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX,
                                audio_token_index=AUDIO_TOKEN_INDEX, return_tensors=None):
    chunks = []
    for piece in re.split(r"(<audio>|<image>)", prompt):
        if piece == "<audio>":
            chunks.append(audio_token_index)
        elif piece == "<image>":
            chunks.append(image_token_index)
        else:
            chunks.append(list(tokenizer(piece).input_ids))
    return _ret(_splice_chunks(chunks, tokenizer, (image_token_index, audio_token_index)), return_tensors)


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """Stop when the generated tail equals a keyword's ids or its decoded text contains the keyword
    (mm_utils.py:121-155).  Callable as stopping_criteria(output_ids, scores)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords, self.tokenizer = keywords, tokenizer
        self.keyword_ids, self.max_keyword_len = [], 0
        for kw in keywords:
            ids = list(tokenizer(kw).input_ids)
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.start_len = input_ids.shape[1]

    def _one(self, output_ids):
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        for kid in self.keyword_ids:
            kid = kid.to(output_ids.device)
            if output_ids.shape[1] >= kid.shape[0] and torch.equal(output_ids[0, -kid.shape[0]:], kid):
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores=None, **kwargs):
        return all(self._one(output_ids[i].unsqueeze(0)) for i in range(output_ids.shape[0]))

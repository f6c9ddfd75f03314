"""Host mirror of the Klein prompt-embedding path (FluxTextEncoders/Embeddings/KleinEmbeddingExtractor.swift):
chat template + padding on the host, Qwen3 hidden states on the GPU (csrc/textenc.cpp)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import Flux2Error, check, lib
from .api import Context, _ptr, _req, _set_tensors

KLEIN_MAX_SEQUENCE_LENGTH = 512          # KleinConfig.maxSequenceLength
QWEN3_PAD_TOKEN_ID = 151643              # <|endoftext|>  (KleinEmbeddingExtractor.swift:78-95)
QWEN3_IM_START, QWEN3_IM_END = 151644, 151645


def format_qwen3_chat_template(user_message: str, add_generation_prompt: bool = True) -> str:
    """KleinEmbeddingExtractor.formatQwen3ChatTemplate (:143-162): no system message, the assistant
    turn opened with an empty <think> block."""
    s = "<|im_start|>user\n" + user_message + "<|im_end|>\n"
    if add_generation_prompt:
        s += "<|im_start|>assistant\n<think>\n\n</think>\n\n"
    return s


# TekkenTokenizer's special token ranks (FluxTextEncoders/Tokenizer/TekkenTokenizer.swift:46-53; overridden from
# tekken.json / generation_config.json when those are present)
MISTRAL_SPECIAL_TOKENS = {"bos": 1, "eos": 2, "inst": 3, "end_inst": 4, "pad": 11, "system_prompt": 17, "end_system_prompt": 18}
# FluxConfig.systemMessage (Embeddings/EmbeddingExtractor.swift:17-19): the fixed system turn of FLUX.2-dev prompts
FLUX_SYSTEM_MESSAGE = ("You are an AI that reasons about image descriptions. You give structured responses focusing on object "
                       "relationships, object attribution and actions without speculation.")


def encode_mistral_chat_messages(messages, encode, special: Optional[Dict[str, int]] = None) -> List[int]:
    """`TekkenTokenizer.encodeChatMessages(messages:addGenerationPrompt:)`, manual layout (:516-536):
    <s> then per message  system: [SYSTEM_PROMPT] text [/SYSTEM_PROMPT];  user: [INST] text [/INST];
    assistant: text </s>.  `encode(text) -> ids` must not add special tokens.  Unknown roles and messages without
    role/content are skipped, like the reference."""
    sp = dict(MISTRAL_SPECIAL_TOKENS, **(special or {}))
    tokens = [sp["bos"]]
    for m in messages:
        role, content = m.get("role"), m.get("content")
        if role is None or content is None:
            continue
        if role == "system":
            tokens += [sp["system_prompt"], *encode(content), sp["end_system_prompt"]]
        elif role == "user":
            tokens += [sp["inst"], *encode(content), sp["end_inst"]]
        elif role == "assistant":
            tokens += [*encode(content), sp["eos"]]
    return tokens


def flux_prompt_token_ids(prompt: str, encode, special: Optional[Dict[str, int]] = None) -> List[int]:
    """`getFluxTokenIds` before padding (EmbeddingExtractor.swift:206-218): "[IMG]" stripped from the prompt,
    [system = FluxConfig.systemMessage, user = prompt], no generation prompt."""
    return encode_mistral_chat_messages(
        [{"role": "system", "content": FLUX_SYSTEM_MESSAGE}, {"role": "user", "content": prompt.replace("[IMG]", "")}], encode, special)


def pad_klein_token_ids(token_ids: Sequence[int], max_length: int = KLEIN_MAX_SEQUENCE_LENGTH,
                        pad_id: int = QWEN3_PAD_TOKEN_ID) -> Tuple[List[int], int]:
    """Truncate to max_length, RIGHT-pad with pad_id (:70-90).  Returns (ids, number of real tokens)."""
    ids = list(token_ids)[:max_length]
    n = len(ids)
    return ids + [pad_id] * (max_length - n), n


class TextEncoder:
    """`KleinTextEncoder` restricted to embedding extraction (flux2mi_textenc)."""

    def __init__(self, ctx: Context, config):
        self.ctx = ctx
        cfg = L.TextEncConfig()
        if isinstance(config, str):
            check(lib().flux2mi_textenc_preset(config.encode(), C.byref(cfg)))
        else:
            for k in ("vocab", "hidden", "intermediate", "n_layers", "heads", "kv_heads"):
                setattr(cfg, k, int(config[k]))
            cfg.head_dim = int(config.get("head_dim", 128))
            cfg.rope_theta = float(config.get("rope_theta", 1.0e6))
            cfg.rms_eps = float(config.get("rms_eps", 1.0e-6))
            for i, v in enumerate(config.get("extract_layers", (9, 18, 27))):
                cfg.extract_layers[i] = int(v)
            cfg.qk_norm = int(config.get("qk_norm", 1))
        self.cfg = cfg
        h = C.c_void_p()
        check(lib().flux2mi_textenc_create(ctx.handle, C.byref(cfg), C.byref(h)))
        self.handle = h
        self.tokenizer = None
        ctx._children.append(__import__("weakref").ref(self))

    def load(self, tensors: Dict[str, torch.Tensor]):
        """HF Qwen3 tensors by key; layers past the last extracted one, lm_head and the final norm are skipped."""
        run = self.cfg.extract_layers[2]
        keep = {}
        for k, v in tensors.items():
            base = k[len("model."):] if k.startswith("model.") else k
            if base.startswith("layers."):
                if int(base.split(".")[1]) >= run:
                    continue
            elif base != "embed_tokens.weight":
                continue
            keep[k] = v
        _set_tensors(lib().flux2mi_textenc_set_tensor, self.handle, keep)
        return self

    def finalize(self):
        check(lib().flux2mi_textenc_finalize(self.handle))
        return self

    @property
    def param_bytes(self) -> int:
        return int(lib().flux2mi_textenc_param_bytes(self.handle))

    def forward(self, token_ids: torch.Tensor, valid_len: int, pad_left: int = 0) -> torch.Tensor:
        """ids i32 [S] (device), real tokens [pad_left, pad_left + valid_len) -> bf16 [1, S, 3*hidden]  (async)."""
        ids = _req(token_ids.reshape(-1), torch.int32, "token_ids")
        S = ids.shape[0]
        out = torch.empty((1, S, 3 * self.cfg.hidden), dtype=torch.bfloat16, device=ids.device)
        self.ctx._enter()
        check(lib().flux2mi_textenc_forward(self.handle, _ptr(ids), S, int(pad_left), int(valid_len), _ptr(out)))
        self.ctx._exit()
        return out

    # ---- KleinEmbeddingExtractor.extractKleinEmbeddings(prompt:maxLength:) -------------------------
    def set_tokenizer(self, tokenizer_json_path: str):
        from tokenizers import Tokenizer
        self.tokenizer = Tokenizer.from_file(tokenizer_json_path)
        return self

    def extract_klein_embeddings(self, prompt: str, max_length: int = KLEIN_MAX_SEQUENCE_LENGTH,
                                 token_ids: Optional[Sequence[int]] = None) -> torch.Tensor:
        """[1, max_length, 3*hidden] bf16.  `token_ids` bypasses the tokenizer (ids of the formatted prompt)."""
        if token_ids is None:
            if self.tokenizer is None:
                raise Flux2Error(L.ERR_NOT_LOADED, "tokenizer (set_tokenizer(path to tokenizer.json) or pass token_ids)")
            text = format_qwen3_chat_template(prompt.replace("[IMG]", ""), True)
            token_ids = self.tokenizer.encode(text, add_special_tokens=False).ids
        ids, n = pad_klein_token_ids(token_ids, max_length)
        if n == 0:
            raise Flux2Error(L.ERR_INVALID_ARG, "empty prompt")
        t = torch.tensor(ids, dtype=torch.int32, device=self.ctx.device)
        return self.forward(t, n)

    # ---- EmbeddingExtractor.extractFluxEmbeddings(prompt:maxLength:) (Dev, Mistral) -------------------
    def extract_flux_embeddings(self, token_ids: Optional[Sequence[int]] = None, max_length: int = KLEIN_MAX_SEQUENCE_LENGTH,
                                pad_id: int = MISTRAL_SPECIAL_TOKENS["pad"], prompt: Optional[str] = None) -> torch.Tensor:
        """`prompt` is encoded as [system message, user prompt] with the Mistral chat layout
        (`encode_mistral_chat_messages`, EmbeddingExtractor.swift:206-218) through the loaded tokenizer, or
        `token_ids` carries that encoding; then truncated to max_length and LEFT-padded with the tokenizer's pad
        token (:226-237).  -> [1, max_length, 3*hidden] bf16."""
        if token_ids is None:
            if prompt is None or self.tokenizer is None:
                raise Flux2Error(L.ERR_NOT_LOADED, "tokenizer (set_tokenizer(path to tokenizer.json) + prompt, or pass token_ids)")
            token_ids = flux_prompt_token_ids(prompt, lambda t: self.tokenizer.encode(t, add_special_tokens=False).ids)
        ids = list(token_ids)[:max_length]
        n = len(ids)
        if n == 0:
            raise Flux2Error(L.ERR_INVALID_ARG, "empty prompt")
        t = torch.tensor([pad_id] * (max_length - n) + ids, dtype=torch.int32, device=self.ctx.device)
        return self.forward(t, n, pad_left=max_length - n)

    def close(self):
        if self.handle and self.ctx.handle:
            lib().flux2mi_textenc_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

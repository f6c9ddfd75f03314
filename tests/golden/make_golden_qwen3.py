#!/usr/bin/env python3
"""tests/golden/qwen3_tiny.npz: hidden states of a randomly initialised small Qwen3 computed by
HuggingFace transformers' `Qwen3Model` — an independent implementation of the text encoder the
reference re-implements in Swift (FluxTextEncoders/Model/Qwen3).  The fixture stores the weights'
seed-free VALUES (bf16-exact, as f16-safe f32 would be large: weights are regenerated from a torch
seed and a checksum is stored), the token ids, the attention mask length and the hidden states after
layers 1, 2, 3 of a 4-layer model.  Usage: python tests/golden/make_golden_qwen3.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

TINY = dict(vocab=1000, hidden=256, intermediate=512, n_layers=4, heads=4, kv_heads=2, head_dim=128,
            rope_theta=1.0e6, rms_eps=1.0e-6, extract_layers=(1, 2, 3))


MISTRAL_TINY = dict(vocab=1000, hidden=256, intermediate=512, n_layers=4, heads=4, kv_heads=2, head_dim=128,
                    rope_theta=1.0e9, rms_eps=1.0e-5, extract_layers=(1, 2, 3), qk_norm=0)


def make_qwen3_weights(tcfg, seed=77):
    """Deterministic bf16-exact weights in HF naming (model.* prefix dropped)."""
    g = torch.Generator().manual_seed(seed)
    d, I, H, Hkv, hd = tcfg["hidden"], tcfg["intermediate"], tcfg["heads"], tcfg["kv_heads"], tcfg["head_dim"]
    P = {"embed_tokens.weight": torch.randn((tcfg["vocab"], d), generator=g)}
    for l in range(tcfg["n_layers"]):
        p = f"layers.{l}."
        P[p + "self_attn.q_proj.weight"] = torch.randn((H * hd, d), generator=g) * d ** -0.5
        P[p + "self_attn.k_proj.weight"] = torch.randn((Hkv * hd, d), generator=g) * d ** -0.5
        P[p + "self_attn.v_proj.weight"] = torch.randn((Hkv * hd, d), generator=g) * d ** -0.5
        P[p + "self_attn.o_proj.weight"] = torch.randn((d, H * hd), generator=g) * (H * hd) ** -0.5
        if tcfg.get("qk_norm", 1):
            P[p + "self_attn.q_norm.weight"] = 1.0 + 0.1 * torch.randn((hd,), generator=g)
            P[p + "self_attn.k_norm.weight"] = 1.0 + 0.1 * torch.randn((hd,), generator=g)
        P[p + "mlp.gate_proj.weight"] = torch.randn((I, d), generator=g) * d ** -0.5
        P[p + "mlp.up_proj.weight"] = torch.randn((I, d), generator=g) * d ** -0.5
        P[p + "mlp.down_proj.weight"] = torch.randn((d, I), generator=g) * I ** -0.5
        P[p + "input_layernorm.weight"] = 1.0 + 0.1 * torch.randn((d,), generator=g)
        P[p + "post_attention_layernorm.weight"] = 1.0 + 0.1 * torch.randn((d,), generator=g)
    return {k: v.to(torch.bfloat16) for k, v in P.items()}


def main():
    from transformers import Qwen3Config, Qwen3Model
    t = TINY
    P = make_qwen3_weights(t)
    cfg = Qwen3Config(vocab_size=t["vocab"], hidden_size=t["hidden"], intermediate_size=t["intermediate"],
                      num_hidden_layers=t["n_layers"], num_attention_heads=t["heads"], num_key_value_heads=t["kv_heads"],
                      head_dim=t["head_dim"], max_position_embeddings=1024, rope_theta=t["rope_theta"],
                      rms_norm_eps=t["rms_eps"], attention_bias=False, tie_word_embeddings=True)
    m = Qwen3Model(cfg).eval().float()
    sd = {k: v.float() for k, v in P.items()}
    sd["norm.weight"] = torch.ones(t["hidden"])
    missing = m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    S, valid = 48, 29
    ids = torch.randint(0, t["vocab"], (S,), generator=g)
    ids[valid:] = 7   # pad id
    mask = (torch.arange(S) < valid).long()[None]
    with torch.no_grad():
        out = m(input_ids=ids[None], attention_mask=mask, output_hidden_states=True)
    hs = out.hidden_states   # [0] = embeddings, [i] = after layer i (the last one after the final norm)
    np.savez_compressed(os.path.join(OUT, "qwen3_tiny.npz"), ids=ids.numpy(), valid_len=np.array([valid]),
                        h1=hs[1][0].numpy(), h2=hs[2][0].numpy(), h3=hs[3][0].numpy(),
                        weight_checksum=np.array([float(sum(v.float().double().sum() for v in P.values()))]))
    print("written", os.path.join(OUT, "qwen3_tiny.npz"), [tuple(h.shape) for h in hs])

    # ---- Mistral (Dev's encoder): LEFT padding.  Only the rows of real tokens are comparable with HF:
    # a fully masked (pad) query row is implementation-defined there, while the reference's finite
    # -1e9 mask makes it a uniform average (restated and tested in the oracle).
    from transformers import MistralConfig, MistralModel
    t = MISTRAL_TINY
    P = make_qwen3_weights(t, seed=78)
    cfg = MistralConfig(vocab_size=t["vocab"], hidden_size=t["hidden"], intermediate_size=t["intermediate"],
                        num_hidden_layers=t["n_layers"], num_attention_heads=t["heads"], num_key_value_heads=t["kv_heads"],
                        head_dim=t["head_dim"], max_position_embeddings=1024, rope_theta=t["rope_theta"],
                        rms_norm_eps=t["rms_eps"], sliding_window=None, attention_dropout=0.0)
    m = MistralModel(cfg).eval().float()
    sd = {k: v.float() for k, v in P.items()}
    sd["norm.weight"] = torch.ones(t["hidden"])
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(6)
    S, valid = 48, 29
    pad = S - valid
    ids = torch.randint(0, t["vocab"], (S,), generator=g)
    ids[:pad] = 11
    mask = (torch.arange(S) >= pad).long()[None]
    with torch.no_grad():
        out = m(input_ids=ids[None], attention_mask=mask, position_ids=torch.arange(S)[None], output_hidden_states=True)
    hs = out.hidden_states
    np.savez_compressed(os.path.join(OUT, "mistral_tiny.npz"), ids=ids.numpy(), valid_len=np.array([valid]),
                        pad_left=np.array([pad]), h1=hs[1][0].numpy(), h2=hs[2][0].numpy(), h3=hs[3][0].numpy())
    print("written", os.path.join(OUT, "mistral_tiny.npz"))


if __name__ == "__main__":
    main()

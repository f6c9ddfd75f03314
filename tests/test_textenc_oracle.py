"""CPU: the oracle's Qwen3 restatement against the committed HF-transformers fixture
(tests/golden/qwen3_tiny.npz, made by tests/golden/make_golden_qwen3.py with `transformers.Qwen3Model`) —
this part of the oracle is PINNED against an independent implementation — plus the host-side
Klein prompt formatting rules."""
import os
import sys

import numpy as np
import torch

from oracle import flux2_ref as R

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_qwen3 as G  # noqa: E402


def test_qwen3_restatement_matches_hf_fixture():
    g = np.load("tests/golden/qwen3_tiny.npz")
    P = G.make_qwen3_weights(G.TINY)
    assert abs(float(sum(v.float().double().sum() for v in P.values())) - float(g["weight_checksum"][0])) < 1e-6
    hs = R.qwen3_hidden_states(P, G.TINY, torch.from_numpy(g["ids"]), int(g["valid_len"][0]), (1, 2, 3))
    for i in (1, 2, 3):
        np.testing.assert_allclose(hs[i][0].numpy(), g[f"h{i}"], rtol=2e-5, atol=2e-5)
    emb = R.klein_embeddings(P, G.TINY, torch.from_numpy(g["ids"]), int(g["valid_len"][0]))
    assert emb.shape == (1, 48, 3 * 256)
    np.testing.assert_allclose(emb[0, :, 256:512].numpy(), g["h2"], rtol=2e-5, atol=2e-5)


def test_padding_does_not_change_real_tokens():
    """Causal + right padding: hidden states of the real tokens do not depend on the pad length."""
    g = np.load("tests/golden/qwen3_tiny.npz")
    P = G.make_qwen3_weights(G.TINY)
    ids, v = torch.from_numpy(g["ids"]), int(g["valid_len"][0])
    a = R.qwen3_hidden_states(P, G.TINY, ids, v, (3,))[3][0, :v]
    b = R.qwen3_hidden_states(P, G.TINY, ids[:v], v, (3,))[3][0]
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-5)


def test_klein_prompt_formatting(flux2mi):
    te = flux2mi.textenc
    # KleinEmbeddingExtractor.swift:143-162
    assert te.format_qwen3_chat_template("a cat") == "<|im_start|>user\na cat<|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n"
    assert te.format_qwen3_chat_template("x", False) == "<|im_start|>user\nx<|im_end|>\n"
    ids, n = te.pad_klein_token_ids([5, 6, 7], 8)
    assert ids == [5, 6, 7] + [151643] * 5 and n == 3       # RIGHT padding with <|endoftext|> (:70-90)
    ids, n = te.pad_klein_token_ids(list(range(600)))
    assert len(ids) == 512 and n == 512                     # truncation to maxSequenceLength


def test_mistral_left_padding_matches_hf_fixture_on_real_rows():
    """Dev's encoder (Mistral: no q/k norm, LEFT padding).  The rows of real tokens are pinned against
    HF `MistralModel` (tests/golden/mistral_tiny.npz); pad rows are implementation-defined in HF, the
    reference's finite -1e9 pad mask makes them uniform averages over the pads before them."""
    g = np.load("tests/golden/mistral_tiny.npz")
    P = G.make_qwen3_weights(G.MISTRAL_TINY, seed=78)
    v, pad = int(g["valid_len"][0]), int(g["pad_left"][0])
    ids = torch.from_numpy(g["ids"])
    hs = R.qwen3_hidden_states(P, G.MISTRAL_TINY, ids, v, (1, 2, 3), pad_left=pad)
    for i in (1, 2, 3):
        np.testing.assert_allclose(hs[i][0, pad:].numpy(), g[f"h{i}"][pad:], rtol=2e-5, atol=2e-5)
    # pad query row i attends uniformly to keys 0..i: layer-1 attention output of row 0 = V row 0 path,
    # checked through the defining property: the first pad rows do not depend on any later token
    ids2 = ids.clone()
    ids2[pad:] = (ids2[pad:] + 17) % G.MISTRAL_TINY["vocab"]
    hs2 = R.qwen3_hidden_states(P, G.MISTRAL_TINY, ids2, v, (3,), pad_left=pad)
    np.testing.assert_allclose(hs[3][0, :pad].numpy(), hs2[3][0, :pad].numpy(), rtol=1e-5, atol=1e-5)


def test_mistral_chat_message_layout():
    """TekkenTokenizer.encodeChatMessages manual layout (FluxTextEncoders/Tokenizer/TekkenTokenizer.swift:516-536) and
    the FLUX.2-dev prompt of EmbeddingExtractor.extractFluxEmbeddings (:206-218): host string/id logic, no GPU."""
    from flux2mi.textenc import FLUX_SYSTEM_MESSAGE, encode_mistral_chat_messages, flux_prompt_token_ids
    enc = lambda text: [1000 + b for b in text.encode("utf-8")]
    ids = encode_mistral_chat_messages([{"role": "system", "content": "S"}, {"role": "user", "content": "hi"},
                                        {"role": "assistant", "content": "ok"}, {"role": "tool", "content": "x"}, {"role": "user"}], enc)
    assert ids == [1, 17, 1000 + ord("S"), 18, 3, 1000 + ord("h"), 1000 + ord("i"), 4, 1000 + ord("o"), 1000 + ord("k"), 2]
    ids = flux_prompt_token_ids("a cat[IMG] on a mat", enc)
    sys_ids, user_ids = enc(FLUX_SYSTEM_MESSAGE), enc("a cat on a mat")
    assert ids == [1, 17, *sys_ids, 18, 3, *user_ids, 4]                 # no generation prompt, [IMG] stripped
    assert FLUX_SYSTEM_MESSAGE.startswith("You are an AI that reasons about image descriptions.")
    # special token ranks can come from the tokenizer files
    assert encode_mistral_chat_messages([{"role": "user", "content": ""}], enc, {"bos": 7, "inst": 8, "end_inst": 9}) == [7, 8, 9]

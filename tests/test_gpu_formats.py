"""-m gpu: on-disk formats: native safetensors reader, quanto qint8 pairs, LoRA merge.
The checks compare model outputs after ingestion through the format path with outputs after ingestion
of the equivalent in-memory tensors (bit-identical where the arithmetic is the same), and the LoRA merge
against the oracle run on merged weights."""
import os

import pytest
import torch

from oracle import flux2_ref as R
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _fwd(dit, cfg, inp, lat, h, w, s_txt):
    from flux2mi import api
    return dit.forward(lat.cuda(), inp["txt"].cuda(), 0.6, 4.0, torch.from_numpy(api.image_position_ids(h, w)).cuda(),
                       torch.from_numpy(api.text_position_ids(s_txt)).cuda()).cpu()


def _setup(seed=31):
    from flux2mi import synthetic as syn
    cfg = syn.DIT_PRESETS["tiny"]
    P = syn.make_dit_weights(cfg, seed=seed, std=0.05)
    inp = syn.make_inputs(cfg, 64, 64, s_txt=16)
    return syn, cfg, P, inp, R.pack_patchified_to_sequence(inp["noise"])[0]


def _to_diffusers(P, cfg):
    """BFL-named synthetic weights re-keyed to the diffusers scheme (split q/k/v, swapped adaLN halves)."""
    d = cfg["dim"]
    D = {}
    ren = {"img_in.weight": "x_embedder.weight", "txt_in.weight": "context_embedder.weight",
           "time_in.in_layer.weight": "time_guidance_embed.timestep_embedder.linear_1.weight",
           "time_in.out_layer.weight": "time_guidance_embed.timestep_embedder.linear_2.weight",
           "double_stream_modulation_img.lin.weight": "double_stream_modulation_img.linear.weight",
           "double_stream_modulation_txt.lin.weight": "double_stream_modulation_txt.linear.weight",
           "single_stream_modulation.lin.weight": "single_stream_modulation.linear.weight",
           "final_layer.linear.weight": "proj_out.weight"}
    for k, v in P.items():
        if k in ren:
            D[ren[k]] = v
        elif k == "final_layer.adaLN_modulation.1.weight":
            D["norm_out.linear.weight"] = torch.cat([v[d:], v[:d]], 0)
        elif k.startswith("double_blocks."):
            i = k.split(".")[1]
            t = f"transformer_blocks.{i}."
            for s, (q, kk, vv, o, nq, nk, ff) in {"img": ("to_q", "to_k", "to_v", "to_out.0", "norm_q", "norm_k", "ff"),
                                                  "txt": ("add_q_proj", "add_k_proj", "add_v_proj", "to_add_out", "norm_added_q", "norm_added_k", "ff_context")}.items():
                if k.endswith(f"{s}_attn.qkv.weight"):
                    D[t + f"attn.{q}.weight"], D[t + f"attn.{kk}.weight"], D[t + f"attn.{vv}.weight"] = v[:d], v[d:2 * d], v[2 * d:]
                elif k.endswith(f"{s}_attn.proj.weight"):
                    D[t + f"attn.{o}.weight"] = v
                elif k.endswith(f"{s}_attn.norm.query_norm.scale"):
                    D[t + f"attn.{nq}.weight"] = v
                elif k.endswith(f"{s}_attn.norm.key_norm.scale"):
                    D[t + f"attn.{nk}.weight"] = v
                elif k.endswith(f"{s}_mlp.0.weight"):
                    D[t + f"{ff}.linear_in.weight"] = v
                elif k.endswith(f"{s}_mlp.2.weight"):
                    D[t + f"{ff}.linear_out.weight"] = v
        elif k.startswith("single_blocks."):
            i = k.split(".")[1]
            t = f"single_transformer_blocks.{i}.attn."
            tail = k.split(".", 2)[2]
            D[t + {"linear1.weight": "to_qkv_mlp_proj.weight", "linear2.weight": "to_out.weight",
                   "norm.query_norm.scale": "norm_q.weight", "norm.key_norm.scale": "norm_k.weight"}[tail]] = v
    return D


def test_safetensors_bfl_and_diffusers_prefixed(flux2mi, ctx, tmp_path):
    """The native reader on a BFL-keyed bf16 file and on a diffusers-keyed f16/f32 file with the
    `transformer.` prefix and foreign tensors: same output as in-memory ingestion (bit-identical for
    bf16; the f16 file carries bf16-exact values, so identical too)."""
    from safetensors.torch import save_file
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup()
    base = DiT(ctx, cfg).load(P).finalize()
    ref = _fwd(base, cfg, inp, lat, 64, 64, 16)
    base.close()
    f1 = str(tmp_path / "bfl.safetensors")
    save_file({k: v.contiguous() for k, v in P.items()}, f1, metadata={"format": "pt"})
    d1 = DiT(ctx, cfg)
    n, skipped = d1.load_safetensors(f1)
    assert n == len(P) and skipped == 0
    assert torch.equal(_fwd(d1.finalize(), cfg, inp, lat, 64, 64, 16), ref)
    d1.close()
    D = _to_diffusers(P, cfg)
    tens = {"transformer." + k: (v.to(torch.float16) if i % 2 else v.float()).contiguous() for i, (k, v) in enumerate(sorted(D.items()))}
    tens["text_encoder.embed.weight"] = torch.zeros((4, 4))
    tens["transformer.something.input_scale"] = torch.ones((1,))
    f2 = str(tmp_path / "diffusers.safetensors")
    save_file(tens, f2)
    d2 = DiT(ctx, cfg)
    n, skipped = d2.load_safetensors(f2)
    assert n == len(D) and skipped == 1
    out = _fwd(d2.finalize(), cfg, inp, lat, 64, 64, 16)
    d2.close()
    # f16 cannot hold every bf16 value exactly (exponent range), so allow rounding-level differences
    assert rel_l2(out, ref) < 2e-3


def test_safetensors_errors(flux2mi, ctx, tmp_path):
    from flux2mi import Flux2Error
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup()
    d = DiT(ctx, cfg)
    with pytest.raises(Flux2Error):
        d.load_safetensors(str(tmp_path / "missing.safetensors"))
    bad = tmp_path / "bad.safetensors"
    bad.write_bytes(b"\x10\x00\x00\x00\x00\x00\x00\x00{not json at all}")
    with pytest.raises(Flux2Error):
        d.load_safetensors(str(bad))
    with pytest.raises(Flux2Error):   # nothing loaded -> finalize names the first missing tensor
        d.finalize()
    d.close()


def test_quanto_qint8_pairs(flux2mi, ctx, tmp_path):
    """quanto `._data` / `._scale` pairs: weight = int8 * scale (WeightLoader.swift:347-363).  Checked
    against in-memory ingestion of the same de-quantised values (bit-identical) and, loosely, against
    the unquantised model (int8 per-row quantisation: ~1e-2)."""
    from safetensors.torch import save_file
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup()
    tens, deq = {}, {}
    for k, v in P.items():
        if v.dim() == 2 and v.shape[0] >= 128 and "norm" not in k:
            sc = (v.float().abs().amax(dim=1, keepdim=True) / 127.0).clamp_min(1e-12)
            q = torch.round(v.float() / sc).clamp(-127, 127).to(torch.int8)
            tens["transformer." + k + "._data"] = q
            tens["transformer." + k + "._scale"] = sc.float()
            deq[k] = (q.float() * sc).to(torch.bfloat16)
        else:
            tens["transformer." + k] = v.contiguous()
            deq[k] = v
    path = str(tmp_path / "qint8.safetensors")
    save_file(tens, path)
    a = DiT(ctx, cfg)
    n, skipped = a.load_safetensors(path)
    assert n == len(P) and skipped == 0
    out_q = _fwd(a.finalize(), cfg, inp, lat, 64, 64, 16)
    a.close()
    b = DiT(ctx, cfg).load(deq).finalize()
    assert torch.equal(out_q, _fwd(b, cfg, inp, lat, 64, 64, 16))
    b.close()
    c = DiT(ctx, cfg).load(P).finalize()
    assert rel_l2(out_q, _fwd(c, cfg, inp, lat, 64, 64, 16)) < 3e-2
    c.close()


def test_lora_merge_vs_oracle(flux2mi, ctx, tmp_path):
    """`loadLoRA` + merge: W += (alpha/rank) * scale * B A on layers of every layout class (fused q/k/v
    via diffusers keys, SwiGLU-interleaved mlp, single-block linear1, plain proj) vs the oracle run on
    weights merged in f32; rel-L2 <= 2e-2 and the merge must matter (> 1e-2 away from the base model)."""
    from safetensors.torch import save_file
    from flux2mi.pipeline import Flux2Pipeline
    syn, cfg, P, inp, lat = _setup()
    d, rank, alpha = cfg["dim"], 8, 16.0
    g = torch.Generator().manual_seed(3)
    targets = {  # LoRA layer path (diffusers naming) -> (BFL key, row slice)
        "transformer_blocks.0.attn.to_k": ("double_blocks.0.img_attn.qkv.weight", slice(d, 2 * d)),
        "transformer_blocks.0.attn.add_q_proj": ("double_blocks.0.txt_attn.qkv.weight", slice(0, d)),
        "transformer_blocks.0.ff.linear_in": ("double_blocks.0.img_mlp.0.weight", slice(None)),
        "transformer_blocks.0.attn.to_out.0": ("double_blocks.0.img_attn.proj.weight", slice(None)),
        "single_transformer_blocks.1.attn.to_qkv_mlp_proj": ("single_blocks.1.linear1.weight", slice(None)),
        "single_transformer_blocks.0.attn.to_out": ("single_blocks.0.linear2.weight", slice(None)),
        "not_a_layer_of_this_model.proj": (None, None),
    }
    lora, Pm = {}, {k: v.float().clone() for k, v in P.items()}
    user_scale = 0.8
    for layer, (key, rows) in targets.items():
        out_f, in_f = (64, 64) if key is None else (Pm[key][rows].shape[0], Pm[key].shape[1])
        A = (torch.randn((rank, in_f), generator=g) * 0.1).to(torch.bfloat16)
        B = (torch.randn((out_f, rank), generator=g) * 0.1).to(torch.bfloat16)
        lora[f"transformer.{layer}.lora_A.weight"] = A
        lora[f"transformer.{layer}.lora_B.weight"] = B
        if key is not None:
            Pm[key][rows] += (alpha / rank) * user_scale * (B.float() @ A.float())
    path = str(tmp_path / "lora.safetensors")
    save_file(lora, path, metadata={"lora_alpha": str(alpha), "lora_rank": str(rank)})
    vcfg = syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, syn.make_vae_weights(vcfg))
    base = _fwd(pipe.transformer, cfg, inp, lat, 64, 64, 16)
    assert not pipe.hasLoRA
    assert pipe.load_lora(path, scale=user_scale) == 6
    assert pipe.hasLoRA
    out = _fwd(pipe.transformer, cfg, inp, lat, 64, 64, 16)
    ref = R.dit_forward(Pm, cfg, lat.unsqueeze(0), inp["txt"].float(), torch.tensor([0.6]), None,
                        R.image_position_ids(64, 64), R.text_position_ids(16))[0]
    assert rel_l2(out, ref) < 2e-2
    assert rel_l2(out, base) > 1e-2


def test_vae_safetensors(flux2mi, ctx, tmp_path):
    from safetensors.torch import save_file
    from flux2mi import synthetic as syn
    from flux2mi.api import VAE
    vcfg = syn.VAE_PRESETS["tiny"]
    VP = syn.make_vae_weights(vcfg)
    path = str(tmp_path / "vae.safetensors")
    tens = {k: v.contiguous() for k, v in VP.items()}
    tens["bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    save_file(tens, path)
    a = VAE(ctx, vcfg)
    n, skipped = a.load_safetensors(path)
    assert n == len(VP)
    a.finalize()
    b = VAE(ctx, vcfg).load(VP).finalize()
    z = torch.randn((1, 32, 8, 8), generator=torch.Generator().manual_seed(2)).cuda()
    assert torch.equal(a.decode(z)[1], b.decode(z)[1])
    assert a.has_encoder
    a.close(); b.close()


@pytest.mark.parametrize("weights", ["fp8", "bf16"])
def test_prequantized_checkpoint_round_trip(flux2mi, ctx, tmp_path, weights):
    """Flux2PrequantizedCheckpoint: export the resident weights in the library's own layout, reload into a fresh
    model -> bit-identical forward; the file is a valid safetensors file with the documented metadata and, for fp8,
    about half the bytes of the bf16 one."""
    import json, struct
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup(seed=41)
    a = DiT(ctx, cfg, weights=weights).load(P).finalize()
    ref = _fwd(a, cfg, inp, lat, 64, 64, 16)
    path = str(tmp_path / f"transformer_{weights}.safetensors")
    a.export_prequantized(path, source_fingerprint="ckpt.safetensors:123:456")
    a.close()
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]            # temp file renamed away
    raw = open(path, "rb").read()
    hlen = struct.unpack("<Q", raw[:8])[0]
    hdr = json.loads(raw[8:8 + hlen])
    md = hdr.pop("__metadata__")
    assert md["format"] == "flux2mi-prequantized-v1" and md["quantization"] == weights and md["lora_baked"] == "false"
    assert md["source_fingerprint"] == "ckpt.safetensors:123:456" and f"dim={cfg['dim']}" in md["config"]
    assert 8 + hlen + max(t["data_offsets"][1] for t in hdr.values()) == len(raw)
    dtypes = {t["dtype"] for t in hdr.values()}
    assert dtypes == ({"F8_E4M3", "F32", "BF16"} if weights == "fp8" else {"BF16"})
    b = DiT(ctx, cfg, weights=weights)
    assert b.load_prequantized(path, expected_fingerprint="ckpt.safetensors:123:456") is False
    assert torch.equal(_fwd(b, cfg, inp, lat, 64, 64, 16), ref)              # finalized by the load, same bits
    b.close()
    if weights == "fp8":
        other = str(tmp_path / "bf16.safetensors")
        c = DiT(ctx, cfg, weights="bf16").load(P).finalize()
        c.export_prequantized(other)
        c.close()
        assert os.path.getsize(path) < 0.75 * os.path.getsize(other)


def test_prequantized_checkpoint_safety(flux2mi, ctx, tmp_path):
    """The load validates before it mutates: wrong quantization / config / fingerprint, a truncated payload, a missing
    or foreign tensor are all refused, and the refused model still takes the standard load path to the right answer."""
    import json, struct
    from flux2mi.api import DiT
    from flux2mi.pipeline import Flux2Pipeline
    syn, cfg, P, inp, lat = _setup(seed=43)
    a = DiT(ctx, cfg, weights="fp8").load(P).finalize()
    ref = _fwd(a, cfg, inp, lat, 64, 64, 16)
    good = str(tmp_path / "good.safetensors")
    a.export_prequantized(good, "fp-A")
    with pytest.raises(flux2mi.Flux2Error):
        DiT(ctx, cfg, weights="fp8").export_prequantized(str(tmp_path / "x.safetensors"))     # not finalized
    a.close()
    raw = open(good, "rb").read()
    hlen = struct.unpack("<Q", raw[:8])[0]

    def variant(name, mutate_header=None, cut=0):
        hdr = json.loads(raw[8:8 + hlen])
        if mutate_header:
            mutate_header(hdr)
        h = json.dumps(hdr, separators=(",", ":")).encode()
        h += b" " * (-len(h) % 8)
        body = raw[8 + hlen:]
        p = str(tmp_path / name)
        open(p, "wb").write(struct.pack("<Q", len(h)) + h + (body[:-cut] if cut else body))
        return p

    def drop_one(h):
        h.pop(next(k for k in h if k.endswith(".scale")))

    def add_one(h):
        k = next(k for k in h if k.endswith(".scale"))
        h["foreign.weight"] = dict(h[k])

    def reshape_one(h):
        k = next(k for k in h if k.endswith(".scale"))
        h[k]["shape"] = [h[k]["shape"][0] // 2, 2]

    bad = {
        "truncated": variant("trunc.safetensors", cut=4096),
        "format": variant("fmt.safetensors", lambda h: h["__metadata__"].update(format="something-else")),
        "config": variant("cfg.safetensors", lambda h: h["__metadata__"].update(config="dim=1")),
        "missing": variant("missing.safetensors", drop_one),
        "foreign": variant("foreign.safetensors", add_one),
        "shape": variant("shape.safetensors", reshape_one),
    }
    m = DiT(ctx, cfg, weights="fp8")
    for why, p in bad.items():
        with pytest.raises(flux2mi.Flux2Error):
            m.load_prequantized(p)
    with pytest.raises(flux2mi.Flux2Error):
        m.load_prequantized(good, expected_fingerprint="fp-B")                 # exported from other source weights
    with pytest.raises(flux2mi.Flux2Error):
        DiT(ctx, cfg, weights="bf16").load_prequantized(good)                  # quantization mismatch
    with pytest.raises(flux2mi.Flux2Error):
        m.load_prequantized(str(tmp_path / "does_not_exist.safetensors"))
    # the refused model is untouched: the standard path still gives the reference answer, and so does the good file
    assert torch.equal(_fwd(m.load(P).finalize(), cfg, inp, lat, 64, 64, 16), ref)
    m.close()
    # pipeline-level rule (Flux2Pipeline.swift:763-766): bf16 configurations do not export
    pipe = Flux2Pipeline(model=cfg, vae_variant=syn.VAE_PRESETS["tiny"]).load_models(P, syn.make_vae_weights(syn.VAE_PRESETS["tiny"]))
    with pytest.raises(flux2mi.Flux2Error):
        pipe.exportPrequantizedTransformer(str(tmp_path / "no.safetensors"))
    pipe8 = Flux2Pipeline(model=cfg, vae_variant=syn.VAE_PRESETS["tiny"], quantization="fp8").load_models(
        P, syn.make_vae_weights(syn.VAE_PRESETS["tiny"]))
    out = pipe8.exportPrequantizedTransformer(str(tmp_path / "pipe.safetensors"), "fp-A")
    assert os.path.getsize(out) == len(raw)


def test_prequantized_checkpoint_lora_tag(flux2mi, ctx, tmp_path):
    """An export taken after a LoRA merge is tagged and reported at load (PrequantizedCheckpoint.swift, LoRA-baked exports)."""
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup(seed=47)
    a = DiT(ctx, cfg).load(P)
    name = "double_blocks.0.img_attn.proj.weight"
    g = torch.Generator().manual_seed(1)
    a.merge_lora(name, torch.randn((4, P[name].shape[1]), generator=g) * 0.05, torch.randn((P[name].shape[0], 4), generator=g) * 0.05, 0.8)
    a.finalize()
    ref = _fwd(a, cfg, inp, lat, 64, 64, 16)
    path = str(tmp_path / "baked.safetensors")
    a.export_prequantized(path)
    a.close()
    b = DiT(ctx, cfg)
    assert b.load_prequantized(path) is True
    assert torch.equal(_fwd(b, cfg, inp, lat, 64, 64, 16), ref)
    b.close()


# ---- round 2: hostile / inconsistent headers, slot coverage, LoRA prefixes + scheduler overrides -----------------
def _write_raw_safetensors(path, header: dict, payload: bytes):
    import json
    import struct
    h = json.dumps(header).encode()
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)) + h + payload)


@pytest.mark.parametrize("case", ["negative_offsets", "overflowing_offset", "span_shorter_than_shape", "span_longer_than_shape",
                                  "q8_data_span", "q8_scale_span"])
def test_safetensors_header_is_validated(flux2mi, ctx, tmp_path, case):
    """A corrupt or hostile header must be refused before any byte is read through it: negative / overflowing
    data_offsets (they used to wrap past the bounds check), and a byte span that does not equal prod(shape) x dtype size
    (set_tensor copies shape-many bytes, whatever the span says).  The model stays usable afterwards."""
    from flux2mi import Flux2Error
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup()
    d = cfg["dim"]
    name = "double_blocks.0.img_attn.proj.weight"
    good = P[name].contiguous().view(torch.uint8).numpy().tobytes()
    n = len(good)
    hdr = {name: {"dtype": "BF16", "shape": [d, d], "data_offsets": [0, n]}}
    payload = good
    if case == "negative_offsets":
        hdr[name]["data_offsets"] = [-16, -8]
    elif case == "overflowing_offset":
        hdr[name]["data_offsets"] = [0, 2 ** 64 - 8]
    elif case == "span_shorter_than_shape":
        hdr[name]["data_offsets"] = [0, n // 2]
    elif case == "span_longer_than_shape":
        hdr[name]["shape"] = [d, d // 2]
    else:
        q = torch.zeros((d, d), dtype=torch.int8).numpy().tobytes()
        sc = torch.ones((d,), dtype=torch.float32).numpy().tobytes()
        key = name[:-len(".weight")] + ".weight"
        hdr = {key + "._data": {"dtype": "I8", "shape": [d, d], "data_offsets": [0, len(q) - (d if case == "q8_data_span" else 0)]},
               key + "._scale": {"dtype": "F32", "shape": [d], "data_offsets": [len(q), len(q) + len(sc) - (4 if case == "q8_scale_span" else 0)]}}
        payload = q + sc
    path = str(tmp_path / f"{case}.safetensors")
    _write_raw_safetensors(path, hdr, payload)
    dit = DiT(ctx, cfg).load(P).finalize()
    before = _fwd(dit, cfg, inp, lat, 64, 64, 16)
    with pytest.raises(Flux2Error):
        dit.load_safetensors(path)
    dit.finalize()
    assert torch.equal(_fwd(dit, cfg, inp, lat, 64, 64, 16), before)
    dit.close()


def test_duplicate_slot_does_not_stand_in_for_missing_rows(flux2mi, ctx):
    """`to_q` loaded three times covers d rows of the fused [3d, d] qkv buffer, not 3d: finalize must still report the
    tensor missing (it used to count rows and pass with k / v never written); BFL + diffusers aliases of the same
    tensor in one load are fine."""
    from flux2mi import Flux2Error
    from flux2mi.api import DiT
    syn, cfg, P, inp, lat = _setup()
    d = cfg["dim"]
    qkv = P.pop("double_blocks.0.img_attn.qkv.weight")
    dit = DiT(ctx, cfg).load(P)
    for _ in range(3):
        dit.load({"transformer_blocks.0.attn.to_q.weight": qkv[:d].contiguous()})
    with pytest.raises(Flux2Error) as e:
        dit.finalize()
    assert "dbl0.img.qkv" in str(e.value)
    dit.load({"transformer_blocks.0.attn.to_k.weight": qkv[d:2 * d].contiguous(),
              "transformer_blocks.0.attn.to_v.weight": qkv[2 * d:].contiguous()}).finalize()
    ref = DiT(ctx, cfg).load({**P, "double_blocks.0.img_attn.qkv.weight": qkv}).finalize()
    ref.load({"transformer_blocks.0.attn.to_q.weight": qkv[:d].contiguous()}).finalize()     # alias on top: still complete
    assert torch.equal(_fwd(dit, cfg, inp, lat, 64, 64, 16), _fwd(ref, cfg, inp, lat, 64, 64, 16))
    dit.close(); ref.close()


def test_lora_prefixes_overrides_and_no_match(flux2mi, ctx, tmp_path):
    """LoRALoader.swift:132-140 strips `base_model.model.`, `transformer.`, `diffusion_model.` in sequence (a ComfyUI-style
    file uses the last); a file that matches NO layer is an error and leaves hasLoRA false; LoRAConfig scheduler overrides
    become activeSchedulerOverrides, drive the schedule (custom sigmas) and are cleared with the last LoRA
    (Flux2Pipeline.swift:618-689, LoRAConfig.swift:7-35)."""
    from safetensors.torch import save_file
    from flux2mi import Flux2Error
    from flux2mi.pipeline import Flux2Pipeline, LoRAConfig, SchedulerOverrides
    syn, cfg, P, inp, lat = _setup()
    g = torch.Generator().manual_seed(4)
    key = "single_blocks.0.linear2.weight"
    A = (torch.randn((4, P[key].shape[1]), generator=g) * 0.1).to(torch.bfloat16)
    B = (torch.randn((P[key].shape[0], 4), generator=g) * 0.1).to(torch.bfloat16)
    vcfg = syn.VAE_PRESETS["tiny"]
    outs = []
    for prefix in ("transformer.", "diffusion_model.", "base_model.model.transformer."):
        path = str(tmp_path / "l.safetensors")
        save_file({f"{prefix}single_blocks.0.linear2.lora_A.weight": A, f"{prefix}single_blocks.0.linear2.lora_B.weight": B}, path)
        pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, syn.make_vae_weights(vcfg))
        assert pipe.load_lora(path) == 1 and pipe.hasLoRA
        outs.append(_fwd(pipe.transformer, cfg, inp, lat, 64, 64, 16))
        pipe.clearAll()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])

    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, syn.make_vae_weights(vcfg))
    bad = str(tmp_path / "bad.safetensors")
    save_file({"unet.down.0.lora_A.weight": A, "unet.down.0.lora_B.weight": B}, bad)
    with pytest.raises(Flux2Error):
        pipe.load_lora(bad)
    assert not pipe.hasLoRA and pipe.activeSchedulerOverrides is None

    sig = [1.0, 0.55, 0.2]
    cfg_json = str(tmp_path / "turbo.json")
    LoRAConfig(filePath=path, scale=0.5, schedulerOverrides=SchedulerOverrides(customSigmas=sig, numSteps=3, guidance=1.0)).save(cfg_json)
    lc = LoRAConfig.load(cfg_json)
    assert lc.name == "l" and lc.effectiveScale == 0.5 and lc.schedulerOverrides.customSigmas == sig
    assert pipe.loadLoRA(lc) == 1
    assert pipe.activeSchedulerOverrides.numSteps == 3
    # the override drives the schedule: same result as passing the sigmas explicitly, and init latents / strength are ignored
    kw = dict(height=64, width=64, steps=28, precomputed_embeddings=inp["txt"], noise=inp["noise"], skip_decode=True)
    a = pipe.generate_with_result("textToImage", **kw)
    assert a.effectiveSteps == 3
    b = pipe.generate_with_result("textToImage", init_latents=torch.ones((1, 16, 128)), strength=0.3, **kw)
    pipe.unloadLoRA("l")
    assert pipe.activeSchedulerOverrides is None and not pipe.hasLoRA
    c = pipe.generate_with_result("textToImage", custom_sigmas=sig, **kw)
    assert torch.equal(a.latents, b.latents) and torch.equal(a.latents, c.latents)
    pipe.clearAll()


def test_lora_of_another_width_is_a_shape_error_and_changes_nothing(flux2mi, ctx, tmp_path):
    """a LoRA trained for another model width carries the same key names (Klein-9B vs 4B, Dev vs Klein).  The
    factors' shapes come from the FILE and must fit the addressed weight: FLUX2MI_ERR_SHAPE naming the layer (the reference fails in
    `matmul`), no read past the mapped tensor, the weights untouched — for a narrower file (the over-read case), a wider one, a
    rank mismatch between A and B, and through the per-layer entry point as well."""
    from safetensors.torch import save_file
    from flux2mi import Flux2Error
    from flux2mi.pipeline import Flux2Pipeline
    syn, cfg, P, inp, lat = _setup()
    vcfg = syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, syn.make_vae_weights(vcfg))
    base = _fwd(pipe.transformer, cfg, inp, lat, 64, 64, 16)
    key = "single_blocks.0.linear2.weight"
    out_f, in_f = P[key].shape
    assert pipe.transformer.slot_shape(key) == (out_f, in_f)
    g = torch.Generator().manual_seed(5)
    cases = {"narrower": ((4, in_f // 2), (out_f // 2, 4)), "wider": ((4, in_f * 2), (out_f * 2, 4)), "in only": ((4, in_f // 2), (out_f, 4)),
             "out only": ((4, in_f), (out_f + 32, 4))}
    for name, (sa, sb) in cases.items():
        path = str(tmp_path / "w.safetensors")
        save_file({"transformer.single_blocks.0.linear2.lora_A.weight": torch.randn(sa, generator=g).to(torch.bfloat16),
                   "transformer.single_blocks.0.linear2.lora_B.weight": torch.randn(sb, generator=g).to(torch.bfloat16)}, path)
        with pytest.raises(Flux2Error) as e:
            pipe.load_lora(path)
        assert e.value.code == flux2mi._lib.ERR_SHAPE and "single_blocks.0.linear2" in str(e.value), (name, str(e.value))
        assert not pipe.hasLoRA
    with pytest.raises(Flux2Error) as e:     # A and B disagree on the rank
        pipe.transformer.merge_lora(key, torch.zeros((4, in_f)), torch.zeros((out_f, 8)), 1.0)
    assert e.value.code == flux2mi._lib.ERR_SHAPE
    assert torch.equal(_fwd(pipe.transformer, cfg, inp, lat, 64, 64, 16), base)
    pipe.clearAll()


def test_lora_factors_of_different_float_dtypes(flux2mi, ctx, tmp_path):
    """lora_A f32 with lora_B bf16 (and the reverse) is accepted — each factor is converted on its own — and merges to
    the same weights as the all-bf16 file of the same (bf16-representable) values."""
    from safetensors.torch import save_file
    from flux2mi.pipeline import Flux2Pipeline
    syn, cfg, P, inp, lat = _setup()
    vcfg = syn.VAE_PRESETS["tiny"]
    key = "double_blocks.0.img_attn.proj.weight"
    g = torch.Generator().manual_seed(6)
    A = (torch.randn((4, P[key].shape[1]), generator=g) * 0.1).to(torch.bfloat16)
    B = (torch.randn((P[key].shape[0], 4), generator=g) * 0.1).to(torch.bfloat16)
    outs = []
    for da, db in [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16), (torch.float16, torch.float32)]:
        path = str(tmp_path / "m.safetensors")
        save_file({"double_blocks.0.img_attn.proj.lora_A.weight": A.to(da), "double_blocks.0.img_attn.proj.lora_B.weight": B.to(db)}, path)
        pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, syn.make_vae_weights(vcfg))
        assert pipe.load_lora(path) == 1
        outs.append(_fwd(pipe.transformer, cfg, inp, lat, 64, 64, 16))
        pipe.clearAll()
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[2], outs[0]) < 2e-3      # f16 rounds the bf16 values' small ones differently


def test_memory_stats_and_trim(flux2mi, ctx):
    """MLX.Memory.{activeMemory, peakMemory, cacheMemory, clearCache} (Flux2CLI/InpaintCommand.swift:253-271) on the C ABI:
    flux2mi_ctx_memory_stats reports the library's own allocation table — weights count as active exactly (param_bytes), a forward
    grows the workspace ("cache"), peak >= active + cache, trim gives the workspace back and the next forward is bit-identical; from
    inside a generation's hook trim releases nothing; destroying the model returns active to where it was."""
    from flux2mi import api, synthetic as syn
    from flux2mi.pipeline import Flux2Pipeline
    _, cfg, P, inp, lat = _setup()
    ctx.clear_cache()
    m0 = ctx.memory_stats()
    dit = api.DiT(ctx, cfg).load(P).finalize()
    ctx.clear_cache()                                   # staging buffers of the ingestion
    m1 = ctx.memory_stats()
    assert m1["active"] - m0["active"] == dit.param_bytes
    assert m1["cache"] == m0["cache"] and m1["peak"] >= m1["active"]
    out1 = _fwd(dit, cfg, inp, lat, 64, 64, 16)
    m2 = ctx.memory_stats()
    assert m2["cache"] > m1["cache"] and m2["active"] == m1["active"] and m2["peak"] >= m2["active"] + m2["cache"]
    freed = ctx.clear_cache()
    m3 = ctx.memory_stats()
    assert freed == m2["cache"] - m3["cache"] > 0 and m3["cache"] == m1["cache"] and m3["peak"] == m2["peak"]
    ctx.reset_peak_memory()
    assert ctx.memory_stats()["peak"] == m3["active"] + m3["cache"]
    assert torch.equal(_fwd(dit, cfg, inp, lat, 64, 64, 16), out1)
    arr = api.MLXArray(ctx, torch.zeros((1, 64, 128)))          # a flux2mi_array counts as active memory, exactly
    assert ctx.memory_stats()["active"] == m3["active"] + 64 * 128 * 4
    del arr
    dit.close()
    ctx.clear_cache()
    assert ctx.memory_stats()["active"] == m0["active"]

    # a hook that calls clearCache() mid-generation: nothing is released, the result equals the undisturbed run
    vcfg = syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, syn.make_vae_weights(vcfg))
    kw = dict(height=64, width=64, steps=3, precomputed_embeddings=inp["txt"], noise=inp["noise"], return_latents=True)
    ref = pipe.generate_with_result("textToImage", **kw)
    freed_in_hook = []

    def hook(c, latents):
        freed_in_hook.append(pipe.ctx.clear_cache())
        return latents
    got = pipe.generate_with_result("textToImage", on_step=hook, **kw)
    assert freed_in_hook == [0, 0, 0]
    assert torch.equal(got.latents, ref.latents) and torch.equal(torch.as_tensor(got.image), torch.as_tensor(ref.image))
    assert pipe.ctx.clear_cache() > 0                   # after the generation the decoder's and the DiT's workspaces go
    again = pipe.generate_with_result("textToImage", **kw)
    assert torch.equal(again.latents, ref.latents) and torch.equal(torch.as_tensor(again.image), torch.as_tensor(ref.image))
    pipe.clearAll()


def test_trim_from_another_thread_while_calls_run(flux2mi, ctx):
    """MLX.Memory.clearCache() is callable from any thread: a second thread trims in a tight loop while this one runs
    forwards, decodes and a model teardown / rebuild on the same GPU (ctypes drops the GIL around every C call, so the trims really
    run beside them).  Every entry point that touches the workspaces registers as a user of its GPU: a trim beside a running call
    gives nothing back, a call beside a running trim waits — results stay bit-identical and nothing is freed under a kernel."""
    import threading
    from flux2mi import api, synthetic as syn
    _, cfg, P, inp, lat = _setup()
    vcfg = syn.VAE_PRESETS["tiny"]
    dit = api.DiT(ctx, cfg).load(P).finalize()
    vae = api.VAE(ctx, vcfg).load(syn.make_vae_weights(vcfg)).finalize()
    z = torch.randn((1, vcfg["latent"], 16, 16), generator=torch.Generator().manual_seed(3)).cuda()
    ref_f = _fwd(dit, cfg, inp, lat, 64, 64, 16)
    ref_d = vae.decode(z)[1].cpu()
    stop = threading.Event()
    trims = {"n": 0, "freed": 0, "err": None}

    def trimmer():
        try:
            while not stop.is_set():
                trims["freed"] += ctx.clear_cache()
                trims["n"] += 1
        except Exception as e:   # noqa: BLE001
            trims["err"] = e
    th = threading.Thread(target=trimmer)
    th.start()
    try:
        for i in range(40):
            assert torch.equal(_fwd(dit, cfg, inp, lat, 64, 64, 16), ref_f), i
            assert torch.equal(vae.decode(z)[1].cpu(), ref_d), i
            if i % 10 == 9:      # teardown and rebuild beside the trims: a DevBuf is never destroyed under a trim that holds it
                dit.close()
                dit = api.DiT(ctx, cfg).load(P).finalize()
    finally:
        stop.set()
        th.join()
    assert trims["err"] is None, trims["err"]
    assert trims["n"] > 0                               # the trimmer ran beside the calls (how often it won a gap is the box's business)
    # the deterministic half: with the thread joined and no call running, one trim gives the workspaces back and results do not change
    assert torch.equal(_fwd(dit, cfg, inp, lat, 64, 64, 16), ref_f)
    assert ctx.clear_cache() > 0
    assert torch.equal(_fwd(dit, cfg, inp, lat, 64, 64, 16), ref_f)
    dit.close()
    vae.close()

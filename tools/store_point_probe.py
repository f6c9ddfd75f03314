#!/usr/bin/env python3
"""Where does the HIP path differ from the oracle's bf16-store-point emulation (oracle/flux2_ref.py, device_roundings)?  One single block at
Klein-9B width, op by op through flux2mi_op_*, each op fed the ORACLE's input of that op: mismatching bf16 values / rel-L2 per op.
Dev tool (GPU); output -> profiles/r04_*_store_point_probe.txt."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import flux2mi  # noqa: F401
from flux2mi import synthetic as syn
from flux2mi.api import Context, DiT
from oracle import flux2_ref as R

torch.manual_seed(0)
ctx = Context(0)
cfg = dict(dim=4096, heads=32, n_double=1, n_single=1, joint_dim=12288, guidance=False)
d, H = cfg["dim"], cfg["heads"]
P = syn.make_dit_weights(cfg, seed=77, device="cuda")
inp = syn.make_inputs(cfg, 512, 512, s_txt=512)
lat = R.pack_patchified_to_sequence(inp["noise"])
ii, ti = R.image_position_ids(512, 512), R.text_position_ids(512)
Pc = {k: v.cpu() for k, v in P.items()}
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())


def mism(a, b):
    a, b = a.cpu().view(torch.int16), b.cpu().view(torch.int16)
    return float((a != b).float().mean())


taps = {}
with R.device_roundings():
    ref16 = R.dit_forward(Pc, cfg, lat, inp["txt"].float(), torch.tensor([0.6]), None, ii, ti, taps=taps)[0]
    hs = torch.cat([taps["double0.txt"], taps["double0.img"]], 1)          # [1, S, d]
    S = hs.shape[1]
    smod = R.modulation(taps["temb"], Pc["single_stream_modulation.lin.weight"], 1)
    cos, sin = taps["cos"], taps["sin"]
    pre = "single_blocks.0."
    x = R._r16(R.apply_modulation(R.layer_norm_noaffine(hs), smod[0][0], smod[0][1]))
    p = R.block_linear(x, Pc[pre + "linear1.weight"])
    q, k, v = p[..., :d], p[..., d:2 * d], p[..., 2 * d:3 * d]
    gate, up = p[..., 3 * d:3 * d + 3 * d], p[..., 6 * d:]
    qh, kh, vh = R._heads(q, H), R._heads(k, H), R._heads(v, H)
    qh = R.apply_rope(R.rms_norm(qh, Pc[pre + "norm.query_norm.scale"]), cos, sin)
    kh = R.apply_rope(R.rms_norm(kh, Pc[pre + "norm.key_norm.scale"]), cos, sin)
    c = math.log2(math.e) / math.sqrt(128.0)
    q16 = (qh * c).to(torch.bfloat16)          # what the epilogue stores
    k16 = kh.to(torch.bfloat16)
    v16 = vh.to(torch.bfloat16)
    qr, kr = R.fp8_qk(qh, kh, 1.0, 1.0)
    attn = R._unheads(R.sdpa(qr, kr, vh))
    comb = R._store16(torch.cat([attn, F.silu(gate) * up], dim=-1))
    out = hs + R.block_linear(comb, Pc[pre + "linear2.weight"], blocks=True) * smod[0][2].unsqueeze(1)

dev = "cuda"
sh, sc, gt = smod[0][0][0].to(dev).contiguous(), smod[0][1][0].to(dev).contiguous(), smod[0][2][0].to(dev).contiguous()
print(f"S {S}, d {d}")
# 1. LayerNorm + modulate
x_hip = ctx.op_layernorm_modulate(hs[0].to(dev).contiguous(), S, sh, sc, sh, sc)
print(f"layernorm_modulate: bf16 values differing {mism(x_hip, x[0].to(torch.bfloat16)):.2e}, rel-L2 {rel(x_hip.cpu(), x[0]):.2e}")
# 2. q | k projection + RMSNorm + RoPE (fused epilogue), on the oracle's xn
ids = torch.cat([ti, ii], 0).to(torch.int32).to(dev)
W1 = P[pre + "linear1.weight"]
qk_hip = ctx.op_qk_proj_norm_rope(x[0].to(torch.bfloat16).to(dev), W1[:2 * d].contiguous(), P[pre + "norm.query_norm.scale"], P[pre + "norm.key_norm.scale"], ids, q_mul=c)
q_ref = R._unheads(q16.float())[0].to(torch.bfloat16)
k_ref = R._unheads(k16.float())[0].to(torch.bfloat16)
print(f"q (x c) projection+norm+rope: bf16 values differing {mism(qk_hip[:, :d], q_ref):.2e}, rel-L2 {rel(qk_hip[:, :d].cpu(), q_ref):.2e}")
print(f"k projection+norm+rope:       bf16 values differing {mism(qk_hip[:, d:], k_ref):.2e}, rel-L2 {rel(qk_hip[:, d:].cpu(), k_ref):.2e}")
# 3. attention on the oracle's q', k, v
qkv = torch.cat([q_ref, k_ref, R._unheads(v16.float())[0].to(torch.bfloat16)], 1).to(dev).contiguous()
a_hip = ctx.op_attention_prescaled(qkv, H, 40.0)
a_ref = attn[0].to(torch.bfloat16)
print(f"attention: bf16 values differing {mism(a_hip, a_ref):.2e}, rel-L2 {rel(a_hip.cpu(), a_ref):.2e}")
with torch.no_grad():
    a32 = R._unheads(R.sdpa.__wrapped__(qr, kr, vh) if hasattr(R.sdpa, "__wrapped__") else F.scaled_dot_product_attention(qr, kr, vh))[0]
print(f"   (f32 attention of the same q, k, v vs the bf16-P emulation: rel-L2 {rel(a32, attn[0]):.2e}; HIP vs that f32 attention {rel(a_hip.cpu(), a32):.2e})")
# 4. SwiGLU hidden: GEMM with bf16 output of gate | up is not an op; check linear2 instead on the oracle's comb
x_io = hs[0].to(dev).contiguous().clone()
o_hip = ctx.op_gemm_gate_resid(comb[0].to(torch.bfloat16).to(dev), P[pre + "linear2.weight"], gt, x_io)
print(f"linear2 + gate + residual: rel-L2 of the output {rel(o_hip.cpu(), out[0]):.2e}, of the increment {rel(o_hip.cpu() - hs[0], out[0] - hs[0]):.2e}")
# 5. the whole block, teacher-forced
dit = DiT(ctx, cfg).load(P).finalize()
for layer, (x_in, y_ref) in enumerate([(torch.cat([taps["txt_in"], taps["img_in"]], 1)[0], hs[0]), (hs[0], taps["single0"][0])]):
    y = dit.forward_block(layer, x_in.to(dev), inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev)).cpu()
    print(f"block {layer} teacher-forced: rel-L2 of the output {rel(y, y_ref):.2e}, of the increment {rel(y - x_in, y_ref - x_in):.2e}  (|increment| / |input| = {float((y_ref - x_in).norm() / x_in.norm()):.2e})")
    inc, inc_ref = y - x_in, y_ref - x_in
    row_err = (inc - inc_ref).norm(dim=1) / inc_ref.norm(dim=1)
    col_err = (inc - inc_ref).norm(dim=0) / inc_ref.norm(dim=0)
    print(f"   increment error: text rows {rel(inc[:512], inc_ref[:512]):.2e}, image rows {rel(inc[512:], inc_ref[512:]):.2e}; per row median {float(row_err.median()):.2e} max {float(row_err.max()):.2e} "
          f"(row {int(row_err.argmax())}); per column median {float(col_err.median()):.2e} max {float(col_err.max()):.2e} (col {int(col_err.argmax())})")
    err = (inc - inc_ref)
    print(f"   error energy: top 1 % of rows carry {float(err.pow(2).sum(1).topk(15).values.sum() / err.pow(2).sum()):.2f}, top 1 % of columns {float(err.pow(2).sum(0).topk(41).values.sum() / err.pow(2).sum()):.2f}")
out_hip = dit.forward(lat.to(dev), inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev)).cpu()
print(f"whole forward: HIP vs store-point oracle {rel(out_hip, ref16):.2e}")
mo = dit.last_modulation()
with R.device_roundings():
    sin_ref = R.sinusoid_embedding(torch.tensor([0.6]) * 1000.0)[0]
    mi = R.modulation(taps["temb"], Pc["double_stream_modulation_img.lin.weight"], 2)
    mt = R.modulation(taps["temb"], Pc["double_stream_modulation_txt.lin.weight"], 2)
    ppf = R.linear(F.silu(taps["temb"]), Pc["final_layer.adaLN_modulation.1.weight"])[0]
cat = lambda mm: torch.cat([mm[0][0][0], mm[0][1][0], mm[0][2][0]] + ([mm[1][0][0], mm[1][1][0], mm[1][2][0]] if len(mm) > 1 else []))
print(f"modulation: sinusoid {rel(mo['sinusoid'], sin_ref):.2e} (max abs {float((mo['sinusoid'] - sin_ref).abs().max()):.2e}), temb {rel(mo['temb'], taps['temb'][0]):.2e}, "
      f"img {rel(mo['img'], cat(mi)):.2e}, txt {rel(mo['txt'], cat(mt)):.2e}, single {rel(mo['single'], cat(smod)):.2e}, final {rel(mo['final'], ppf):.2e}")
# 6. chained (not teacher-forced): embedders by op, block 0, block 1, then the ORACLE's final layer on the HIP stream
img_in = ctx.op_gemm_bf16(lat[0].to(torch.bfloat16).to(dev), P["img_in.weight"]).cpu()
txt_in = ctx.op_gemm_bf16(inp["txt"][0].to(torch.bfloat16).to(dev), P["txt_in.weight"]).cpu()
print(f"embedders: img_in rel-L2 {rel(img_in, taps['img_in'][0]):.2e}, txt_in {rel(txt_in, taps['txt_in'][0]):.2e}")
y0 = dit.forward_block(0, torch.cat([txt_in, img_in], 0).to(dev), inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev))
y1, f1 = dit.forward_block(1, y0, inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev), with_final=True)
y1, f1 = y1.cpu(), f1.cpu()
print(f"HIP chained (embedders by op -> block 0 -> block 1 -> output layer) vs the HIP forward: {rel(f1, out_hip):.2e}")
o0 = dit.forward_block(0, None, inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev), latents=lat.to(dev)).cpu()
o1, of = dit.forward_block(1, None, inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev), latents=lat.to(dev), with_final=True)
o1, of = o1.cpu(), of.cpu()
print(f"OBSERVED stream of the whole forward: behind block 0 vs chained {rel(o0, y0.cpu()):.2e}, vs oracle {rel(o0, hs[0]):.2e}; behind block 1 vs chained {rel(o1, y1):.2e}, "
      f"vs oracle {rel(o1, taps['single0'][0]):.2e}; prediction vs the plain forward {rel(of, out_hip):.2e}")
dd = o0 - y0.cpu()
x_emb = torch.cat([txt_in, img_in], 0)
print(f"   observed - chained behind block 0: text rows {rel(o0[:512], y0.cpu()[:512]):.2e}, image rows {rel(o0[512:], y0.cpu()[512:]):.2e}; "
      f"top 1 % of rows carry {float(dd.pow(2).sum(1).topk(15).values.sum() / dd.pow(2).sum()):.2f} of the difference, top 1 % of columns {float(dd.pow(2).sum(0).topk(41).values.sum() / dd.pow(2).sum()):.2f}")
# sensitivity of block 0 to its input: the same block on the embedder output with one bf16 ulp of noise on 1e-3 of the elements / a 1e-6 relative perturbation
g = torch.Generator().manual_seed(5)
for eps in (1e-7, 1e-6, 1e-5):
    xp = x_emb * (1.0 + eps * torch.randn(x_emb.shape, generator=g))
    yp = dit.forward_block(0, xp.to(dev), inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev)).cpu()
    print(f"   block 0 on the embedder output perturbed by {eps:.0e} relative: output moves by {rel(yp, y0.cpu()):.2e}")
ya = dit.forward_block(0, x_emb.to(dev), inp["txt"].to(dev), 0.6, 0.0, ii.to(dev), ti.to(dev)).cpu()
print(f"   block 0 twice on identical input: {rel(ya, y0.cpu()):.2e}")
print(f"chained: after block 0 {rel(y0.cpu(), hs[0]):.2e}, after block 1 {rel(y1, taps['single0'][0]):.2e}")
with R.device_roundings():
    img = y1[512:].unsqueeze(0)
    pp = R.linear(F.silu(taps["temb"]), Pc["final_layer.adaLN_modulation.1.weight"])
    fin = R.linear(R._r16(R.apply_modulation(R.layer_norm_noaffine(img), pp[:, :d], pp[:, d:])), Pc["final_layer.linear.weight"])[0]
    img_o = taps["single0"][:, 512:]
    fin_o = R.linear(R._r16(R.apply_modulation(R.layer_norm_noaffine(img_o), pp[:, :d], pp[:, d:])), Pc["final_layer.linear.weight"])[0]
print(f"oracle final layer on the HIP stream vs HIP output {rel(out_hip, fin):.2e}; vs oracle output {rel(fin, ref16):.2e}; oracle final on oracle stream vs oracle output {rel(fin_o, ref16):.2e}")
print(f"|out| {float(ref16.norm()):.3e}  |hs img| {float(img_o.norm()):.3e}")
shf, scf = pp[0, :d].to(dev).contiguous(), pp[0, d:].to(dev).contiguous()
xn_f = ctx.op_layernorm_modulate(y1[512:].to(dev).contiguous(), 1024, shf, scf, shf, scf)
D = ctx.op_gemm_bf16(xn_f, P["final_layer.linear.weight"]).cpu()
print(f"final layer by ops (oracle's shift / scale) on the HIP stream: vs oracle final on the same stream {rel(D, fin):.2e}; vs the HIP forward's output {rel(D, out_hip):.2e}")
# the same with the halves swapped (is the forward reading [scale | shift]?)
xn_s = ctx.op_layernorm_modulate(y1[512:].to(dev).contiguous(), 1024, scf, shf, scf, shf)
Dsw = ctx.op_gemm_bf16(xn_s, P["final_layer.linear.weight"]).cpu()
print(f"   halves swapped: vs the HIP forward's output {rel(Dsw, out_hip):.2e}")
# a second whole forward at another timestep: how much does the output depend on t through the final modulation
out_t2 = dit.forward(lat.to(dev), inp["txt"].to(dev), 0.61, 0.0, ii.to(dev), ti.to(dev)).cpu()
print(f"HIP forward t 0.60 vs 0.61: {rel(out_t2, out_hip):.2e}")

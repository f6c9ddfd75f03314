"""-m gpu: the two exchanges of a data-parallel batch behind the C ABI (csrc/dp.cpp: flux2mi_dp_*, RCCL) as a ONE-RANK communicator
on the box's single GPU — the RCCL library loads, a communicator initialises on the context's device, both collectives and the
max-reduce run ON THE CONTEXT'S STREAM in stream order with a generation (no host synchronisation in between: the all-gather
must see the image bytes flux2mi_generate left in HBM), and the communicator can be finalised and re-created.  The multi-rank
index mapping is covered on CPU by tests/test_parallel_gloo.py; no multi-GPU box is reachable from the builder's side."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dp_one_rank_rccl_collectives_in_stream_order_with_generate(flux2mi):
    from flux2mi import parallel as par
    from flux2mi import synthetic as syn
    from flux2mi._lib import check, lib
    from flux2mi.pipeline import Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg))
    ctx = pipe.ctx
    assert lib().flux2mi_dp_world(ctx.handle) == 1 and lib().flux2mi_dp_rank(ctx.handle) == 0
    # a collective without a communicator is a status, not a crash
    buf = torch.zeros(16, dtype=torch.bfloat16, device="cuda")
    assert lib().flux2mi_dp_broadcast_embeddings(ctx.handle, C.c_void_p(buf.data_ptr()), 16, 0) != 0
    for round_ in range(2):                                   # finalise + re-create
        par.attach_context(ctx)
        assert lib().flux2mi_dp_world(ctx.handle) == 1
        par.collective_stats(reset=True)
        par.collective_timing(round_ == 0)                    # round 0: HIP events around both exchanges; round 1: counted, not timed
        inp = syn.make_inputs(cfg, 128, 128, s_txt=16)
        emb = par.broadcast_text_embeddings(inp["txt"].cuda(), tuple(inp["txt"].shape), torch.device("cuda"))
        assert torch.equal(emb.cpu().view(torch.int16), inp["txt"].view(torch.int16))
        img_dev = torch.zeros((1, 128, 128, 3), dtype=torch.uint8, device="cuda")
        res = pipe.generate_with_result("textToImage", height=128, width=128, steps=4, precomputed_embeddings=emb, noise=inp["noise"],
                                        image_device=img_dev[0], return_latents=True)
        allimgs = par.all_gather_images(img_dev)              # no sync between generate and the gather: stream order
        ref = pipe.generate_with_result("textToImage", height=128, width=128, steps=4, precomputed_embeddings=emb, noise=inp["noise"])
        assert tuple(allimgs.shape) == (1, 128, 128, 3)
        assert np.array_equal(allimgs[0].cpu().numpy(), ref.image) and len(np.unique(ref.image)) > 16
        assert par.all_reduce_max(3.25) == 3.25
        # flux2mi_dp_collective_stats: one call of each kind with its payload; event time only while timing is on
        st = par.collective_stats(reset=True)
        assert st["broadcast"]["calls"] == 1 and st["broadcast"]["bytes"] == inp["txt"].numel() * 2
        assert st["all_gather"]["calls"] == 1 and st["all_gather"]["bytes"] == 128 * 128 * 3
        if round_ == 0:
            assert 0.0 < st["broadcast"]["ms"] < 1000.0 and 0.0 < st["all_gather"]["ms"] < 1000.0
        else:
            assert st["broadcast"]["ms"] == 0.0 and st["all_gather"]["ms"] == 0.0
        assert par.collective_stats()["broadcast"]["calls"] == 0
        # a second communicator on the same context is refused
        ident = (C.c_uint8 * 128)()
        check(lib().flux2mi_dp_unique_id(ident))
        assert lib().flux2mi_dp_init(ctx.handle, 0, 1, ident) != 0
        check(lib().flux2mi_dp_finalize(ctx.handle))
        par._CTX = None
    pipe.clearAll()
    ctx.close()


def test_buffer_helpers_and_array_to_bf16(flux2mi, ctx):
    """flux2mi_buffer_* (what a host without a HIP binding uses) and flux2mi_array_to_bf16 (`precomputedEmbeddings: MLXArray?` -> the
    bf16 buffer of flux2mi_generate_params.txt_emb): round trip and the rounding of torch's own bf16 cast."""
    from flux2mi._lib import check, lib
    from flux2mi.api import MLXArray
    x = torch.randn((1, 40, 384), generator=torch.Generator().manual_seed(3))
    p = C.c_void_p()
    check(lib().flux2mi_buffer_alloc(ctx.handle, x.numel() * 2, C.byref(p)))
    a = MLXArray(ctx, x)                                      # flux2mi_array_from_host
    check(lib().flux2mi_array_to_bf16(ctx.handle, a.handle, p))
    out = torch.empty(x.shape, dtype=torch.bfloat16)
    check(lib().flux2mi_buffer_download(ctx.handle, C.c_void_p(out.data_ptr()), p, x.numel() * 2))
    assert torch.equal(out.view(torch.int16), x.to(torch.bfloat16).view(torch.int16))
    back = torch.empty_like(out)
    check(lib().flux2mi_buffer_upload(ctx.handle, p, C.c_void_p(out.data_ptr()), x.numel() * 2))
    check(lib().flux2mi_buffer_download(ctx.handle, C.c_void_p(back.data_ptr()), p, x.numel() * 2))
    assert torch.equal(back.view(torch.int16), out.view(torch.int16))
    check(lib().flux2mi_buffer_free(ctx.handle, p))

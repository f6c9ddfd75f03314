#!/usr/bin/env python3
"""One rank of tests/test_gpu_dp_ranks.py — a FRESH process per GPU (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment, as
torchrun or `bench.py --gpus N` set them): broadcast -> generate -> all-gather over the library's RCCL communicator, n_local = 2 images
per rank, checked against images this rank generates again on its own.  Prints one JSON line; exit code 0 = every check passed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    import flux2mi  # noqa: F401
    from flux2mi import parallel as par
    from flux2mi import synthetic as syn
    from flux2mi._lib import lib
    from flux2mi.pipeline import Flux2Pipeline

    rank, world, local = par.init_distributed()
    n_local, H, W, steps = 2, 128, 128, 3
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg, device=local).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg))
    par.attach_context(pipe.ctx)                       # ncclCommInitRank across the rank processes, on the generating stream
    assert lib().flux2mi_dp_world(pipe.ctx.handle) == world and lib().flux2mi_dp_rank(pipe.ctx.handle) == rank

    # exchange 1: per-image prompt embeddings exist on rank 0 only
    n_img = n_local * world
    shape = (n_img, 16, cfg["joint_dim"])
    ref_emb = torch.randn(shape, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    emb = par.broadcast_text_embeddings(ref_emb.to(dev) if rank == 0 else None, shape, dev, src=0)
    assert torch.equal(emb.cpu().view(torch.int16), ref_emb.view(torch.int16)), "broadcast embeddings differ from rank 0's"

    def noise_of(i):
        return torch.randn((1, 128, H // 16, W // 16), generator=torch.Generator().manual_seed(42 + i))

    # image i -> rank i mod world: this rank's slot s is global image s * world + rank
    mine = par.shard_indices(n_img, rank, world)
    assert mine == [s * world + rank for s in range(n_local)]
    local_imgs = torch.zeros((n_local, H, W, 3), dtype=torch.uint8, device=dev)
    for s, i in enumerate(mine):
        pipe.generate_with_result("textToImage", height=H, width=W, steps=steps, precomputed_embeddings=emb[i], noise=noise_of(i),
                                  image_device=local_imgs[s])
    allimgs = par.all_gather_images(local_imgs)        # exchange 2, in stream order behind the generations (no host sync)
    assert tuple(allimgs.shape) == (n_img, H, W, 3)
    got = allimgs.cpu().numpy()
    # every rank re-generates ALL images on its own GPU: global order (slot, rank) restored from RCCL's rank-major layout
    levels = 0
    for i in range(n_img):
        ref = pipe.generate_with_result("textToImage", height=H, width=W, steps=steps, precomputed_embeddings=emb[i], noise=noise_of(i)).image
        assert np.array_equal(got[i], ref), f"rank {rank}: gathered image {i} is not global image {i}"
        levels = max(levels, len(np.unique(ref)))
    assert levels > 16, "degenerate images"
    assert len({got[i].tobytes() for i in range(n_img)}) == n_img, "the images of a batch must differ (different seeds / prompts)"
    mx = par.all_reduce_max(10.0 + rank)
    assert mx == 10.0 + world - 1
    par.finalize()
    pipe.clearAll()
    print(json.dumps({"rank": rank, "world": world, "images": n_img, "ok": True}), flush=True)


if __name__ == "__main__":
    main()

"""Shared helpers for the parity tests."""
import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_dev(d, device="cuda"):
    return {k: v.to(device) for k, v in d.items()}

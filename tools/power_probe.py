#!/usr/bin/env python3
"""Is the chip holding its clock down under the GEMM / attention kernels?  Samples the card's power and shader clock
(amdgpu sysfs / rocm-smi, read-only) while a kernel runs back to back for a few seconds.  Dev tool."""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi
from flux2mi.api import Context


def read_first(patterns):
    for p in patterns:
        for f in glob.glob(p):
            try:
                return open(f).read().strip(), f
            except OSError:
                pass
    return None, None


def hwmon_dir():
    """hwmon directory of the GPU this process sees as device 0 (matched by PCI address; a host has eight cards)."""
    try:
        pr = torch.cuda.get_device_properties(0)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        hits = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        if hits:
            return hits[0]
    except Exception as e:
        print("hwmon lookup:", e)
    return None


HW = None


def sample(stop, out):
    global HW
    if HW is None:
        HW = hwmon_dir() or "/sys/class/drm/card*/device/hwmon/hwmon*"
    while not stop.is_set():
        pw, _ = read_first([HW + "/power1_average", HW + "/power1_input"])
        ck, _ = read_first([HW + "/freq1_input"])
        out.append((time.time(), pw, ck))
        time.sleep(0.05)


def run(name, fn, seconds=3.0):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop.set()
    th.join()
    pw = [float(p) / 1e6 for _, p, _ in out if p]
    ck = [float(c) / 1e6 for _, _, c in out if c]
    tail = lambda v: sorted(v[len(v) // 3:])[len(v[len(v) // 3:]) // 2] if v else None
    print(f"{name:34s}: {dt / n * 1e3:7.3f} ms/call | power W median(last 2/3) {tail(pw)} max {max(pw) if pw else None} | sclk MHz median {tail(ck)} min {min(ck) if ck else None}", flush=True)


def main():
    hw = hwmon_dir()
    cap, f = read_first([(hw or "/sys/class/drm/card*/device/hwmon/hwmon*") + "/power1_cap"])
    print("hwmon", hw, "power1_cap", cap)
    try:
        print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout[-1500:])
    except Exception as e:
        print("rocm-smi:", e)
    ctx = Context(0)
    M = N = K = 8192
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    z = torch.zeros_like(a)
    run("gemm 8192^3 random", lambda: ctx.op_gemm_bf16(a, w))
    run("gemm 8192^3 zeros", lambda: ctx.op_gemm_bf16(z, z))
    run("hipBLASLt 8192^3 random", lambda: torch.matmul(a, w.t()))
    a1 = torch.randn((4608, 3072), device="cuda").to(torch.bfloat16)
    w1 = (torch.randn((27648, 3072), device="cuda") * 0.05).to(torch.bfloat16)
    run("gemm linear1 4608x27648x3072", lambda: ctx.op_gemm_bf16(a1, w1))
    a2 = torch.randn((4608, 12288), device="cuda").to(torch.bfloat16)
    w2 = (torch.randn((3072, 12288), device="cuda") * 0.05).to(torch.bfloat16)
    run("gemm linear2 4608x3072x12288 (216)", lambda: ctx.op_gemm_bf16(a2, w2))
    a3 = torch.randn((4096, 12288), device="cuda").to(torch.bfloat16)
    run("gemm 4096x3072x12288 (192 tiles)", lambda: ctx.op_gemm_bf16(a3, w2))
    a4 = torch.randn((2048, 12288), device="cuda").to(torch.bfloat16)
    w4 = (torch.randn((4096, 12288), device="cuda") * 0.05).to(torch.bfloat16)
    run("gemm 2048x4096x12288 (128 tiles)", lambda: ctx.op_gemm_bf16(a4, w4))
    qkv = (torch.randn((4608, 3 * 3072), device="cuda") * 0.7).to(torch.bfloat16)
    run("attention 4608 x 24 (432 items)", lambda: ctx.op_attention(qkv, 24))
    qkv2 = (torch.randn((4096, 3 * 2048), device="cuda") * 0.7).to(torch.bfloat16)
    run("attention 4096 x 16 (256 items)", lambda: ctx.op_attention(qkv2, 16))
    qkv3 = (torch.randn((4096, 3 * 1024), device="cuda") * 0.7).to(torch.bfloat16)
    run("attention 4096 x 8 (128 items)", lambda: ctx.op_attention(qkv3, 8))


if __name__ == "__main__":
    main()

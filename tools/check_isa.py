#!/usr/bin/env python3
"""Invariants of the hand-placed kernels that only the generated ISA can show (no GPU needed; ~3 min of hipcc):

  gemm256s.hip / attention64b.hip keep asynchronous results (ds_read / LDS-DMA / MFMA) in registers the compiler believes ready, count
  their own vmcnt / lgkmcnt, and own M0.  That is only sound while, INSIDE the steady-state loops, hipcc adds nothing of its own:
    * no register copies (v_mov / v_accvgpr_*): a copy of a fragment register could read it before its ds_read has landed
    * no scratch traffic, no vector-memory instruction besides the LDS-DMA pieces (they would shift the counted vmcnt)
    * no M0 reference besides the DMA setup (check_m0 below: the whole object)
    * exactly the instruction mix the schedule tables describe (MFMA / ds_read / DMA counts per trip)
  and the kernels use no scratch at all; in particular no block that issues an MFMA may touch scratch: a spill store of an accumulator
  placed behind the asm MFMA that writes it stores a register whose result is still in flight.
Part of the product build: csrc/Makefile compiles the three objects with -save-temps=obj and runs `check_isa.py --from build` on the
device assembly OF THE OBJECTS THAT ARE LINKED (build/isa.ok is a prerequisite of libflux2mi.so), so a toolchain bump or a source edit
that breaks an invariant fails the build; tests/test_isa_invariants.py re-runs it on the same files.  Stand-alone (compiles its own
assembly, ~5 min): python tools/check_isa.py"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flux-2-swift-mlx_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "isa")


FROM = None   # --from DIR: read <name>-hip-amdgcn-amd-amdhsa-gfx950.s left there by `hipcc -save-temps=obj` instead of compiling


def compile_s(name):
    if FROM:
        return open(os.path.join(FROM, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    os.makedirs(OUT, exist_ok=True)
    dst = os.path.join(OUT, name + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-S", "--offload-device-only",
                    os.path.join(CSRC, name + ".hip"), "-o", dst], check=True)
    return open(dst).read()


def compile_many(names):   # side by side (each is minutes of hipcc)
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(len(names)) as ex:
        return list(ex.map(compile_s, names))


def kernels(text):
    for m in re.finditer(r"^(_Z\S+):\s*;.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        yield m.group(1), m.group(2)


def blocks(body):
    parts = re.split(r"\n(\.LBB\d+_\d+):", body)
    for k in range(1, len(parts), 2):
        ins = [l.split(";")[0].strip() for l in parts[k + 1].split("\n") if l.startswith("\t") and not l.lstrip().startswith((";", "."))]
        ins = [i for i in ins if i]
        # a loop body ends at its backward branch: what follows (a fall-through block hipcc gave no .LBB label) is not part of it
        for n, i in enumerate(ins):
            if i.startswith("s_cbranch") and i.split()[-1] == parts[k]:
                ins = ins[:n + 1]
                break
        yield parts[k], ins


def meta(text, kname, key):
    m = re.search(r"\.name:\s+" + re.escape(kname) + r"\n(?:.*\n)*?\s+\." + key + r":\s+(\d+)", text)
    return int(m.group(1)) if m else None


def check_loop(kname, label, ins, n_mfma, n_dsr, n_dma, dma_op, errors):
    c = collections.Counter(i.split()[0] for i in ins)
    bad = [op for op in c if op.startswith(("v_accvgpr", "scratch_", "global_store", "buffer_store", "global_load_dword", "flat_"))]
    # a v_mov from a literal / SGPR is address arithmetic; from a VGPR / AGPR it is a register copy
    bad += [i for i in ins if i.startswith("v_mov") and re.search(r",\s*[va]\[?\d", i)]
    if dma_op == "buffer_load_dwordx4":
        bad += [op for op in c if op.startswith("global_load")]
    else:
        bad += [op for op in c if op.startswith("buffer_load")]
    m0 = [i for i in ins if re.search(r"\bm0\b", i) and not re.match(r"s_(add_u32|mov_b32) m0, s", i)]
    mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    ok = not bad and not m0 and mf == n_mfma and c.get("ds_read_b128", 0) == n_dsr and c.get(dma_op, 0) == n_dma
    print(f"  {label}: {len(ins)} instructions, {mf} MFMA, {c.get('ds_read_b128', 0)} ds_read_b128, {c.get(dma_op, 0)} LDS-DMA, "
          f"{c.get('s_nop', 0)} s_nop, {sum(v for k, v in c.items() if k.startswith('s_cbranch'))} branch  {'ok' if ok else 'VIOLATION'}")
    if not ok:
        errors.append((kname, label, bad, m0, mf, dict(c)))


def check_m0(name, text, errors):
    """M0 is written by the LDS-DMA asm without save / restore: hipcc must emit no other M0 reference in these objects (an M0 use of the
    compiler's own between an M0 write and its DMA would break the destination address)."""
    bad = [l.strip() for l in text.split("\n") if re.search(r"\bm0\b", l.split(";")[0]) and not re.search(r"s_(add_u32|mov_b32) m0, s", l)]
    print(f"{name}: M0 references besides the LDS-DMA setup: {len(bad)}")
    if bad:
        errors.append((name, "foreign M0 references", bad[:5]))


def main():
    global FROM
    if "--from" in sys.argv:
        FROM = sys.argv[sys.argv.index("--from") + 1]
    errors = []
    # ---- GEMM: steady-state K-tile loop = the block with NI*16 (bf16) / NI*8 (fp8) MFMAs that carries the LDS-DMA pieces
    text = compile_s("gemm256s")
    check_m0("gemm256s", text, errors)
    for kname, body in kernels(text):
        m = re.search(r"gemm256s_kernelILi(\d)ELb(\d)E(?:Lb(\d)E)?(?:Lb(\d)E)?(?:Lb(\d)E)?(?:Lb(\d)E)?(?:Lb(\d)E)?", kname)
        if not m:
            continue
        ni, fp8 = int(m.group(1)), int(m.group(2))
        conv, wide = int(m.group(3) or 0), int(m.group(4) or 0)   # the convolution form; its 512 x 128 tile (16 A + 4 W pieces per wave)
        lean = int(m.group(5) or 0)                               # the convolution's lean epilogue (same K loop)
        grt = int(m.group(6) or 0)                                # the transposed-domain gate-residual epilogue alone (same K loop)
        bt = int(m.group(7) or 0)                                 # the transposed-domain bf16 store alone (same K loop)
        scratch = meta(text, kname, "private_segment_fixed_size")
        spills = meta(text, kname, "vgpr_spill_count")
        print(f"gemm256s_kernel<{ni}, {'fp8' if fp8 else 'bf16'}{', conv' if conv else ''}{', 512x128' if wide else ''}{', lean epilogue' if lean else ''}{', gate-residual epilogue' if grt else ''}{', transposed bf16 store' if bt else ''}>: scratch {scratch} B, spills {spills}")
        if scratch:
            errors.append((kname, "scratch", scratch))
        for label, ins in blocks(body):
            if any(i.startswith("v_mfma") for i in ins) and any(i.startswith("scratch_") for i in ins):
                errors.append((kname, label, "scratch access in a block that issues MFMAs"))
        n_mfma = ni * (8 if fp8 else 16)
        npieces = 20 if wide else (6 if ni <= 6 else 8) + 8
        found = False
        for label, ins in blocks(body):
            c = collections.Counter(i.split()[0] for i in ins)
            if sum(v for k, v in c.items() if k.startswith("v_mfma")) == n_mfma and c.get("global_load_lds_dwordx4", 0) == npieces:
                check_loop(kname, label, ins, n_mfma, (ni + 8) * 2, npieces, "global_load_lds_dwordx4", errors)
                found = True
        if not found:
            errors.append((kname, "steady-state loop not found"))
    # ---- attention: the three-tile loop of the four instantiations (softmax offset: fixed zero / per-query data offset; Q, K: bf16 / e4m3)
    names = ["attention64b", "attention64b_doff", "attention64b_qk8", "attention64b_qk8_doff"]
    texts = compile_many(names)
    for n, t in zip(names, texts):
        check_m0(n, t, errors)
    for text, (kname, body) in [(t, kb) for t in texts for kb in kernels(t)]:
        m = re.search(r"attn64b_kernelILb(\d)E(?:Lb(\d)E)?", kname)
        if not m:
            continue
        doff, qk8 = int(m.group(1)), int(m.group(2) or 0)
        scratch = meta(text, kname, "private_segment_fixed_size")
        print(f"attn64b_kernel<{'data offset' if doff else 'fixed offset'}{', e4m3 Q K' if qk8 else ''}>: scratch {scratch} B, spills {meta(text, kname, 'vgpr_spill_count')}")
        # the data-offset form keeps 16 more VGPRs (the offsets): a few set-up values of the prologue / epilogue go to scratch,
        # never inside a block that issues MFMAs
        if scratch > (64 if doff else 0):
            errors.append((kname, "scratch", scratch))
        for label, ins in blocks(body):
            if any(i.startswith("v_mfma") for i in ins) and any(i.startswith("scratch_") for i in ins):
                errors.append((kname, label, "scratch access in a block that issues MFMAs"))
        # per tile: bf16 64 QK^T + 72 PV / row-sum MFMAs, 32 fragment reads, 8 DMA pieces; e4m3 Q K: 16 block-scaled QK^T MFMAs, 24 reads, 6 pieces
        n_mfma, n_dsr, n_dma = (3 * 88, 3 * 24, 3 * 6) if qk8 else (3 * 136, 3 * 32, 3 * 8)
        found = False
        for label, ins in blocks(body):
            c = collections.Counter(i.split()[0] for i in ins)
            if sum(v for k, v in c.items() if k.startswith("v_mfma")) == n_mfma:
                check_loop(kname, label, ins, n_mfma, n_dsr, n_dma, "buffer_load_dwordx4", errors)
                cc = collections.Counter(i.split()[0] for i in ins)
                if cc.get("v_exp_f32", 0) != 3 * 64 or cc.get("v_cvt_pk_bf16_f32", 0) != 3 * 32 or (qk8 and cc.get("v_mfma_scale_f32_16x16x128_f8f6f4", 0) != 3 * 16):
                    errors.append((kname, "softmax / MFMA instruction counts", cc.get("v_exp_f32"), cc.get("v_cvt_pk_bf16_f32")))
                found = True
        if not found:
            errors.append((kname, "three-tile loop not found"))
    if errors:
        print("FAILED:", *errors, sep="\n  ")
        sys.exit(1)
    print("all ISA invariants hold")


if __name__ == "__main__":
    main()

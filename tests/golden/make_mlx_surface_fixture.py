#!/usr/bin/env python3
"""Writes tests/golden/mlx_surface_used.json: every name of the MLX family (`MLX.`-qualified names, `MLXRandom.*`, `Memory.*`, `eval(`,
`MLXArray(` initialiser forms, members and operators on values typed MLXArray) and every `import` of a module the shim provides, as
USED by the reference's code outside Flux2Core that north_star wants to "drop in unchanged":

  Sources/Flux2Chains/*.swift                      the chains (RePaint inpainting, outpainting, compositing, intent, VLM prompt builder)
  Sources/Flux2CLI/Flux2CLI.swift                  the t2i / i2i commands
  Sources/Flux2CLI/{InpaintCommand,OutpaintCommand}.swift + the parsing helpers they share

It is interface data — names, labels, operand types, file:line of each use; no source text.  tests/test_swift_mlx_surface.py requires
integration/swift/Sources/{MLX,MLXRandom}/ to declare each entry and Package.swift to carry a product per imported module, and
re-derives this file from /root/reference where the reference is present (this container only).

Excluded CLI files, and why (SURVEY §2 OUT OF SCOPE rows): CompareEncoders / TestQwen35 / TestVLGeneration / VLMTest (text-encoder
and VLM comparison tools), TrainLoRACommand / TrainingConfigYAML / TrainingControlCommand / EvaluateLoRA (training), ExportQuantizedCommand
and ProfileCommand (call only Flux2Pipeline methods covered by flux2pipeline_public_api.json; no MLX name), MaskSubjectCommand (Apple
Vision)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.swift_mlx_usage import summarize, used_surface  # noqa: E402

REF = "/root/reference/Sources"
CLI_FILES = ["Flux2CLI.swift", "InpaintCommand.swift", "OutpaintCommand.swift", "ModelParsing.swift", "QuantizationParsing.swift", "BeaconOptions.swift"]
EXCLUDED = {"CompareEncoders.swift": "encoder comparison tool", "TestQwen35.swift": "VLM test tool", "TestVLGeneration.swift": "VLM test tool",
            "VLMTest.swift": "VLM test tool", "TrainLoRACommand.swift": "training", "TrainingConfigYAML.swift": "training",
            "TrainingControlCommand.swift": "training", "EvaluateLoRA.swift": "training evaluation",
            "ExportQuantizedCommand.swift": "no MLX name; pipeline method only", "ProfileCommand.swift": "no MLX name; pipeline method only",
            "MaskSubjectCommand.swift": "Apple Vision"}


def typing_context():
    """return types / member types the usage reader needs, from the committed interface fixture of Flux2Pipeline"""
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "flux2pipeline_public_api.json")))["api"]
    rets, members = {}, {}
    for s in api["Flux2Pipeline"]:
        m = re.match(r"(?:@discardableResult )?(?:static )?(\w+)\(.*\)\s*(?:async )?(?:throws )?->\s*(.+)$", s)
        if m:
            rets[m.group(1)] = m.group(2).rstrip("?")
    for s in api["Flux2StepContext"]:
        m = re.match(r"var (\w+): (\w+)", s)
        if m:
            members["Flux2StepContext." + m.group(1)] = m.group(2)
    # Flux2StepHook = @Sendable (Flux2StepContext, MLXArray) -> MLXArray   (Pipeline/Flux2Pipeline.swift:64)
    return rets, {"Flux2StepHook": ["Flux2StepContext", "MLXArray"]}, members


def files():
    out = sorted(glob.glob(os.path.join(REF, "Flux2Chains", "*.swift")))
    out += [os.path.join(REF, "Flux2CLI", f) for f in CLI_FILES]
    return out


def flux2core_types_used():
    """Public types declared in the Flux2Core files that import an MLX module (the files the shim replaces: the MLX-free ones stay in the
    reference's tree), as far as the in-scope callers name them: {type: {"declared_in", "kind", "members": [statically accessed
    `Type.member` / `Type.shared.member` / `.case` names], "used_in": [files]}}.  Interface data: names only."""
    from tests.swift_mlx_usage import strip
    decl = {}
    for f in sorted(glob.glob(os.path.join(REF, "Flux2Core", "**", "*.swift"), recursive=True)):
        src = strip(open(f).read())
        if not re.search(r"^\s*(?:@\w+\s+)*import\s+MLX", src, re.M):
            continue
        for m in re.finditer(r"public\s+(?:final\s+)?(class|struct|enum|protocol|typealias|actor)\s+(\w+)", src):
            decl.setdefault(m.group(2), (os.path.relpath(f, REF), m.group(1)))
    out = {}
    for f in files():
        src = strip(open(f).read())
        src = re.sub(r"^\s*(?:@\w+\s+)*import\s+\w+.*$", "", src, flags=re.M)
        for t, (where, kind) in decl.items():
            if not re.search(r"\b" + t + r"\b", src):
                continue
            e = out.setdefault(t, {"declared_in": where, "kind": kind, "members": set(), "used_in": set()})
            e["used_in"].add(os.path.relpath(f, REF))
            for m in re.finditer(r"\b" + t + r"\.(\w+)(?:\.(\w+))?", src):
                e["members"].add(m.group(1))
                if m.group(1) == "shared" and m.group(2):
                    e["members"].add(m.group(2))
            if t == "Flux2Profiler":      # `let profiler = Flux2Profiler.shared` ... `profiler.start(..)`
                for m in re.finditer(r"\bprofiler\.(\w+)", src):
                    e["members"].add(m.group(1))
    return {t: {"declared_in": e["declared_in"], "kind": e["kind"], "members": sorted(e["members"]), "used_in": sorted(e["used_in"])}
            for t, e in sorted(out.items())}


def build():
    rets, aliases, members = typing_context()
    uses = []
    for f in files():
        uses += used_surface(os.path.relpath(f, REF), open(f).read(), rets, aliases, members)
    return {"files": [os.path.relpath(f, REF) for f in files()], "excluded_cli_files": EXCLUDED,
            "uses": uses, "summary": [list(t) for t in summarize(uses)], "flux2core_types": flux2core_types_used()}


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "mlx_surface_used.json")
    with open(out, "w") as f:
        json.dump({"source": "VincentGourbin/flux-2-swift-mlx v2.4.0, Sources/Flux2Chains + the t2i / i2i / inpaint / outpaint commands of Sources/Flux2CLI",
                   **build()}, f, indent=1)
    print(out)

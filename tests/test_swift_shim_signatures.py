"""CPU: the Swift shim's `Flux2Pipeline` (integration/swift/Sources/Flux2Core/Flux2Pipeline.swift) has EXACTLY the reference's public
surface — every `public func / init / var / let` of `Flux2Pipeline`, `Flux2GenerationResult` and `Flux2StepContext` in
Sources/Flux2Core/Pipeline/{Flux2Pipeline,Flux2Pipeline+ChainHelpers}.swift, with the same argument labels, order, types, default
values, `async` / `throws` and return type — so that Flux2CLI and the chains compile against it unchanged (north_star).
The reference's interface is the committed fixture tests/golden/flux2pipeline_public_api.json (made by make_swift_api_fixture.py);
where the reference is present (this container) the fixture is re-derived from its sources and must be current.
No Swift toolchain exists here: this is a textual interface check, not a compile."""
import json
import os
import re

import pytest

from tests.swift_api import key, public_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "swift", "Sources", "Flux2Core")
SOURCES = os.path.join(ROOT, "integration", "swift", "Sources")
FIXTURE = os.path.join(ROOT, "tests", "golden", "flux2pipeline_public_api.json")
# public members the shim has and the reference does not: the hook through which the host supplies FluxTextEncoders' tokenizer
SHIM_ONLY = {"var promptTokenizer: Flux2PromptTokenizer? {get set}"}


def _shim(type_name):
    decls = []
    for f in sorted(os.listdir(SHIM)):
        if f.endswith(".swift"):
            decls += public_api(open(os.path.join(SHIM, f)).read(), type_name)
    return sorted(key(d) for d in decls)


def _ref():
    return json.load(open(FIXTURE))["api"]


def test_fixture_is_current_with_the_reference():
    if not os.path.isdir("/root/reference/Sources/Flux2Core/Pipeline"):
        pytest.skip("the reference is not on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_swift_api_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert mk.build() == _ref()


@pytest.mark.parametrize("type_name", ["Flux2Pipeline", "Flux2GenerationResult", "Flux2StepContext"])
def test_shim_public_surface_equals_the_reference(type_name):
    ref, shim = set(_ref()[type_name]), set(_shim(type_name))
    missing = sorted(ref - shim)
    extra = sorted(shim - ref - SHIM_ONLY)
    assert not missing, "declared by the reference, absent or different in the shim:\n  " + "\n  ".join(missing)
    assert not extra, "public in the shim, not in the reference:\n  " + "\n  ".join(extra)


def test_the_headline_signatures_by_name():
    """The declarations VERDICT r2 names, spelled out (a parser bug that dropped them from BOTH sides would otherwise pass)."""
    shim = _shim("Flux2Pipeline")
    gw = [s for s in shim if s.startswith("generateWithResult(")]
    assert gw == ["generateWithResult(mode: Flux2GenerationMode, prompt: String, interpretImagePaths: [String]? = nil, height: Int, width: Int, "
                  "steps: Int, guidance: Float, seed: UInt64?, upsamplePrompt: Bool, precomputedEmbeddings: MLXArray? = nil, "
                  "checkpointInterval: Int?, maxReferencePixels: Int = 1024 * 1024, initLatents: MLXArray? = nil, strength: Float = 1.0, "
                  "onProgress: Flux2ProgressCallback?, onCheckpoint: Flux2CheckpointCallback?, onStep: Flux2StepHook? = nil) "
                  "async throws -> Flux2GenerationResult"]
    names = {re.match(r"(?:@discardableResult )?(?:static )?(?:var )?(\w+)", s).group(1) for s in shim}
    for n in ["generateTextToImage", "generateImageToImage", "generateTextToImageWithResult", "generateImageToImageWithResult", "generate",
              "loadModels", "loadLoRA", "unloadLoRA", "unloadAllLoRAs", "hasLoRA", "activeSchedulerOverrides", "exportPrequantizedTransformer",
              "encodeImageToPackedSequence", "packMaskForLatentBlending", "resolveChainDimensions", "clearAll", "hasRequiredModels", "missingModels"]:
        assert n in names, n
    assert sum(s.startswith("generateImageToImage(") for s in shim) == 2          # images: / imageData: overloads


def test_chain_call_site_type_checks_on_paper():
    """Flux2MaskedInpaintingChain.swift:414-431 calls generateWithResult with these labels in this order; Swift requires call-site labels
    to be a subsequence of the declaration's, every skipped parameter having a default."""
    call = ["mode", "prompt", "interpretImagePaths", "height", "width", "steps", "guidance", "seed", "upsamplePrompt", "precomputedEmbeddings",
            "checkpointInterval", "initLatents", "strength", "onProgress", "onCheckpoint", "onStep"]
    src = "".join(open(os.path.join(SHIM, f)).read() for f in sorted(os.listdir(SHIM)) if f.endswith(".swift"))
    decl = [d for d in public_api(src, "Flux2Pipeline") if d["name"] == "generateWithResult"][0]
    it = iter(decl["params"])
    for label in call:
        for p in it:
            if p["label"] == label:
                break
            assert p["default"] is not None, f"call site skips '{p['label']}', which has no default"
        else:
            raise AssertionError(f"label '{label}' not found in order")
    assert decl["effects_and_return"] == "async throws -> Flux2GenerationResult"


def test_every_c_function_the_shim_calls_is_declared():
    hdr = open(os.path.join(ROOT, "include", "flux2mi.h")).read()
    declared = set(re.findall(r"\b(flux2mi_[a-z0-9_]+)\s*\(", hdr)) | set(re.findall(r"\b(flux2mi_[a-z0-9_]+)\b", hdr))
    for mod in sorted(os.listdir(SOURCES)):
        for f in sorted(os.listdir(os.path.join(SOURCES, mod))):
            if f.endswith(".swift"):
                src = re.sub(r"//[^\n]*", "", open(os.path.join(SOURCES, mod, f)).read())
                for name in set(re.findall(r"\b(flux2mi_[a-z0-9_]+)\s*\(", src)):
                    assert name in declared, f"{mod}/{f}: {name} is not in include/flux2mi.h"

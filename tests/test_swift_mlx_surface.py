"""CPU: the WHOLE MLX surface the reference's callers use, under the module names they import.

The chains and the t2i / i2i / inpaint / outpaint commands `import Flux2Core`, `import MLX`, `@preconcurrency import MLXRandom` and use
`MLXRandom.seed / normal`, `eval`, `MLXArray(Float)`, `1 - a`, `a * b`, `a + b`, `.shape`, `MLX.Memory.activeMemory / peakMemory /
cacheMemory / clearCache()` (Flux2MaskedInpaintingChain.swift:17-19, 380-403; InpaintCommand.swift:17, 253-271).  The used surface is
the committed fixture tests/golden/mlx_surface_used.json (made by make_mlx_surface_fixture.py, re-derived from /root/reference where it
is present); this test requires integration/swift/ to provide a library product per imported module and a PUBLIC declaration of the
matching shape for every used name, and the C ABI to export what those declarations call.  Textual — no Swift toolchain here."""
import json
import os
import re

import pytest

from tests.swift_mlx_usage import strip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "integration", "swift")
FIXTURE = os.path.join(ROOT, "tests", "golden", "mlx_surface_used.json")


def _fixture():
    return json.load(open(FIXTURE))


def _module_src(name):
    d = os.path.join(PKG, "Sources", name)
    return "\n".join(strip(open(os.path.join(d, f)).read()) for f in sorted(os.listdir(d)) if f.endswith(".swift"))


def _visible_from(module):
    """source text a file that says `import <module>` sees: the module's own + what it re-exports (`@_exported import X`)"""
    seen, todo, out = set(), [module], []
    while todo:
        m = todo.pop()
        if m in seen or not os.path.isdir(os.path.join(PKG, "Sources", m)):
            continue
        seen.add(m)
        src = _module_src(m)
        out.append(src)
        todo += re.findall(r"@_exported\s+import\s+(\w+)", src)
    return "\n".join(out)


def _enum_body(src, name):
    m = re.search(r"public\s+enum\s+" + name + r"\b[^{]*\{", src)
    assert m, f"no `public enum {name}`"
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.end():i - 1]


def test_fixture_is_current_with_the_reference():
    if not os.path.isdir("/root/reference/Sources/Flux2Chains"):
        pytest.skip("the reference is not on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk_mlx", os.path.join(ROOT, "tests", "golden", "make_mlx_surface_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    now, fx = mk.build(), _fixture()
    assert now["summary"] == fx["summary"] and now["uses"] == fx["uses"] and now["files"] == fx["files"]


def test_fixture_covers_the_call_sites_the_verdict_names():
    where = {u["where"] for u in _fixture()["uses"]}
    for w in ["Flux2Chains/Flux2MaskedInpaintingChain.swift:383", "Flux2Chains/Flux2MaskedInpaintingChain.swift:385",
              "Flux2Chains/Flux2MaskedInpaintingChain.swift:386", "Flux2Chains/Flux2MaskedInpaintingChain.swift:400",
              "Flux2Chains/Flux2MaskedInpaintingChain.swift:401", "Flux2Chains/Flux2MaskedInpaintingChain.swift:402",
              "Flux2CLI/InpaintCommand.swift:253", "Flux2CLI/InpaintCommand.swift:254", "Flux2CLI/InpaintCommand.swift:255",
              "Flux2CLI/InpaintCommand.swift:271"]:
        assert w in where, w


def test_every_imported_module_is_a_library_product_and_target():
    pkg = strip(open(os.path.join(PKG, "Package.swift")).read())
    pkg_raw = open(os.path.join(PKG, "Package.swift")).read()
    for kind, *rest in _fixture()["summary"]:
        if kind != "import":
            continue
        mod = rest[0]
        assert re.search(r'\.library\(\s*name:\s*"' + mod + r'"\s*,\s*targets:\s*\[\s*"' + mod + r'"\s*\]', pkg_raw), f"no library product '{mod}'"
        assert re.search(r'\.target\(\s*name:\s*"' + mod + r'"', pkg_raw), f"no target '{mod}'"
        assert os.path.isdir(os.path.join(PKG, "Sources", mod)), f"Sources/{mod} missing"
    assert pkg  # parsed


@pytest.mark.parametrize("entry", [tuple(e) for e in json.load(open(FIXTURE))["summary"] if e[0] != "import"], ids=lambda e: " ".join(e))
def test_the_facade_declares_it(entry):
    kind = entry[0]
    mlx = _visible_from("MLX")
    if kind == "qualified":
        _, name, form = entry
        parts = name.split(".")
        if parts[0] == "MLX":                       # module-qualified: MLX.Memory.x -> enum Memory in module MLX
            src, parts = _module_src("MLX"), parts[1:]
        elif parts[0] == "MLXRandom":               # `import MLXRandom` + MLXRandom.f(..): the namespace enum, visible through that import
            src = _visible_from("MLXRandom")
        else:
            src = mlx
        body = _enum_body(src, parts[0])
        member = parts[1]
        if form == "property":
            assert re.search(r"public\s+static\s+var\s+" + member + r"\s*:", body), f"{name}: no public static var"
        else:
            labels = form[5:-1].split(",") if form[5:-1] else []
            m = re.search(r"public\s+static\s+func\s+" + member + r"\s*\(([^)]*)\)", body)
            assert m, f"{name}: no public static func"
            params = [p.strip() for p in m.group(1).split(",") if p.strip()]
            assert len(params) == len(labels), f"{name}: {len(params)} parameters, the call site passes {len(labels)}"
            for p, l in zip(params, labels):
                assert p.split(":")[0].split()[0] == l, f"{name}: parameter label '{p}' vs call-site label '{l}'"
    elif kind == "property_arith":
        _, name, op, other = entry
        body = _enum_body(_module_src("MLX"), name.split(".")[1])
        m = re.search(r"public\s+static\s+var\s+" + name.split(".")[2] + r"\s*:\s*(\w+)", body)
        assert m and m.group(1) in ("Int", "Int64", "UInt64", "UInt"), f"{name} {op} {other}: needs an integer property, found {m and m.group(1)}"
    elif kind == "func":
        _, name, argt = entry
        m = re.search(r"public\s+func\s+" + name + r"\s*\(\s*_\s+\w+\s*:\s*([\w.\[\]]+)", mlx)
        assert m and m.group(1).rstrip(".") in (argt, argt + "..."), f"no public func {name}(_: {argt}...)"
    elif kind == "init":
        _, typ, form = entry
        label, argt = form.split(":")
        assert re.search(r"public\s+(?:convenience\s+)?init\(\s*" + label + r"\s+\w+\s*:\s*" + argt + r"\s*\)", mlx), f"no public init({label} : {argt}) on {typ}"
    elif kind == "member":
        _, typ, name, form = entry
        assert re.search(r"public\s+(?:var|let)\s+" + name + r"\s*:", mlx), f"no public var {name} on {typ}"
    elif kind == "operator":
        _, lhs, op, rhs = entry
        swift_t = {"MLXArray": ["MLXArray"], "IntLiteral": ["Int"], "FloatLiteral": ["Float", "Double"], "Float": ["Float"], "Int": ["Int"]}
        ok = any(re.search(r"public\s+func\s+" + re.escape(op) + r"\s*\(\s*\w+\s*:\s*" + lt + r"\s*,\s*\w+\s*:\s*" + rt + r"\s*\)\s*->\s*MLXArray", mlx)
                 for lt in swift_t[lhs] for rt in swift_t[rhs])
        assert ok, f"no public func {op} ({lhs}, {rhs}) -> MLXArray"
    else:
        raise AssertionError(f"unknown fixture entry {entry}")


def test_the_memory_facade_is_backed_by_the_c_abi():
    """real numbers from the library's allocation table, not constants: the façade calls flux2mi_ctx_memory_stats / _trim, the header
    declares them, the library exports them (tests/test_capi_symbols.py checks every header symbol against the .so)."""
    body = _enum_body(_module_src("MLX"), "Memory")
    for fn in ["flux2mi_ctx_memory_stats", "flux2mi_ctx_trim"]:
        assert fn in body, fn
        assert re.search(r"\bint\s+" + fn + r"\s*\(", open(os.path.join(ROOT, "include", "flux2mi.h")).read()), fn


def test_mlxrandom_module_reexports_mlx():
    assert re.search(r"@_exported\s+import\s+MLX\b", _module_src("MLXRandom"))
    assert re.search(r"public\s+enum\s+MLXRandom\b", _module_src("MLX"))


def _flux2core_shim_src():
    return _module_src("Flux2Core")


@pytest.mark.parametrize("type_name", sorted(json.load(open(FIXTURE))["flux2core_types"]))
def test_flux2core_types_the_callers_name_are_declared(type_name):
    """Every public type of the Flux2Core files the shim REPLACES (the ones that import an MLX module; the MLX-free files stay in the
    reference's tree) that the chains or the t2i / i2i / inpaint / outpaint commands name must exist in the shim's `Flux2Core` module, with
    the statically accessed members (`Flux2Core.version`, `Flux2Profiler.shared.enable()`, `LoRAConfig.load(from:)`, enum cases, ...) —
    except members the callers add themselves in extensions (`TransformerQuantization.parseCLI`: Flux2CLI/QuantizationParsing.swift)."""
    fx = _fixture()["flux2core_types"][type_name]
    src = _flux2core_shim_src()
    assert re.search(r"public\s+(?:final\s+)?(?:class|struct|enum|protocol|typealias|actor)\s+" + type_name + r"\b", src), f"{type_name} ({fx['declared_in']}) is not declared"
    caller_side = set()
    if os.path.isdir("/root/reference/Sources"):
        for f in _fixture()["files"]:
            text = strip(open(os.path.join("/root/reference/Sources", f)).read())
            for m in re.finditer(r"extension\s+" + type_name + r"\b[^{]*\{", text):
                depth, i = 1, m.end()
                while depth:
                    depth += {"{": 1, "}": -1}.get(text[i], 0)
                    i += 1
                caller_side |= set(re.findall(r"func\s+(\w+)", text[m.end():i]))
    else:
        caller_side = {"parseCLI"}
    m = re.search(r"public\s+(?:final\s+)?(?:class|struct|enum)\s+" + type_name + r"\b[^{]*\{", src)
    body = ""
    if m:
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        body = src[m.end():i]
        for mm in re.finditer(r"extension\s+" + type_name + r"\b[^{]*\{", src):
            depth, i = 1, mm.end()
            while depth:
                depth += {"{": 1, "}": -1}.get(src[i], 0)
                i += 1
            body += src[mm.end():i]
    for member in fx["members"]:
        if member in caller_side:
            continue
        assert re.search(r"(?:public\s+(?:static\s+)?(?:func|var|let)\s+|case\s+(?:\w+\s*(?:\([^)]*\))?\s*,\s*)*)" + member + r"\b", body), \
            f"{type_name}.{member} (used in {fx['used_in']}) has no public declaration in the shim"


def test_swift_sources_are_well_formed_as_far_as_text_can_tell():
    """No Swift toolchain here: at least every shim source has balanced (), [], {} outside comments and strings, every `import` names a
    target of the package, the system library target, or an Apple / Foundation framework, and every target directory of Package.swift exists."""
    pkg = open(os.path.join(PKG, "Package.swift")).read()
    targets = set(re.findall(r'\.(?:target|systemLibrary)\(\s*name:\s*"(\w+)"', pkg))
    assert {"CFlux2MI", "MLX", "MLXRandom", "Flux2Core"} <= targets
    allowed = targets | {"Foundation", "CoreGraphics", "ImageIO", "PackageDescription"}
    for t in targets:
        assert os.path.isdir(os.path.join(PKG, "Sources", t)), t
    for mod in sorted(os.listdir(os.path.join(PKG, "Sources"))):
        for f in sorted(os.listdir(os.path.join(PKG, "Sources", mod))):
            if not f.endswith(".swift"):
                continue
            text = strip(open(os.path.join(PKG, "Sources", mod, f)).read())
            stack = []
            for i, c in enumerate(text):
                if c in "([{":
                    stack.append(c)
                elif c in ")]}":
                    assert stack and stack[-1] == {")": "(", "]": "[", "}": "{"}[c], f"{mod}/{f}: unbalanced '{c}' near line {text.count(chr(10), 0, i) + 1}"
                    stack.pop()
            assert not stack, f"{mod}/{f}: {len(stack)} unclosed bracket(s)"
            for m in re.finditer(r"^\s*(?:@\w+\s+)*import\s+(\w+)", text, flags=re.M):
                assert m.group(1) in allowed, f"{mod}/{f}: import {m.group(1)} is neither a target of the package nor a system framework"

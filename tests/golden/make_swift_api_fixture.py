#!/usr/bin/env python3
"""Writes tests/golden/flux2pipeline_public_api.json: the PUBLIC INTERFACE (names, argument labels, types, defaults, async / throws,
return types — no bodies, no source text) of the reference's `Flux2Pipeline`, parsed from
  /root/reference/Sources/Flux2Core/Pipeline/Flux2Pipeline.swift  and  Flux2Pipeline+ChainHelpers.swift  (v2.4.0)
by tests/swift_api.py.  It is the contract the Swift shim under integration/swift/ must match; tests/test_swift_shim_signatures.py
compares both, and re-derives this file from the reference where the reference is present (this container only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.swift_api import key, public_api  # noqa: E402

REF = "/root/reference/Sources/Flux2Core/Pipeline"
FILES = ["Flux2Pipeline.swift", "Flux2Pipeline+ChainHelpers.swift"]
TYPES = ["Flux2Pipeline", "Flux2GenerationResult", "Flux2StepContext"]


def build():
    api = {}
    for t in TYPES:
        decls = []
        for f in FILES:
            decls += public_api(open(os.path.join(REF, f)).read(), t)
        api[t] = sorted(key(d) for d in decls)
    return api


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "flux2pipeline_public_api.json")
    with open(out, "w") as f:
        json.dump({"source": "VincentGourbin/flux-2-swift-mlx v2.4.0, Sources/Flux2Core/Pipeline/{Flux2Pipeline,Flux2Pipeline+ChainHelpers}.swift",
                   "api": build()}, f, indent=1)
    print(out)

"""A small parser for the PUBLIC declarations of a Swift type: `public [static] func`, `public init`, `public [private(set)] var|let`
inside `class|struct|extension <TypeName>` bodies.  Used to compare the shim's `Flux2Pipeline` (integration/swift/) with the
reference's (tests/test_swift_shim_signatures.py): names, argument labels, order, types, default values, async / throws, return type."""
import re


def _strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)) if "\n" not in m.group(0) else re.sub(r"[^\n]", " ", m.group(0)), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _match(src: str, i: int, open_ch: str, close_ch: str) -> int:
    """index of the bracket closing the one at src[i]"""
    depth = 0
    in_str = False
    k = i
    while k < len(src):
        c = src[k]
        if in_str:
            if c == "\\":
                k += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return k
        k += 1
    raise ValueError("unbalanced")


def _split_top(s: str, sep: str = ","):
    out, depth, cur = [], 0, []
    for c in s:
        if c in "([<":
            depth += 1
        elif c in ")]>" and not (c == ">" and cur and cur[-1] == "-"):
            depth -= 1
        if c == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
    if "".join(cur).strip():
        out.append("".join(cur))
    return out


def _norm(s: str) -> str:
    return re.sub(r"\s+", " ", s).strip()


def _param(p: str) -> dict:
    names, rest = p.split(":", 1)
    names = names.split()
    label = names[0]
    default = None
    parts = _split_top(rest, "=")
    typ = _norm(parts[0])
    if len(parts) > 1:
        default = _norm("=".join(parts[1:]))
    return {"label": label, "type": typ, "default": default}


def type_bodies(src: str, type_name: str):
    """(start, end) of every `class|struct|extension type_name {...}` body"""
    src_nc = _strip_comments(src)
    for m in re.finditer(r"\b(?:final\s+)?(?:class|struct|extension)\s+" + re.escape(type_name) + r"\b[^{\n]*\{", src_nc):
        i = m.end() - 1
        yield src_nc, i, _match(src_nc, i, "{", "}")


def public_api(src: str, type_name: str) -> list:
    """Sorted list of declaration records of the type's public surface (nested types' members excluded by brace depth)."""
    out = []
    for text, b0, b1 in type_bodies(src, type_name):
        body = text[b0 + 1:b1]
        # blank out nested braces' contents so that only depth-1 declarations are seen (function bodies, nested types)
        depth, flat = 0, []
        for c in body:
            if c == "{":
                depth += 1
                flat.append(c if depth == 1 else " ")
            elif c == "}":
                flat.append(c if depth == 1 else " ")
                depth -= 1
            else:
                flat.append(c if depth == 0 or c == "\n" else " ")
        flat = "".join(flat)
        for m in re.finditer(r"(?:@discardableResult\s+)?public\s+(static\s+)?(?:func\s+(\w+)|(init))\s*(?:<[^>]*>)?\s*\(", flat):
            i = m.end() - 1
            j = _match(flat, i, "(", ")")
            params = [_param(p) for p in _split_top(flat[i + 1:j]) if p.strip()]
            tail = flat[j + 1:flat.index("{", j)]
            out.append({"kind": "init" if m.group(3) else "func", "name": m.group(2) or "init", "static": bool(m.group(1)),
                        "params": params, "effects_and_return": _norm(tail),
                        "discardable": flat[m.start():m.end()].lstrip().startswith("@discardableResult")})
        for m in re.finditer(r"public\s+(private\(set\)\s+)?(static\s+)?(var|let)\s+(\w+)\s*(?::\s*([^={\n]+?))?\s*(?:=\s*([^{\n]+?))?\s*(\{|\n)", flat):
            typ = _norm(m.group(5)) if m.group(5) else None
            init = _norm(m.group(6)) if m.group(6) else None
            if typ is None and init in ("false", "true"):
                typ = "Bool"
            out.append({"kind": "property", "name": m.group(4), "static": bool(m.group(2)), "type": typ,
                        "settable": m.group(3) == "var" and not m.group(1) and m.group(7) != "{"})
    return sorted(out, key=lambda d: (d["kind"], d["name"], [p["label"] for p in d.get("params", [])]))


def key(d: dict) -> str:
    if d["kind"] == "property":
        return f"{'static ' if d['static'] else ''}var {d['name']}: {d['type']}{' {get set}' if d['settable'] else ' {get}'}"
    ps = ", ".join(f"{p['label']}: {p['type']}" + (f" = {p['default']}" if p["default"] is not None else "") for p in d["params"])
    return f"{'@discardableResult ' if d.get('discardable') else ''}{'static ' if d['static'] else ''}{d['name']}({ps}) {d['effects_and_return']}".strip()

"""What MLX surface does Swift code OUTSIDE Flux2Core use?  A small, purpose-built reader of Swift source (no Swift toolchain on the
build image) that lists, for a set of files:

  import      every `import X` / `@preconcurrency import X` of a module the shim has to provide (MLX*, Flux2Core)
  qualified   module- or namespace-qualified names: `MLX.Memory.activeMemory`, `MLXRandom.seed(_:)`, `Memory.clearCache()`
  func        free functions of MLX: `eval(...)`
  init        `MLXArray(...)` initialiser forms: labels + argument types
  member      members read on a value typed MLXArray: `.shape`
  operator    every binary operator with an MLXArray operand: (lhs type, op, rhs type), literals kept apart (`1 - a` is IntLiteral - MLXArray)

Types are tracked just far enough for the chains' code: names bound by `let x = MLXRandom.normal(..)` / `MLXArray(..)` / a pipeline
method whose declared return type is MLXArray / another MLXArray name or expression, and the parameters of closures annotated with a
typealias whose parameter types are known (`Flux2StepHook = (Flux2StepContext, MLXArray) -> MLXArray`).

Used by tests/golden/make_mlx_surface_fixture.py (reference side -> tests/golden/mlx_surface_used.json) and by
tests/test_swift_mlx_surface.py (the fixture against the declarations of integration/swift/Sources/{MLX,MLXRandom})."""
import re

SHIM_MODULES = ("MLX", "MLXRandom", "MLXNN", "MLXFast", "MLXOptimizers", "MLXLinalg", "MLXFFT", "Flux2Core")


def strip(src: str) -> str:
    """comments and string-literal CONTENTS blanked (positions and newlines kept)"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append(re.sub(r"[^\n]", " ", src[i:j]))
            i = j
        elif src.startswith('"""', i):
            j = src.find('"""', i + 3)
            j = n if j < 0 else j + 3
            out.append('"' + re.sub(r"[^\n]", " ", src[i + 1:j - 1]) + '"')
            i = j
        elif c == '"':
            j, depth = i + 1, 0
            while j < n:
                if src[j] == "\\" and j + 1 < n:
                    if src[j + 1] == "(":
                        depth += 1
                    j += 2
                    continue
                if depth and src[j] == "(":
                    depth += 1
                elif depth and src[j] == ")":
                    depth -= 1
                elif src[j] == '"' and depth == 0:
                    break
                j += 1
            out.append('"' + " " * (j - i - 1) + '"')
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def imports(src: str):
    return sorted({m.group(1) for m in re.finditer(r"^\s*(?:@\w+\s+)*import\s+(\w+)", strip(src), flags=re.M)})


# ------------------------------------------------------------------ expressions
_TOK = re.compile(r"\s*(?:(\d[\d_]*\.\d[\d_]*(?:[eE][+-]?\d+)?)|(\d[\d_]*)|([A-Za-z_]\w*)|(==|!=|<=|>=|&&|\|\||\?\?|->|[-+*/%<>=!?:.,()\[\]{}&|]))")


def _tokens(s: str):
    out, i = [], 0
    while i < len(s):
        m = _TOK.match(s, i)
        if not m:
            if s[i:].strip() == "":
                break
            i += 1
            continue
        i = m.end()
        if m.group(1):
            out.append(("float", m.group(1)))
        elif m.group(2):
            out.append(("int", m.group(2)))
        elif m.group(3):
            out.append(("id", m.group(3)))
        else:
            out.append(("op", m.group(4)))
    return out


class _Expr:
    """Pratt parser over the arithmetic subset; returns a type name and appends what it saw to `uses`."""
    PREC = {"+": 10, "-": 10, "*": 20, "/": 20, "%": 20}

    def __init__(self, toks, env, uses, where):
        self.t, self.i, self.env, self.uses, self.where = toks, 0, env, uses, where

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else ("end", "")

    def eat(self):
        tok = self.peek()
        self.i += 1
        return tok

    def parse(self, min_prec=0):
        lhs = self.unary()
        while True:
            k, v = self.peek()
            if k != "op" or v not in self.PREC or self.PREC[v] < min_prec:
                return lhs
            self.eat()
            rhs = self.parse(self.PREC[v] + 1)
            if "MLXArray" in (lhs, rhs):
                self.uses.append({"kind": "operator", "lhs": lhs, "op": v, "rhs": rhs, "where": self.where})
                lhs = "MLXArray"
            elif lhs.startswith("prop:") or rhs.startswith("prop:"):      # arithmetic on a façade property: fixes its type class
                prop, other = (lhs, rhs) if lhs.startswith("prop:") else (rhs, lhs)
                self.uses.append({"kind": "property_arith", "name": prop[5:], "op": v, "other": other, "where": self.where})
                lhs = "?"
            elif lhs == rhs:
                pass
            elif "FloatLiteral" in (lhs, rhs) or "IntLiteral" in (lhs, rhs):
                lhs = rhs if lhs.endswith("Literal") else lhs
            else:
                lhs = "?"

    def unary(self):
        k, v = self.peek()
        if k == "op" and v == "-":
            self.eat()
            return self.unary()
        return self.postfix(self.primary())

    def args(self):
        """after '(' : list of (label, type); consumes the closing ')'"""
        out = []
        while self.peek() != ("op", ")") and self.peek()[0] != "end":
            label = None
            if self.peek()[0] == "id" and self.i + 1 < len(self.t) and self.t[self.i + 1] == ("op", ":"):
                label = self.eat()[1]
                self.eat()
            out.append((label, self.parse()))
            if self.peek() == ("op", ","):
                self.eat()
        self.eat()
        return out

    def primary(self):
        k, v = self.eat()
        if k == "int":
            return "IntLiteral"
        if k == "float":
            return "FloatLiteral"
        if k == "op" and v == "(":
            t = self.parse()
            if self.peek() == ("op", ")"):
                self.eat()
            return t
        if k == "op" and v == "[":   # array literal: skip
            depth = 1
            while depth and self.peek()[0] != "end":
                tok = self.eat()
                depth += tok == ("op", "[")
                depth -= tok == ("op", "]")
            return "[?]"
        if k == "id":
            if v in ("try", "await"):
                if self.peek() in (("op", "?"), ("op", "!")):
                    self.eat()
                return self.unary()
            if v == "MLXArray" and self.peek() == ("op", "("):
                self.eat()
                a = self.args()
                self.uses.append({"kind": "init", "type": "MLXArray", "labels": [l or "_" for l, _ in a], "arg_types": [t for _, t in a],
                                  "where": self.where})
                return "MLXArray"
            return self.env.get(v, "name:" + v)
        return "?"

    def postfix(self, t):
        while True:
            k, v = self.peek()
            if k == "op" and v == "." and self.i + 1 < len(self.t) and self.t[self.i + 1][0] == "id":
                self.eat()
                name = self.eat()[1]
                call = self.peek() == ("op", "(")
                a = None
                if call:
                    self.eat()
                    a = self.args()
                t = self.member(t, name, a)
            elif k == "op" and v == "(" and t.startswith("name:"):
                self.eat()
                a = self.args()
                t = self.call(t[5:], a)
            elif k == "op" and v in ("!", "?") and not (self.i + 1 < len(self.t) and self.t[self.i + 1][0] in ("int", "float", "id")):
                self.eat()
            else:
                return t

    def member(self, t, name, args):
        if t == "MLXArray":
            self.uses.append({"kind": "member", "of": "MLXArray", "name": name, "call": args is not None, "where": self.where})
            return {"shape": "[Int]"}.get(name, "?")
        if t.startswith("name:"):
            q = t[5:] + "." + name
            if q.split(".")[0] in ("MLX", "MLXRandom", "Memory"):
                if args is None and q == "MLX.Memory":
                    return "name:" + q
                self.uses.append({"kind": "qualified", "name": q, "call": args is not None,
                                  "labels": [l or "_" for l, _ in args] if args is not None else None, "where": self.where})
                return self.env.get("ret:" + q, "prop:" + q if args is None else "?")
            if args is not None and "ret:." + name in self.env:       # a method whose return type the caller told us
                return self.env["ret:." + name]
            return self.env.get("member:" + t[5:] + "." + name, "name:" + q)
        return self.env.get("member:" + t + "." + name, "?")

    def call(self, name, args):
        if name == "eval":
            self.uses.append({"kind": "func", "name": "eval", "arg_types": [t for _, t in args], "where": self.where})
            return "Void"
        return self.env.get("ret:" + name, "?")


def _statements(text: str):
    """(first line number, statement text) with continuation lines joined while brackets are open or the line ends in an operator"""
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        start, cur = i, lines[i]
        while (cur.count("(") > cur.count(")") or cur.count("[") > cur.count("]") or re.search(r"[-+*/=,]\s*$", cur)) and i + 1 < len(lines):
            i += 1
            cur += " " + lines[i].strip()
        yield start + 1, cur
        i += 1


def used_surface(path: str, src: str, method_returns: dict, closure_aliases: dict, member_types: dict, fn_param_types=None):
    """method_returns: {method name: return type} of the pipeline API; closure_aliases: {typealias: [param types]};
    member_types: {"Flux2StepContext.sigmaNext": "Float", ...}."""
    text = strip(src)
    rel = path
    uses = [{"kind": "import", "module": m, "where": rel} for m in imports(src) if m in SHIM_MODULES]
    env = {"ret:MLXRandom.normal": "MLXArray", "ret:MLXRandom.uniform": "MLXArray"}
    for k, v in method_returns.items():
        env["ret:." + k] = v
    for k, v in member_types.items():
        env["member:" + k] = v
    for ln, st in _statements(text):
        where = f"{rel}:{ln}"
        # closure heads: `let f: Alias = { a, b in`
        m = re.search(r":\s*(\w+)\s*=\s*\{\s*([\w\s,]+?)\s+in\b", st)
        if m and m.group(1) in closure_aliases:
            for name, typ in zip([x.strip() for x in m.group(2).split(",")], closure_aliases[m.group(1)]):
                env[name] = typ
            st = st[m.end():]
        m = re.match(r"\s*(?:let|var)\s+(\w+)\s*(?::\s*([\w\[\]?<>.]+))?\s*=\s*(.+)$", st)
        if m:
            t = _Expr(_tokens(m.group(3)), env, uses, where).parse()
            declared = m.group(2)
            if declared:
                env[m.group(1)] = declared.rstrip("?")
            elif not t.startswith("name:") and t != "?":
                env[m.group(1)] = t
            continue
        m = re.match(r"\s*(?:if|guard)\s+let\s+(\w+)\s*=\s*(\w+)\b", st)
        if m and m.group(2) in env:
            env[m.group(1)] = env[m.group(2)].rstrip("?")
        m = re.match(r"\s*return\s+(.+)$", st)
        body = m.group(1) if m else st
        if re.search(r"\b(MLX|MLXRandom|MLXArray|Memory|eval)\b", body) or any(
                re.search(r"\b" + re.escape(k) + r"\b", body) for k, v in env.items() if v == "MLXArray" and ":" not in k):
            # expression statements: parse every maximal expression start (skip leading keywords / assignments)
            body = re.sub(r"^\s*(?:\w+\s*=\s*)", "", body) if re.match(r"^\s*\w+\s*=[^=]", body) else body
            for piece in re.split(r"[{};]", body):
                if piece.strip():
                    _Expr(_tokens(piece), env, uses, where).parse()
    return uses


def summarize(uses):
    """order-free, location-free view: what the façade must declare"""
    out = set()
    for u in uses:
        k = u["kind"]
        if k == "import":
            out.add(("import", u["module"]))
        elif k == "qualified":
            out.add(("qualified", u["name"], "call(" + ",".join(u["labels"]) + ")" if u["call"] else "property"))
        elif k == "func":
            out.add(("func", u["name"], ",".join(u["arg_types"])))
        elif k == "init":
            out.add(("init", u["type"], ",".join(f"{l}:{t}" for l, t in zip(u["labels"], u["arg_types"]))))
        elif k == "member":
            out.add(("member", u["of"], u["name"], "call" if u["call"] else "property"))
        elif k == "operator":
            out.add(("operator", u["lhs"], u["op"], u["rhs"]))
        elif k == "property_arith":
            out.add(("property_arith", u["name"], u["op"], u["other"]))
    return sorted(out)

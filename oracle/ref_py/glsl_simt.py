"""glsl_simt.py — TEST INFRASTRUCTURE (oracle side); never imported by the product.

A second, independent way of EXECUTING the reference's shader text (STEP_FS_SRC, RENDER_FS_SRC of
pages/airfoil_flow_lbm_aerolab.html:222-422) in the build container, beside oracle/ref_js/glsl2js.js:

    glsl2js.js   (round 1)  a TRANSPILER to JavaScript source, one site at a time, doubles rounded with Math.fround;
    this module  (round 4)  an AST INTERPRETER in Python that runs ALL lattice sites at once, SIMT-style — every value is a
                            NumPy array over the sites, divergent `if` / `return` are execution masks — in NumPy's own
                            binary32 (or binary64) arithmetic.

The two share no code, no language and no arithmetic engine; tests/test_glsl_second_opinion.py feeds both the same shader text
(read from /root/reference when the test runs in the build container; skipped elsewhere) and requires this one to reproduce the
committed goldens — which glsl2js.js produced — bit for bit.  What a reviewer gets from it: the goldens do not rest on one
hand-written GLSL semantics ("the one link a second pair of eyes cannot get from the reference alone").

Arithmetic model (the same contract the CPU oracle states): IEEE binary32 / binary64, one rounding per operation, literal evaluation
order, no contraction; NEAREST sampling with CLAMP_TO_EDGE (texel index = floor(uv * size) clamped), an R8 texel reads value / 255.
GLSL built-ins by their specification formulas: clamp = min(max(x, lo), hi), mix(x, y, a) = x * (1 - a) + y * a,
length(v) = sqrt(v.x * v.x + v.y * v.y).

Supported subset: global uniform / in / out / const declarations, layout(...) qualifiers, functions, float / int / bool / vec2 / vec3 /
vec4 / ivec2 locals, fixed-size float arrays, if / else, for, return, + - * / < > <= >= == != && || ! = += -= *= /= ++, single-
component swizzles, constructors, texture clamp sqrt max min floor mix length.  Anything else raises.
"""
from __future__ import annotations

import re

import numpy as np

TYPES = {"void", "float", "int", "bool", "vec2", "vec3", "vec4", "ivec2", "sampler2D"}
VEC_N = {"vec2": 2, "vec3": 3, "vec4": 4, "ivec2": 2}
SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}

_TOKEN = re.compile(r"\s+|//[^\n]*|/\*.*?\*/|#[^\n]*|(?P<f>\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)|(?P<i>\d+)|(?P<id>[A-Za-z_]\w*)"
                    r"|(?P<op>\+\+|--|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||[-+*/=<>!(){}\[\];,.?:])", re.S)


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"glsl_simt: cannot tokenize at {pos}: {src[pos:pos + 30]!r}")
        pos = m.end()
        for kind in ("f", "i", "id", "op"):
            if m.group(kind) is not None:
                out.append((kind, m.group(kind)))
    out.append(("eof", ""))
    return out


# ------------------------------------------------------------------------------------------------ parser
class Parser:
    def __init__(self, src):
        self.t, self.p = tokenize(src), 0

    def peek(self, o=0):
        return self.t[self.p + o]

    def next(self):
        tok = self.t[self.p]
        self.p += 1
        return tok

    def accept(self, v):
        if self.peek()[1] == v and self.peek()[0] in ("op", "id"):
            self.p += 1
            return True
        return False

    def expect(self, v):
        if not self.accept(v):
            raise SyntaxError(f"glsl_simt: expected {v!r}, got {self.peek()!r}")

    def unit(self):
        decls, funcs = [], {}
        while self.peek()[0] != "eof":
            if self.accept("precision"):
                while not self.accept(";"):
                    self.next()
                continue
            if self.accept("layout"):
                self.expect("(")
                while not self.accept(")"):
                    self.next()
            qual = None
            while self.peek()[1] in ("uniform", "in", "out", "const"):
                qual = self.next()[1]
            ty = self.next()[1]
            if ty == "highp":
                ty = self.next()[1]
            if ty not in TYPES:
                raise SyntaxError(f"glsl_simt: type expected, got {ty!r}")
            name = self.next()[1]
            if self.accept("("):                                   # function
                params = []
                while not self.accept(")"):
                    pt, pn = self.next()[1], self.next()[1]
                    params.append((pt, pn))
                    self.accept(",")
                funcs[name] = (ty, params, self.block())
                continue
            while True:                                            # global declarator list
                init = self.expr() if self.accept("=") else None
                decls.append((qual, ty, name, init))
                if self.accept(","):
                    name = self.next()[1]
                    continue
                self.expect(";")
                break
        return decls, funcs

    def block(self):
        self.expect("{")
        body = []
        while not self.accept("}"):
            body.append(self.stmt())
        return ("block", body)

    def stmt(self):
        k, v = self.peek()
        if v == "{" and k == "op":
            return self.block()
        if v == "if" and k == "id":
            self.next(); self.expect("(")
            c = self.expr(); self.expect(")")
            a = self.stmt()
            b = self.stmt() if self.accept("else") else None
            return ("if", c, a, b)
        if v == "for" and k == "id":
            self.next(); self.expect("(")
            init = self.simple(); self.expect(";")
            cond = self.expr(); self.expect(";")
            step = self.simple(); self.expect(")")
            return ("for", init, cond, step, self.stmt())
        if v == "return" and k == "id":
            self.next()
            e = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", e)
        s = self.simple()
        self.expect(";")
        return s

    def simple(self):
        """declaration (possibly const, with a declarator list) or expression statement, without the semicolon"""
        if self.peek()[1] == "const":
            self.next()
        if self.peek()[0] == "id" and self.peek()[1] in TYPES and self.peek(1)[0] == "id":
            ty = self.next()[1]
            items = []
            while True:
                name = self.next()[1]
                size = None
                if self.accept("["):
                    size = int(self.next()[1]); self.expect("]")
                init = self.expr() if self.accept("=") else None
                items.append((name, size, init))
                if not self.accept(","):
                    break
            return ("decl", ty, items)
        return ("expr", self.expr())

    # precedence climbing: assignment < || < && < == != < relational < + - < * / < unary < postfix
    def expr(self):
        lhs = self.lor()
        if self.peek()[0] == "op" and self.peek()[1] in ("=", "+=", "-=", "*=", "/="):
            op = self.next()[1]
            return ("assign", op, lhs, self.expr())
        return lhs

    def _bin(self, sub, ops):
        e = sub()
        while self.peek()[0] == "op" and self.peek()[1] in ops:
            op = self.next()[1]
            e = ("bin", op, e, sub())
        return e

    def lor(self): return self._bin(self.land, ("||",))
    def land(self): return self._bin(self.eq, ("&&",))
    def eq(self): return self._bin(self.rel, ("==", "!="))
    def rel(self): return self._bin(self.add, ("<", ">", "<=", ">="))
    def add(self): return self._bin(self.mul, ("+", "-"))
    def mul(self): return self._bin(self.unary, ("*", "/"))

    def unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "+", "!"):
            op = self.next()[1]
            return ("un", op, self.unary())
        return self.postfix()

    def postfix(self):
        k, v = self.next()
        if k == "f":
            e = ("float", v)
        elif k == "i":
            e = ("int", int(v))
        elif k == "op" and v == "(":
            e = self.expr(); self.expect(")")
        elif k == "id":
            if self.accept("("):
                args = []
                while not self.accept(")"):
                    args.append(self.expr())
                    self.accept(",")
                e = ("call", v, args)
            elif v in ("true", "false"):
                e = ("bool", v == "true")
            else:
                e = ("var", v)
        else:
            raise SyntaxError(f"glsl_simt: unexpected token {(k, v)!r}")
        while True:
            if self.accept("."):
                e = ("swz", e, self.next()[1])
            elif self.accept("["):
                i = self.expr(); self.expect("]")
                e = ("idx", e, i)
            elif self.peek() == ("op", "++"):
                self.next()
                e = ("assign", "+=", e, ("int", 1))
            else:
                return e


# ------------------------------------------------------------------------------------------------ values
class Vec:
    """A GLSL vector: a list of per-site component arrays (or uniform scalars)."""
    def __init__(self, comps, integer=False):
        self.c, self.integer = list(comps), integer


class Sampler:
    def __init__(self, data, scale=None):
        self.data = data                       # [H][W][C] (C = 4) or [H][W] for an R8 texture
        self.scale = scale


class _Return(Exception):
    pass


class Frame:
    def __init__(self, active):
        self.scopes = [{}]
        self.dm = {}                           # (id(scope), name) -> the execution mask a variable was declared under
        self.active = active                   # per-site execution mask of this function activation (bool array or True)
        self.ret_mask = None                   # sites that have returned
        self.ret_val = None


class Shader:
    """One shader compiled from its text; run(**inputs) executes main() for all sites at once and returns the `out` variables."""

    def __init__(self, src, dtype=np.float32):
        self.ft = np.dtype(dtype).type
        self.decls, self.funcs = Parser(src).unit()
        self.globals = {}
        self.outs = [n for q, _, n, _ in self.decls if q == "out"]

    # ---- helpers
    def _f(self, x):
        return self.ft(x) if np.isscalar(x) else np.asarray(x, dtype=self.ft)

    def _where(self, m, a, b):
        if m is True:
            return a
        return np.where(m, a, b)

    def _and(self, a, b):
        if a is True:
            return b
        if b is True:
            return a
        return a & b

    def _live(self, fr):
        """sites of this activation that are executing: active and not yet returned"""
        if fr.ret_mask is None:
            return fr.active
        return self._and(fr.active, ~fr.ret_mask)

    # ---- execution
    def run(self, nsites, **inputs):
        self.n = nsites
        self.globals = {}
        np.seterr(all="ignore")                # masked-out sites compute on whatever they hold
        fr = Frame(True)
        for qual, ty, name, init in self.decls:
            if qual in ("uniform", "in"):
                self.globals[name] = inputs[name]
            elif qual == "out":
                self.globals[name] = Vec([np.zeros(nsites, self.ft) for _ in range(VEC_N[ty])])
            else:
                self.globals[name] = self.eval(init, fr, True)
        self.call("main", [], True)
        return {n: self.globals[n] for n in self.outs}

    def call(self, name, args, active):
        rty, params, body = self.funcs[name]
        fr = Frame(active)
        for (pt, pn), a in zip(params, args):
            fr.scopes[0][pn] = a
            fr.dm[(id(fr.scopes[0]), pn)] = active
        self.exec(body, fr, active)
        return fr.ret_val

    def lookup(self, fr, name):
        for s in reversed(fr.scopes):
            if name in s:
                return s, name
        if name in self.globals:
            return self.globals, name
        raise NameError(f"glsl_simt: unknown identifier {name}")

    def exec(self, st, fr, mask):
        """mask: sites for which this statement is reached (before removing the returned ones)"""
        live = self._and(mask, True if fr.ret_mask is None else ~fr.ret_mask)
        if live is not True and not np.any(live):
            return
        kind = st[0]
        if kind == "block":
            fr.scopes.append({})
            for s in st[1]:
                self.exec(s, fr, mask)
            fr.scopes.pop()
        elif kind == "decl":
            _, ty, items = st
            for name, size, init in items:
                fr.dm[(id(fr.scopes[-1]), name)] = live
                if size is not None:
                    fr.scopes[-1][name] = [np.zeros(self.n, self.ft) for _ in range(size)]
                elif init is None:
                    fr.scopes[-1][name] = (Vec([np.zeros(self.n, self.ft) for _ in range(VEC_N[ty])], ty == "ivec2") if ty in VEC_N
                                           else (0 if ty == "int" else (False if ty == "bool" else self.ft(0))))
                else:
                    fr.scopes[-1][name] = self.convert(self.eval(init, fr, live), ty)
        elif kind == "expr":
            self.eval(st[1], fr, live)
        elif kind == "if":
            c = self.eval(st[1], fr, live)
            if isinstance(c, (bool, np.bool_)):                    # uniform condition: ordinary control flow
                if c:
                    self.exec(st[2], fr, mask)
                elif st[3] is not None:
                    self.exec(st[3], fr, mask)
            else:
                self.exec(st[2], fr, self._and(mask, c))
                if st[3] is not None:
                    self.exec(st[3], fr, self._and(mask, ~c))
        elif kind == "for":
            fr.scopes.append({})
            self.exec(st[1], fr, mask)
            guard = 0
            while True:
                c = self.eval(st[2], fr, live)
                if not isinstance(c, (bool, np.bool_)):
                    raise NotImplementedError("glsl_simt: loop conditions must be uniform")
                if not c:
                    break
                self.exec(st[4], fr, mask)
                self.exec(st[3], fr, mask)
                guard += 1
                if guard > 4096:
                    raise RuntimeError("glsl_simt: runaway loop")
            fr.scopes.pop()
        elif kind == "return":
            val = None if st[1] is None else self.eval(st[1], fr, live)
            if val is not None:
                fr.ret_val = val if fr.ret_val is None else self.select(live, val, fr.ret_val)
            if live is True:
                fr.ret_mask = np.ones(self.n, bool)
            else:
                fr.ret_mask = live.copy() if fr.ret_mask is None else (fr.ret_mask | live)
        else:
            raise NotImplementedError(kind)

    def select(self, m, a, b):
        if m is True:
            return a
        if isinstance(a, Vec):
            return Vec([self._where(m, x, y) for x, y in zip(a.c, b.c)], a.integer)
        return self._where(m, a, b)

    def convert(self, v, ty):
        if ty == "float":
            return self._f(v)
        return v

    # ---- expressions
    def eval(self, e, fr, live):
        k = e[0]
        if k == "float":
            return self.ft(float(e[1]))
        if k == "int":
            return int(e[1])
        if k == "bool":
            return e[1]
        if k == "var":
            s, n = self.lookup(fr, e[1])
            return s[n]
        if k == "swz":
            v = self.eval(e[1], fr, live)
            if len(e[2]) != 1:
                raise NotImplementedError("glsl_simt: multi-component swizzle")
            return v.c[SWZ[e[2]]]
        if k == "idx":
            a = self.eval(e[1], fr, live)
            i = self.eval(e[2], fr, live)
            if not isinstance(i, (int, np.integer)):
                raise NotImplementedError("glsl_simt: array index must be uniform")
            return a[i]
        if k == "un":
            v = self.eval(e[2], fr, live)
            if e[1] == "!":
                return (not v) if isinstance(v, (bool, np.bool_)) else ~v
            if e[1] == "+":
                return v
            return Vec([-c for c in v.c], v.integer) if isinstance(v, Vec) else -v
        if k == "bin":
            return self.binop(e[1], self.eval(e[2], fr, live), self.eval(e[3], fr, live))
        if k == "assign":
            return self.assign(e, fr, live)
        if k == "call":
            return self.builtin_or_call(e[1], [self.eval(a, fr, live) for a in e[2]], live)
        raise NotImplementedError(k)

    def binop(self, op, a, b):
        if isinstance(a, Vec) or isinstance(b, Vec):
            if op not in ("+", "-", "*", "/"):
                raise NotImplementedError("glsl_simt: vector comparison")
            n = len(a.c) if isinstance(a, Vec) else len(b.c)
            ac = a.c if isinstance(a, Vec) else [a] * n
            bc = b.c if isinstance(b, Vec) else [b] * n
            integer = (a.integer if isinstance(a, Vec) else True) and (b.integer if isinstance(b, Vec) else True)
            return Vec([self.binop(op, x, y) for x, y in zip(ac, bc)], integer)
        if op in ("&&", "||"):
            if isinstance(a, (bool, np.bool_)) and isinstance(b, (bool, np.bool_)):
                return (a and b) if op == "&&" else (a or b)
            return (a & b) if op == "&&" else (a | b)
        both_int = self.is_int(a) and self.is_int(b)
        if not both_int:                                           # GLSL ES has no implicit int -> float: the shaders convert explicitly
            a, b = self._f(a), self._f(b)
        if op == "+": return a + b
        if op == "-": return a - b
        if op == "*": return a * b
        if op == "/":
            if both_int:
                return a // b
            with np.errstate(divide="ignore", invalid="ignore"):
                return a / b
        if op == "<": return a < b
        if op == ">": return a > b
        if op == "<=": return a <= b
        if op == ">=": return a >= b
        if op == "==": return a == b
        if op == "!=": return a != b
        raise NotImplementedError(op)

    @staticmethod
    def is_uniform(v):
        return not isinstance(v, (np.ndarray, Vec, list, Sampler))

    @staticmethod
    def is_int(v):
        if isinstance(v, (bool, np.bool_)):
            return False
        if isinstance(v, (int, np.integer)):
            return True
        return isinstance(v, np.ndarray) and v.dtype.kind == "i"

    def assign(self, e, fr, live):
        _, op, target, rhs = e
        val = self.eval(rhs, fr, live)
        if op != "=":
            val = self.binop(op[0], self.eval(target, fr, live), val)
        # resolve the l-value
        if target[0] == "var":
            s, n = self.lookup(fr, target[1])
            old = s[n]
            if live is not True and self.is_uniform(val) and self.is_uniform(old):
                # a uniform value assigned under a divergent mask stays uniform only if exactly the sites that own the variable execute the
                # assignment (a loop counter declared and stepped inside one branch); otherwise it becomes per-site
                dm = fr.dm.get((id(s), n), True)
                if dm is not True and np.array_equal(dm, live):
                    s[n] = val
                    return val
                old = np.full(self.n, old)
            s[n] = self.merge(live, val, old)
            return s[n]
        if target[0] == "idx":
            arr = self.eval(target[1], fr, live)
            i = self.eval(target[2], fr, live)
            arr[i] = self.merge(live, val, arr[i])
            return arr[i]
        if target[0] == "swz":
            v = self.eval(target[1], fr, live)
            c = SWZ[target[2]]
            v.c[c] = self.merge(live, val, v.c[c])
            return v.c[c]
        raise NotImplementedError("glsl_simt: assignment target")

    def merge(self, live, new, old):
        """masked assignment: sites outside `live` keep the old value"""
        if live is True:
            return new
        if isinstance(new, Vec):
            oc = old.c if isinstance(old, Vec) else [old] * len(new.c)
            return Vec([self.merge(live, x, y) for x, y in zip(new.c, oc)], new.integer)
        if isinstance(new, (bool, np.bool_)) and isinstance(old, (bool, np.bool_)) and new == old:
            return new
        if self.is_int(new) and self.is_int(old):
            return np.where(live, new, old).astype(np.int64)
        if isinstance(new, (bool, np.bool_, np.ndarray)) and getattr(new, "dtype", np.dtype(bool)).kind == "b":
            return np.where(live, new, old)
        return np.where(live, self._f(new), self._f(old)).astype(self.ft)

    def builtin_or_call(self, name, a, live):
        ft = self.ft
        if name in VEC_N:
            n, comps = VEC_N[name], []
            for x in a:
                comps += x.c if isinstance(x, Vec) else [x]
            if len(comps) == 1:
                comps = comps * n
            if len(comps) != n:
                raise TypeError(f"glsl_simt: {name} from {len(comps)} components")
            if name == "ivec2":                                     # float -> int truncates towards zero
                return Vec([c if self.is_int(c) else np.trunc(c).astype(np.int64) for c in comps], True)
            return Vec([self._f(c) for c in comps])
        if name == "float":
            return self._f(a[0])
        if name == "int":
            return a[0] if self.is_int(a[0]) else np.trunc(a[0]).astype(np.int64)
        if name == "texture":
            return self.texture(a[0], a[1])
        if name == "clamp":
            return np.minimum(np.maximum(self._f(a[0]), self._f(a[1])), self._f(a[2]))
        if name == "sqrt":
            with np.errstate(invalid="ignore"):
                return np.sqrt(self._f(a[0]))
        if name == "max":
            return np.maximum(self._f(a[0]), self._f(a[1]))
        if name == "min":
            return np.minimum(self._f(a[0]), self._f(a[1]))
        if name == "floor":
            return np.floor(self._f(a[0]))
        if name == "length":
            v = a[0]
            acc = v.c[0] * v.c[0]
            for c in v.c[1:]:
                acc = acc + c * c
            return np.sqrt(acc)
        if name == "mix":
            x, y, t = a
            one = ft(1.0)
            xc = x.c if isinstance(x, Vec) else [x]
            yc = y.c if isinstance(y, Vec) else [y]
            out = [p * (one - t) + q * t for p, q in zip(xc, yc)]
            return Vec(out) if isinstance(x, Vec) else out[0]
        if name in self.funcs:
            return self.call(name, a, live)
        raise NotImplementedError(f"glsl_simt: function {name}")

    def texture(self, s, uv):
        ft = self.ft
        d = s.data
        h, w = d.shape[0], d.shape[1]
        ix = np.floor(self._f(uv.c[0]) * ft(w)).astype(np.int64)
        iy = np.floor(self._f(uv.c[1]) * ft(h)).astype(np.int64)
        ix = np.clip(ix, 0, w - 1)
        iy = np.clip(iy, 0, h - 1)
        if d.ndim == 2:                                             # R8: value / 255 in .r, (0, 0, 1) behind it
            r = self._f(d[iy, ix].astype(np.float64) * (1.0 / 255.0 if s.scale is None else s.scale))
            z = np.zeros_like(r)
            return Vec([r, z, z, z + ft(1.0)])
        t = d[iy, ix]
        return Vec([t[..., k] for k in range(4)])

"""wgsl_exec.py - executes the REFERENCE'S OWN shader text.  TEST INFRASTRUCTURE ONLY; the reference's shaders are read only where /root/reference exists
(the build container); compile_source() runs any WGSL text of the subset.

Every other oracle in this directory is a restatement written by reading ray.wgsl.  This module instead READS
/root/reference/src/renderer/shaders/ray.wgsl (and sky.wgsl, the resolve pass behind it) at run time, parses it (a recursive-descent parser for the WGSL subset the file
uses), translates every function into Python and runs it - one invocation of `main` per pixel, as
RayPipeline::pass dispatches it (src/renderer/pipelines/ray_pipeline.rs:301-309).  Nothing of the shader is restated here: if a
line of the shader changed, the frames this module produces would change with it.  What IS defined here is what WGSL leaves to the
implementation, and it is defined exactly as the LITERAL evaluation of the other oracles defines it, so that
the frames can be compared bit for bit:
    every operator one IEEE binary32 operation in source order (numpy.float32 scalars); AbstractFloat constant expressions in
    binary64, rounded once when they meet an f32 (numpy's weak-scalar rule is exactly WGSL's); dot = (x*x + y*y) + z*z;
    vector / scalar = vector * (1 / scalar); pow(x, 2 | 4 | 5) by multiplication, pow(x, -0.001) / acos / atan2 / sin / cos / tan by
    the portable forms (oracle/wgsl_builtins.py: a transcription of its own, shared with no restatement), pow(x, 1.3) by glibc powf; min / max / clamp by compare-select; textures RGBA8 unorm,
    bilinear, clamp-to-edge (src/renderer/texture.rs:16-69); float % = truncated remainder; out-of-range textureLoad clamped.
Deviation D1 (the Runge-Kutta retry loop of ray.wgsl:425-451 cannot terminate once e_max > 1, because it cannot change h): a `while` loop
whose variables are bit-identical at the top of two successive passes is left there - which yields exactly what one pass computed.

Its frames are committed as tests/golden/wgsl_exec.npz by tests/golden/make_golden_wgsl.py; the C oracle's literal mode, the NumPy
restatement's and the literal HIP kernel are held to them bit for bit (tests/test_wgsl_pin.py, tests/test_gpu_literal.py; the interpreter's
own unit tests: tests/test_wgsl_exec_unit.py): that
pins the restatements to the reference's text - not to a driver's floating-point choices, which nothing in this image can run.
"""
from __future__ import annotations

import ctypes
import re
import struct

import numpy as np

from . import wgsl_builtins as WB       # the portable forms of N4, written separately from the restatements this module pins (see its header)

F = np.float32
_LIBM = ctypes.CDLL("libm.so.6")
_LIBM.powf.restype = ctypes.c_float
_LIBM.powf.argtypes = [ctypes.c_float, ctypes.c_float]
SHADER = "/root/reference/src/renderer/shaders/ray.wgsl"

# ------------------------------------------------------------------------------------------------------------------ tokens
TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*)
  | (?P<num>(?:0[xX][0-9a-fA-F]+[iu]?)|(?:(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fiuh]?))
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>->|\+\+|--|&&|\|\||==|!=|<=|>=|\+=|-=|\*=|/=|%=|<<|>>|[-+*/%<>=!&|^~.,;:(){}\[\]@])
""", re.X)


def tokenize(src):
    out, i = [], 0
    while i < len(src):
        m = TOKEN.match(src, i)
        if not m:
            raise SyntaxError("wgsl: cannot tokenise at %r" % src[i:i + 30])
        i = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group()))
    out.append(("eof", ""))
    return out


# ------------------------------------------------------------------------------------------------------------------ parser
TEMPLATED = {"vec2", "vec3", "vec4", "mat3x3", "array", "texture_2d", "texture_storage_2d", "ptr"}


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]; self.i += 1
        return tok

    def accept(self, v):
        if self.peek()[1] == v and self.peek()[0] != "eof":
            self.i += 1
            return True
        return False

    def expect(self, v):
        tok = self.next()
        if tok[1] != v:
            raise SyntaxError("wgsl: expected %r, got %r (token %d)" % (v, tok[1], self.i))
        return tok

    def ident(self):
        k, v = self.next()
        if k != "id":
            raise SyntaxError("wgsl: identifier expected, got %r" % v)
        return v

    # -- types: f32 | vec3<f32> | array<Node, 19> | Name
    def type_(self):
        name = self.ident()
        if name in TEMPLATED and self.peek()[1] == "<":
            self.next()
            args = []
            while True:
                if self.peek()[0] == "num":
                    args.append(self.next()[1])
                else:
                    args.append(self.type_())
                if self.accept(">"):
                    break
                self.expect(",")
            return (name, tuple(args))
        return (name, ())

    def attributes(self):
        while self.accept("@"):
            self.ident()
            if self.accept("("):
                depth = 1
                while depth:
                    v = self.next()[1]
                    depth += (v == "(") - (v == ")")

    # -- module
    def module(self):
        decls = []
        while self.peek()[0] != "eof":
            self.attributes()
            kw = self.peek()[1]
            if kw == "const":
                self.next(); name = self.ident()
                ty = self.type_() if self.accept(":") else None
                self.expect("="); e = self.expr(); self.expect(";")
                decls.append(("const", name, ty, e))
            elif kw == "var":
                self.next()
                if self.accept("<"):
                    while not self.accept(">"):
                        self.next()
                name = self.ident(); self.expect(":"); ty = self.type_(); self.expect(";")
                decls.append(("gvar", name, ty))
            elif kw == "struct":
                self.next(); name = self.ident(); self.expect("{")
                fields = []
                while not self.accept("}"):
                    self.attributes()
                    fn = self.ident(); self.expect(":"); ft = self.type_()
                    fields.append((fn, ft))
                    self.accept(",")
                self.accept(";")
                decls.append(("struct", name, fields))
            elif kw == "fn":
                self.next(); name = self.ident(); self.expect("(")
                params = []
                while not self.accept(")"):
                    self.attributes()
                    pn = self.ident(); self.expect(":"); pt = self.type_()
                    params.append((pn, pt))
                    self.accept(",")
                ret = None
                if self.accept("->"):
                    self.attributes(); ret = self.type_()
                decls.append(("fn", name, params, ret, self.block()))
            else:
                raise SyntaxError("wgsl: unexpected %r at module scope" % kw)
        return decls

    # -- statements
    def block(self):
        self.expect("{")
        out = []
        while not self.accept("}"):
            out.append(self.statement())
        return out

    def statement(self):
        k, v = self.peek()
        if v == "{":
            return ("block", self.block())
        if v in ("let", "var"):
            self.next(); name = self.ident()
            ty = self.type_() if self.accept(":") else None
            e = self.expr() if self.accept("=") else None
            self.expect(";")
            return (v, name, ty, e)
        if v == "if":
            self.next(); c = self.expr(); then = self.block(); els = None
            if self.accept("else"):
                els = [self.statement()] if self.peek()[1] == "if" else self.block()
            return ("if", c, then, els)
        if v == "while":
            self.next(); c = self.expr()
            return ("while", c, self.block())
        if v == "for":
            self.next(); self.expect("(")
            init = None if self.peek()[1] == ";" else self.simple()
            self.expect(";")
            cond = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            upd = None if self.peek()[1] == ")" else self.simple()
            self.expect(")")
            return ("for", init, cond, upd, self.block())
        if v == "break":
            self.next(); self.expect(";"); return ("break",)
        if v == "continue":
            raise SyntaxError("wgsl: `continue` is not in the subset (the shader does not use it)")
        if v == "return":
            self.next()
            e = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", e)
        s = self.simple(); self.expect(";")
        return s

    def simple(self):                   # let/var (for-init), assignment, ++, call
        if self.peek()[1] in ("let", "var"):
            kw = self.next()[1]; name = self.ident()
            ty = self.type_() if self.accept(":") else None
            e = self.expr() if self.accept("=") else None
            return (kw, name, ty, e)
        lhs = self.unary()
        v = self.peek()[1]
        if v in ("=", "+=", "-=", "*=", "/=", "%="):
            self.next()
            return ("assign", lhs, v, self.expr())
        if v in ("++", "--"):
            self.next()
            return ("assign", lhs, "+=" if v == "++" else "-=", ("num", "1"))
        return ("expr", lhs)

    # -- expressions (WGSL precedence; the shader mixes && and || only with parentheses or in the natural order)
    LEVELS = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]

    def expr(self, lvl=0):
        if lvl == len(self.LEVELS):
            return self.unary()
        e = self.expr(lvl + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[lvl]:
            op = self.next()[1]
            e = ("bin", op, e, self.expr(lvl + 1))
        return e

    def unary(self):
        if self.peek()[1] in ("-", "!") and self.peek()[0] == "op":
            op = self.next()[1]
            return ("un", op, self.unary())
        return self.postfix(self.primary())

    def primary(self):
        k, v = self.next()
        if k == "num":
            return ("num", v)
        if v == "(":
            e = self.expr(); self.expect(")")
            return ("paren", e)
        if v in ("true", "false"):
            return ("bool", v == "true")
        if k == "id":
            targs = ()
            if v in TEMPLATED and self.peek()[1] == "<":
                self.i -= 1
                ty = self.type_()
                v, targs = ty
            if self.peek()[1] == "(":
                self.next()
                args = []
                while not self.accept(")"):
                    args.append(self.expr()); self.accept(",")
                return ("call", v, targs, args)
            return ("id", v)
        raise SyntaxError("wgsl: unexpected token %r in an expression" % v)

    def postfix(self, e):
        while True:
            if self.accept("."):
                e = ("member", e, self.ident())
            elif self.accept("["):
                idx = self.expr(); self.expect("]")
                e = ("index", e, idx)
            else:
                return e


# ------------------------------------------------------------------------------------------------------------------ runtime
SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def _weak(x):                         # an AbstractFloat / AbstractInt that meets a concrete vector: rounded to f32 once
    return F(x) if isinstance(x, (float, int)) and not isinstance(x, bool) else x


class Vec:
    """vecN<f32> (components numpy.float32), vecN<i32> (Python ints) or vecN<bool>; immutable."""
    __slots__ = ("c",)

    def __init__(self, comps):
        self.c = tuple(comps)

    def __getattr__(self, name):
        try:
            idx = [SWZ[ch] for ch in name]
        except KeyError:
            raise AttributeError(name)
        return self.c[idx[0]] if len(idx) == 1 else Vec(self.c[i] for i in idx)

    def _is_int(self):
        return isinstance(self.c[0], int) and not isinstance(self.c[0], bool)

    def _lift(self, o):
        if isinstance(o, Vec):
            return o.c
        if not self._is_int():
            o = _weak(o)
        return (o,) * len(self.c)

    def __add__(self, o): return Vec(a + b for a, b in zip(self.c, self._lift(o)))
    def __radd__(self, o): return Vec(b + a for a, b in zip(self.c, self._lift(o)))
    def __sub__(self, o): return Vec(a - b for a, b in zip(self.c, self._lift(o)))
    def __rsub__(self, o): return Vec(b - a for a, b in zip(self.c, self._lift(o)))
    def __mul__(self, o):
        if isinstance(o, Mat3):
            return NotImplemented
        return Vec(a * b for a, b in zip(self.c, self._lift(o)))
    def __rmul__(self, o): return Vec(b * a for a, b in zip(self.c, self._lift(o)))
    def __neg__(self): return Vec(-a for a in self.c)
    def __lt__(self, o): return Vec(bool(a < b) for a, b in zip(self.c, self._lift(o)))
    def __gt__(self, o): return Vec(bool(a > b) for a, b in zip(self.c, self._lift(o)))
    def __le__(self, o): return Vec(bool(a <= b) for a, b in zip(self.c, self._lift(o)))
    def __ge__(self, o): return Vec(bool(a >= b) for a, b in zip(self.c, self._lift(o)))
    def __eq__(self, o): return Vec(bool(a == b) for a, b in zip(self.c, self._lift(o)))
    def __ne__(self, o): return Vec(bool(a != b) for a, b in zip(self.c, self._lift(o)))
    __hash__ = None


class Mat3:
    """mat3x3<f32>: three column vectors."""
    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = tuple(cols)

    def __mul__(self, v):             # M * v = (c0*v.x + c1*v.y) + c2*v.z
        c0, c1, c2 = self.cols
        return (c0 * v.c[0] + c1 * v.c[1]) + c2 * v.c[2]


def _cp(x):                           # WGSL values are values: structs and arrays are copied on let / var / assignment / call / return
    if isinstance(x, Struct):
        return x.copy()
    if isinstance(x, list):
        return [_cp(e) for e in x]
    return x


def conc(x):                          # concretisation of a let / var initialiser or an assigned value
    if isinstance(x, float):
        return F(x)
    return _cp(x)


def _key(x):                          # the bits of a value (NaN-safe), for the fixed-point check of `while` loops
    if isinstance(x, np.floating):
        return x.tobytes()
    if isinstance(x, Vec):
        return tuple(_key(c) for c in x.c)
    if isinstance(x, Struct):
        return tuple(_key(getattr(x, n)) for n, _ in x.FIELDS)
    if isinstance(x, (list, tuple)):
        return tuple(_key(e) for e in x)
    if isinstance(x, Mat3):
        return tuple(_key(c) for c in x.cols)
    return x


class Struct:
    FIELDS = ()                       # (name, type)

    def __init__(self, *args):
        for (n, t), a in zip(self.FIELDS, args):
            object.__setattr__(self, n, cv(t, a))

    def __setattr__(self, n, v):
        object.__setattr__(self, n, cv(self.TYPES[n], v))

    def copy(self):
        o = object.__new__(type(self))
        for n, _ in self.FIELDS:
            object.__setattr__(o, n, _cp(getattr(self, n)))
        return o


STRUCTS = {}


def zero(ty):
    name, args = ty
    if name == "f32":
        return F(0.0)
    if name in ("i32", "u32"):
        return 0
    if name == "bool":
        return False
    if name in ("vec2", "vec3", "vec4"):
        n = int(name[3])
        return Vec((zero(args[0]),) * n)
    if name == "mat3x3":
        return Mat3([Vec((F(0),) * 3)] * 3)
    if name == "array":
        return [zero(args[0]) for _ in range(int(eval_const_size(args[1])))]
    cls = STRUCTS[name]
    return cls(*[zero(t) for _, t in cls.FIELDS])


CONST_SIZES = {}


def eval_const_size(tok):
    return CONST_SIZES[tok] if tok in CONST_SIZES else int(str(tok).rstrip("iu"))


def cv(ty, v):                        # conversion of a value to a declared type (params, fields, typed let / var, returns)
    name = ty[0]
    if name == "f32":
        return F(v)
    if name in ("i32", "u32"):
        return int(v)
    if name == "bool":
        return bool(v)
    return _cp(v)


def mk_vec(n, elem, *args):
    comps = []
    for a in args:
        if isinstance(a, Vec):
            comps.extend(a.c)
        else:
            comps.append(a)
    if len(comps) == 1:
        comps = comps * n
    if len(comps) != n:
        raise TypeError("vec%d from %d components" % (n, len(comps)))
    if elem == "f32":
        return Vec(F(x) for x in comps)
    if elem in ("i32", "u32"):
        return Vec(int(x) for x in comps)                    # f32 -> i32: truncation toward zero (int() of a float32)
    return Vec(bool(x) for x in comps)


def _div(a, b):
    if isinstance(a, Vec) or isinstance(b, Vec):
        if isinstance(a, Vec) and not isinstance(b, Vec):
            if a._is_int():
                return Vec(_div(x, b) for x in a.c)
            return a * (F(1.0) / F(b))                       # N2: vector / scalar = vector * (1 / scalar)
        av = a.c if isinstance(a, Vec) else (_weak(a),) * len(b.c)
        return Vec(_div(x, y) for x, y in zip(av, b.c))
    if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool):
        if b == 0:
            return 0
        q = abs(a) // abs(b)                                 # i32 division truncates toward zero
        return q if (a >= 0) == (b >= 0) else -q
    with np.errstate(all="ignore"):
        return a / b                                         # f32 / f32; AbstractFloat / AbstractFloat stays binary64 (Python float)


def _mod(a, b):
    if isinstance(a, Vec):
        bv = b.c if isinstance(b, Vec) else (b,) * len(a.c)
        return Vec(_mod(x, y) for x, y in zip(a.c, bv))
    if isinstance(a, int) and isinstance(b, int):
        return int(np.fmod(a, b))
    a, b = F(a), F(b)
    with np.errstate(all="ignore"):
        return a - b * np.trunc(a / b)                       # WGSL float %: truncated remainder


def _arr1(x):
    return np.asarray([x], dtype=np.float32)


def _fmax(a, b): return b if a < b else a                    # N5: compare-select
def _fmin(a, b): return b if b < a else a


def _map(fn, *xs):
    vs = [x for x in xs if isinstance(x, Vec)]
    if not vs:
        return fn(*[_weak(x) for x in xs])
    n = len(vs[0].c)
    cols = [x.c if isinstance(x, Vec) else (_weak(x),) * n for x in xs]
    return Vec(fn(*col) for col in zip(*cols))


def bi_dot(a, b): return (a.c[0] * b.c[0] + a.c[1] * b.c[1]) + a.c[2] * b.c[2] if len(a.c) == 3 else sum_pairs(a, b)


def sum_pairs(a, b):
    acc = a.c[0] * b.c[0]
    for x, y in zip(a.c[1:], b.c[1:]):
        acc = acc + x * y
    return acc


def bi_length(a):
    with np.errstate(all="ignore"):
        return np.sqrt(bi_dot(a, a)) if isinstance(a, Vec) else abs(F(a))


def bi_distance(a, b): return bi_length(a - b)
def bi_normalize(a): return _div(a, bi_length(a))


def bi_cross(a, b):
    return Vec((a.c[1] * b.c[2] - a.c[2] * b.c[1], a.c[2] * b.c[0] - a.c[0] * b.c[2], a.c[0] * b.c[1] - a.c[1] * b.c[0]))


def bi_determinant(m):
    c0, c1, c2 = (c.c for c in m.cols)
    return (c0[0] * (c1[1] * c2[2] - c2[1] * c1[2]) - c1[0] * (c0[1] * c2[2] - c2[1] * c0[2])) + c2[0] * (c0[1] * c1[2] - c1[1] * c0[2])


def bi_abs(a): return _map(lambda x: abs(x), a)
def bi_floor(a): return _map(lambda x: np.floor(x), a)


def bi_sqrt(a):
    with np.errstate(all="ignore"):
        return _map(lambda x: np.sqrt(F(x)), a)


def bi_inverseSqrt(a):
    with np.errstate(all="ignore"):
        return F(1.0) / np.sqrt(F(a))


def bi_min(a, b):
    if isinstance(a, int) and isinstance(b, int):
        return min(a, b)
    return _map(_fmin, a, b)


def bi_max(a, b):
    if isinstance(a, int) and isinstance(b, int):
        return max(a, b)
    return _map(_fmax, a, b)


def bi_clamp(x, lo, hi): return _map(lambda v, l, h: _fmin(_fmax(v, l), h), x, lo, hi)
def bi_mix(a, b, t): return _map(lambda x, y, s: x * (F(1.0) - s) + y * s, a, b, t)


def bi_smoothstep(e0, e1, x):
    e0, e1, x = F(e0), F(e1), F(x)
    with np.errstate(all="ignore"):
        t = _fmin(_fmax((x - e0) / (e1 - e0), F(0.0)), F(1.0))
    return t * t * (F(3.0) - F(2.0) * t)


def bi_all(v): return all(v.c) if isinstance(v, Vec) else bool(v)
def bi_sin(x): return WB.sin(F(x))
def bi_cos(x): return WB.cos(F(x))
def bi_tan(x): return WB.tan(F(x))
def bi_acos(x): return WB.acos(F(x))
def bi_atan2(y, x): return WB.atan2(F(y), F(x))
def bi_i32(x): return int(x)
def bi_f32(x): return F(x)
def bi_u32(x): return int(x)


def _pow1(x, y):
    y = float(F(y))                                          # the exponent as the f32 the call receives
    x = F(x)
    if y == 2.0:
        return x * x
    if y == 4.0:
        return (x * x) * (x * x)
    if y == 5.0:
        return ((x * x) * (x * x)) * x
    if y == float(F(-0.001)):
        return WB.pow_m001(x)
    return F(_LIBM.powf(float(x), float(F(y))))              # pow(., 1.3): glibc's powf, as oracle/ray_oracle.c calls it (N4; numpy's
                                                             # float32 power is a vectorised approximation that differs from it in the last place)


def bi_pow(x, y): return _map(_pow1, x, y)


class Texture:
    def __init__(self, rgba8=None, f32img=None):
        self.rgba8, self.img = rgba8, f32img

    @property
    def size(self):
        a = self.rgba8 if self.rgba8 is not None else self.img
        return a.shape[1], a.shape[0]


def bi_textureDimensions(t):
    w, h = t.size
    return Vec((w, h))


def bi_textureLoad(t, p, lvl):
    w, h = t.size
    x = min(max(p.c[0], 0), w - 1); y = min(max(p.c[1], 0), h - 1)
    return Vec(F(v) for v in t.img[y, x])


def bi_textureSampleLevel(t, s, uv, lvl):
    with np.errstate(all="ignore"):
        return Vec(WB.sample_bilinear(t.rgba8, uv.c[0], uv.c[1]))


class StoreTarget:
    def __init__(self, w, h):
        self.w, self.h = w, h
        self.out = {}

    @property
    def size(self):
        return self.w, self.h


def bi_textureStore(t, p, v):
    t.out[(p.c[0], p.c[1])] = tuple(v.c)


class StorageArray:
    """array<T, N> inside a storage buffer: elements are built on access from numpy rows."""

    def __init__(self, rows, make):
        self.rows, self.make = rows, make

    def __getitem__(self, i):
        i = int(i)
        if i < 0 or i >= len(self.rows):                     # robust buffer access: clamp (never reached by valid scenes)
            i = min(max(i, 0), len(self.rows) - 1)
        return self.make(self.rows[i])


# ------------------------------------------------------------------------------------------------------------------ code generation
BUILTINS = {"normalize", "pow", "length", "dot", "abs", "distance", "max", "cross", "textureStore", "min", "clamp", "textureLoad", "sqrt", "mix",
            "determinant", "all", "textureSampleLevel", "sin", "cos", "atan2", "tan", "smoothstep", "inverseSqrt", "floor", "acos",
            "textureDimensions", "i32", "f32", "u32"}


class Gen:
    def __init__(self, decls):
        self.decls = decls
        self.fns = {d[1]: d for d in decls if d[0] == "fn"}
        self.structs = {d[1]: d for d in decls if d[0] == "struct"}
        self.consts = [d for d in decls if d[0] == "const"]
        self.gvars = {d[1] for d in decls if d[0] == "gvar"}
        self.uid = 0
        self.lines = []

    def fresh(self, name):
        self.uid += 1
        return "l_%s_%d" % (name, self.uid)

    def lookup(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        if name in self.gvars:
            return "G_" + name
        for c in self.consts:
            if c[1] == name:
                return "C_" + name
        raise NameError("wgsl: unknown identifier %r" % name)

    def ty(self, t):
        return repr(t)

    def expr(self, e):
        k = e[0]
        if k == "num":
            s = e[1]
            if s[-1] == "f" and not s.lower().startswith("0x"):
                return "F(%s)" % s[:-1]
            if s[-1] in "iu":
                return s[:-1]
            if re.fullmatch(r"\d+", s) or s.lower().startswith("0x"):
                return s
            return repr(float(s))                             # AbstractFloat: a Python float (binary64) until it meets an f32
        if k == "bool":
            return "True" if e[1] else "False"
        if k == "paren":
            return "(" + self.expr(e[1]) + ")"
        if k == "id":
            return self.lookup(e[1])
        if k == "un":
            return "(%s%s)" % ("-" if e[1] == "-" else "not ", self.expr(e[2]))
        if k == "bin":
            op, a, b = e[1], self.expr(e[2]), self.expr(e[3])
            if op == "/":
                return "_div(%s, %s)" % (a, b)
            if op == "%":
                return "_mod(%s, %s)" % (a, b)
            if op == "&&":
                return "(%s and %s)" % (a, b)
            if op == "||":
                return "(%s or %s)" % (a, b)
            return "(%s %s %s)" % (a, op, b)
        if k == "member":
            return "%s.%s" % (self.expr(e[1]), e[2])
        if k == "index":
            return "%s[%s]" % (self.expr(e[1]), self.expr(e[2]))
        if k == "call":
            name, targs, args = e[1], e[2], [self.expr(a) for a in e[3]]
            if name in ("vec2", "vec3", "vec4"):
                elem = targs[0][0] if targs else "f32"
                return "mk_vec(%d, %r, %s)" % (int(name[3]), elem, ", ".join(args))
            if name == "mat3x3":
                return "Mat3([%s])" % ", ".join(args)
            if name == "array":
                return "zero(%s)" % self.ty(("array", targs))
            if name in self.structs:
                return "S_%s(%s)" % (name, ", ".join(args))
            if name in self.fns:                              # a user function (checked before the built-ins: none collides)
                ps = self.fns[name][2]
                return "fn_%s(%s)" % (name, ", ".join("cv(%s, %s)" % (self.ty(t), a) for (_, t), a in zip(ps, args)))
            if name in BUILTINS:                              # also when a local shadows the name (`let distance` in hit_torus2d)
                return "bi_%s(%s)" % (name, ", ".join(args))
            raise NameError("wgsl: unknown function %r" % name)
        raise SyntaxError("wgsl: expression node %r" % (k,))

    def assigned(self, body):             # names of the variables a statement list assigns to (roots of the left-hand sides)
        out = set()
        for st in body:
            k = st[0]
            if k == "assign":
                e = st[1]
                while e[0] in ("member", "index", "paren"):
                    e = e[1]
                if e[0] == "id":
                    out.add(e[1])
            elif k == "block":
                out |= self.assigned(st[1])
            elif k == "if":
                out |= self.assigned(st[2]) | (self.assigned(st[3]) if st[3] else set())
            elif k == "while":
                out |= self.assigned(st[2])
            elif k == "for":
                out |= self.assigned([x for x in (st[1], st[3]) if x is not None]) | self.assigned(st[4])
        return out

    def emit(self, ind, s):
        self.lines.append("    " * ind + s)

    def stmts(self, body, ind):
        self.scopes.append({})
        if not body:
            self.emit(ind, "pass")
        for st in body:
            self.stmt(st, ind)
        self.scopes.pop()

    def stmt(self, st, ind):
        k = st[0]
        if k in ("let", "var"):
            _, name, ty, e = st
            val = None
            if e is not None:
                val = "cv(%s, %s)" % (self.ty(ty), self.expr(e)) if ty is not None else "conc(%s)" % self.expr(e)
            else:
                val = "zero(%s)" % self.ty(ty)
            py = self.fresh(name)
            self.emit(ind, "%s = %s" % (py, val))
            self.scopes[-1][name] = py                        # (after the initialiser: `let ray_distance = ... ray_distance ...` sees the outer one)
        elif k == "assign":
            _, lhs, op, rhs = st
            tgt, r = self.expr(lhs), self.expr(rhs)
            if op != "=":
                b = op[0]
                r = "_div(%s, %s)" % (tgt, r) if b == "/" else ("_mod(%s, %s)" % (tgt, r) if b == "%" else "(%s %s %s)" % (tgt, b, r))
            self.emit(ind, "%s = conc(%s)" % (tgt, r))
        elif k == "expr":
            self.emit(ind, self.expr(st[1]))
        elif k == "block":
            self.emit(ind, "if True:")
            self.stmts(st[1], ind + 1)
        elif k == "if":
            self.emit(ind, "if %s:" % self.expr(st[1]))
            self.stmts(st[2], ind + 1)
            if st[3] is not None:
                self.emit(ind, "else:")
                self.stmts(st[3], ind + 1)
        elif k == "while":
            # D1: a loop whose state at the top of a pass is bit-identical to the state at the top of the previous pass can never end
            # (ray.wgsl:425-451 once e_max > 1: h = max(h_temp, h) = h).  The executor leaves such a loop there - the result is what
            # the oracles' "body executed once" computes, since the repeated pass recomputed the same values.
            roots = []
            for nm in sorted(self.assigned(st[2])):
                try:
                    roots.append(self.lookup(nm))
                except NameError:
                    pass
            prev = self.fresh("loopstate")
            self.emit(ind, "%s = None" % prev)
            self.emit(ind, "while %s:" % self.expr(st[1]))
            self.emit(ind + 1, "_k = _key((%s,))" % ", ".join(roots))
            self.emit(ind + 1, "if _k == %s: break" % prev)
            self.emit(ind + 1, "%s = _k" % prev)
            self.stmts(st[2], ind + 1)
        elif k == "for":
            _, init, cond, upd, body = st
            self.scopes.append({})
            if init is not None:
                self.stmt(init, ind)
            self.emit(ind, "while %s:" % (self.expr(cond) if cond is not None else "True"))
            self.stmts(body, ind + 1)                         # (no `continue` in the subset: the update always runs)
            if upd is not None:
                self.stmt(upd, ind + 1)
            self.scopes.pop()
        elif k == "break":
            self.emit(ind, "break")
        elif k == "return":
            self.emit(ind, "return" if st[1] is None else "return %s" % ("cv(%s, %s)" % (self.ty(self.ret), self.expr(st[1])) if self.ret else self.expr(st[1])))
        else:
            raise SyntaxError("wgsl: statement %r" % (k,))

    def module(self):
        self.scopes = [{}]
        for _, name, fields in [d for d in self.decls if d[0] == "struct"]:
            self.emit(0, "class S_%s(Struct):" % name)
            self.emit(1, "FIELDS = %r" % (tuple(fields),))
            self.emit(1, "TYPES = %r" % ({n: t for n, t in fields},))
            self.emit(0, "STRUCTS[%r] = S_%s" % (name, name))
        for _, name, ty, e in self.consts:
            val = self.expr(e)
            self.emit(0, "C_%s = %s" % (name, "cv(%s, %s)" % (self.ty(ty), val) if ty is not None else val))
            self.emit(0, "CONST_SIZES[%r] = C_%s" % (name, name))
        for _, name, params, ret, body in [d for d in self.decls if d[0] == "fn"]:
            self.ret = ret
            self.scopes = [{p: "p_" + p for p, _ in params}]
            self.emit(0, "def fn_%s(%s):" % (name, ", ".join("p_" + p for p, _ in params)))
            self.stmts(body, 1)
        return "\n".join(self.lines)


SKY_SHADER = "/root/reference/src/renderer/shaders/sky.wgsl"
_COMPILED = {}


def compile_source(src, name="<wgsl>"):
    """Translates WGSL text (the subset above) and returns the namespace holding its functions (fn_<name>), constants (C_<name>) and
    struct classes (S_<name>); module-scope `var`s are looked up as G_<name> and must be bound by the caller."""
    decls = Parser(tokenize(src)).module()
    code = Gen(decls).module()
    ns = {k: v for k, v in globals().items() if not k.startswith("__")}
    exec(compile(code, name, "exec"), ns)
    ns["__source__"] = code
    return ns


def compile_shader(path=SHADER):
    """Parses one of the reference's shaders and returns the namespace holding its translated functions (fn_main, fn_trace_ray, ...)."""
    if path in _COMPILED:
        return _COMPILED[path]
    src = open(path).read()
    decls = Parser(tokenize(src)).module()
    code = Gen(decls).module()
    ns = {k: v for k, v in globals().items() if not k.startswith("__")}
    exec(compile(code, "<ray.wgsl>", "exec"), ns)
    ns["__source__"] = code
    _COMPILED[path] = ns
    return ns


# ------------------------------------------------------------------------------------------------------------------ bindings
def _f3(v):
    return Vec(F(x) for x in v)


def bind_scene(ns, camera: bytes, black_hole: bytes, details: bytes, t_temp, t_disk, t_sky, models=()):
    """The shader's bind group (ray.wgsl:5-19) from the same bytes the other oracles and libbhray take (layouts: include/bhray.h)."""
    c = struct.unpack("<3fI3ff", bytes(camera))
    ns["G_camera"] = ns["S_Camera"](_f3(c[0:3]), _f3(c[4:7]), F(c[7]))
    b = struct.unpack("<4f3fi3fi12ff8i", bytes(black_hole))
    m = b[12:24]
    ns["G_black_hole"] = ns["S_BlackHole"](F(b[0]), F(b[1]), F(b[2]), F(b[3]), _f3(b[4:7]), b[7], _f3(b[8:11]), b[11],
                                           Mat3([_f3(m[0:3]), _f3(m[4:7]), _f3(m[8:11])]), F(b[24]))
    d = struct.unpack("<iifififi", bytes(details))
    ns["G_details"] = ns["S_Details"](d[0], d[1], F(d[2]), d[3], F(d[4]), d[5], F(d[6]), d[7])
    ns["G_t_temp"], ns["G_t_disk"], ns["G_t_sky"] = Texture(rgba8=t_temp), Texture(rgba8=t_disk), Texture(rgba8=t_sky)
    ns["G_s_temp"] = ns["G_s_disk"] = ns["G_s_sky"] = None
    ns["G_materials"] = []
    ms = []
    for mdl in models:
        pts = np.ascontiguousarray(mdl["points"], dtype=np.float32).reshape(-1, 4)
        nrm = np.ascontiguousarray(mdl["normals"], dtype=np.float32).reshape(-1, 4)
        tri = np.ascontiguousarray(mdl["triangles"], dtype=np.int32).reshape(-1, 6)
        nodes = np.ascontiguousarray(mdl["nodes"]).view(np.uint8).reshape(-1, 32)
        lut = np.ascontiguousarray(mdl["bvh_lookup"], dtype=np.int32)

        def mk_node(row, S=ns["S_Node"]):
            f = row.view(np.float32); i = row.view(np.int32)
            return S(_f3(f[0:3]), int(i[3]), _f3(f[4:7]), int(i[7]))
        M = object.__new__(ns["S_Model"])
        vals = dict(position=_f3(mdl["position"]), visible=int(mdl.get("visible", 1)), rotation=_f3((0, 0, 0)), point_count=len(pts),
                    normal_count=len(nrm), triangle_count=len(tri), points=StorageArray(pts, lambda r: _f3(r[0:3])),
                    normals=StorageArray(nrm, lambda r: _f3(r[0:3])),
                    triangles=StorageArray(tri, lambda r, S=ns["S_TriangleIndices"]: S(*[int(v) for v in r])),
                    nodes=StorageArray(nodes, mk_node), bvh_lookup=StorageArray(lut, lambda r: int(r)))
        for k_, v_ in vals.items():
            object.__setattr__(M, k_, v_)
        ms.append(M)
    ns["G_models"] = ms


def render_level(ns, size, prev=None, rows=None):
    """One RayPipeline::pass (ray_pipeline.rs:301-309): `main` for every pixel of a (W, H) level.  prev: the previous level's image
    (H', W', 4) float32, or None for the 1x1 base texture (mod.rs:151-168)."""
    w, h = size
    target = StoreTarget(w, h)
    ns["G_color_buffer"] = target
    ns["G_t_prev"] = Texture(f32img=np.zeros((1, 1, 4), np.float32) if prev is None else np.ascontiguousarray(prev, dtype=np.float32))
    main = ns["fn_main"]
    y0, y1 = rows if rows else (0, h)
    with np.errstate(all="ignore"):
        for y in range(y0, y1):
            for x in range(w):
                main(Vec((x, y, 0)))
    out = np.full((h, w, 4), np.nan, dtype=np.float32)
    for (x, y), v in target.out.items():
        out[y, x] = v
    return out


def render_pixels(ns, size, prev, pixels):
    """`main` for the listed (x, y) pixels of a (W, H) level only (samples of a frame too large to execute whole): returns an
    (n, 4) float32 array in the order of `pixels`; prev as in render_level."""
    w, h = size
    target = StoreTarget(w, h)
    ns["G_color_buffer"] = target
    ns["G_t_prev"] = Texture(f32img=np.zeros((1, 1, 4), np.float32) if prev is None else np.ascontiguousarray(prev, dtype=np.float32))
    main = ns["fn_main"]
    out = np.full((len(pixels), 4), np.nan, dtype=np.float32)
    with np.errstate(all="ignore"):
        for i, (x, y) in enumerate(pixels):
            main(Vec((int(x), int(y), 0)))
            v = target.out.get((int(x), int(y)))
            if v is not None:
                out[i] = v
    return out


def render_sky(prev, t_sky):
    """The sky resolve pass: sky.wgsl's `main` for every pixel of the last ladder level (sky_pipeline.rs dispatches it like the ray pass).
    The target is rgba16float (sky.wgsl:1): the stored f32 values are converted with round-to-nearest-even (numpy's cast) - returns
    (H, W, 4) float16."""
    ns = compile_shader(SKY_SHADER)
    prev = np.ascontiguousarray(prev, dtype=np.float32)
    h, w = prev.shape[:2]
    target = StoreTarget(w, h)
    ns["G_color_buffer"] = target
    ns["G_t_prev"] = Texture(f32img=prev)
    ns["G_t_sky"] = Texture(rgba8=t_sky); ns["G_s_sky"] = None
    with np.errstate(all="ignore"):
        for y in range(h):
            for x in range(w):
                ns["fn_main"](Vec((x, y, 0)))
    out = np.zeros((h, w, 4), dtype=np.float32)
    for (x, y), v in target.out.items():
        out[y, x] = v
    with np.errstate(over="ignore"):
        return out.astype(np.float16)


def _rows_job(job):
    (cam, bh, det, tex, models, size, prev, rows) = job
    ns = compile_shader()                                   # (the translated functions close over this one namespace)
    bind_scene(ns, cam, bh, det, *tex, models)
    return rows, render_level(ns, size, prev, rows)[rows[0]:rows[1]]


def render_ladder(camera, black_hole, details, tex, sizes, models=(), processes=8):
    """All levels in order (mod.rs:170-207, 415-417), rows spread over processes.  Returns the list of level images."""
    import multiprocessing as mp
    imgs, prev = [], None
    with mp.get_context("fork").Pool(processes) as pool:
        for (w, h) in sizes:
            step = max(1, (h + processes * 2 - 1) // (processes * 2))
            jobs = [(camera, black_hole, details, tex, models, (w, h), prev, (y, min(h, y + step))) for y in range(0, h, step)]
            img = np.full((h, w, 4), np.nan, dtype=np.float32)
            for rows, part in pool.imap_unordered(_rows_job, jobs):
                img[rows[0]:rows[1]] = part
            imgs.append(img); prev = img
    return imgs

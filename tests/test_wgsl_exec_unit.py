"""CPU: the WGSL-subset interpreter (oracle/wgsl_exec.py) on small programs written for this test - precedence, value semantics, scoping,
AbstractFloat constants, integer and float arithmetic rules, vectors.  Independent of the reference checkout: what the interpreter does
with the reference's shader is only as good as what it does here."""
import numpy as np
import pytest

from oracle import wgsl_exec as W

F = np.float32

SRC = """
struct Inner { v: vec3<f32>, k: i32, }
struct Outer { a: Inner, h: f32, }
const third = 1.0 / 3.0;
const PI: f32 = 3.1415926;
const twice = 2.0 * third;

fn precedence(x: f32, y: f32) -> f32 { return -x * 0.5 + y / 4.0 - x * y; }
fn abstract_once(x: f32) -> f32 { return x * (third - twice); }          // the parenthesis is ONE binary64 subtraction, rounded when it meets x
fn concrete_let() -> f32 { let t = 1.0 / 3.0; return t * 3.0; }          // let concretises: t is the f32 nearest to 1/3
fn int_div(a: i32, b: i32) -> i32 { return a / b; }
fn int_mix(a: i32) -> i32 { return (a - 1) / 2 * 2 + a % 3; }
fn frem(a: f32, b: f32) -> f32 { return a % b; }
fn vec_ops(a: vec3<f32>, b: vec3<f32>, s: f32) -> vec3<f32> { return (a + b * s - a.zxy) / 2.0; }
fn swz(a: vec4<f32>) -> vec3<f32> { return vec3<f32>(1.0, a.rg) + a.xzy; }
fn shadow(x: f32) -> f32 {
    var r = x;
    { let x = r * 2.0; r = x + 1.0; { let r2 = r; let x = r2 * r2; r = x; } }
    if r > 10.0 { let r = 0.5; return r + x; }
    return r;
}
fn copies() -> f32 {
    var o: Outer;
    o.a.v = vec3<f32>(1.0, 2.0, 3.0); o.a.k = 7; o.h = 0.25;
    var p = o;                       // a copy
    p.a.v = p.a.v * 2.0; p.h = 4.0;
    let q = p.a;                     // a copy of the inner struct
    p.a.k = 9;
    return o.a.v.y + p.a.v.y * 10.0 + f32(q.k) * 100.0 + f32(p.a.k) * 1000.0 + o.h;
}
fn touch(i: Inner) -> f32 { var j = i; j.k = j.k + 1; return f32(j.k); }
fn param_copy() -> f32 { var i: Inner; i.k = 4; let a = touch(i); return a * 10.0 + f32(i.k); }
fn arrays() -> i32 {
    var st: array<Inner, 4>;
    var n: u32 = 0;
    for (var i = 0; i < 4; i++) { var e: Inner; e.k = i * i; st[n] = e; n = n + 1; }
    var e2 = st[2]; e2.k = 100;
    var acc = 0;
    while (true) { if n == 0 { break; } n = n - 1; acc += st[n].k; }
    return acc + e2.k;
}
fn loops(n: i32) -> i32 { var s = 0; var i = 0; for (; i < n; i++) { if i == 5 { break; } s += i; } return s * 100 + i; }
fn mixes(a: f32, b: f32, t: f32) -> vec3<f32> { return vec3<f32>(mix(a, b, t), clamp(a, 0.0, 1.0), max(min(a, b), 0.25)); }
fn cmp(a: vec2<i32>, b: vec2<i32>) -> bool { return all(a == b) && !(a.x < b.y) || false; }
fn powers(x: f32) -> vec3<f32> { return vec3<f32>(pow(x, 2.0), pow(x, 5.0), pow(vec3<f32>(x), vec3<f32>(4.0)).z); }
fn distance_shadowed(a: vec3<f32>, b: vec3<f32>) -> f32 { let distance = a - b; return distance(a, b) + distance.x; }
"""


@pytest.fixture(scope="module")
def ns():
    return W.compile_source(SRC)


def v3(*c):
    return W.Vec(F(x) for x in c)


def test_precedence_and_unary_minus(ns):
    x, y = F(1.7), F(-2.3)
    assert ns["fn_precedence"](x, y) == ((-x) * F(0.5) + y / F(4.0)) - x * y


def test_abstract_float_constants_are_binary64_until_they_meet_an_f32(ns):
    assert isinstance(ns["C_third"], float) and ns["C_third"] == 1.0 / 3.0 and isinstance(ns["C_PI"], np.float32)
    x = F(3.0)
    assert ns["fn_abstract_once"](x) == x * F(1.0 / 3.0 - 2.0 * (1.0 / 3.0))
    assert ns["fn_concrete_let"]() == F(1.0 / 3.0) * F(3.0)


def test_integer_division_truncates_and_float_remainder_is_truncated(ns):
    assert [ns["fn_int_div"](a, b) for a, b in ((7, 2), (-7, 2), (7, -2), (-7, -2), (5, 0))] == [3, -3, -3, 3, 0]
    assert ns["fn_int_mix"](8) == (8 - 1) // 2 * 2 + 8 % 3
    assert ns["fn_frem"](F(5.5), F(1.0)) == F(0.5) and ns["fn_frem"](F(-5.5), F(1.0)) == F(-0.5) and ns["fn_frem"](F(7.25), F(2.0)) == F(1.25)


def test_vectors_swizzles_and_scalar_division(ns):
    a, b, s = v3(1, 2, 3), v3(0.5, -1, 4), F(0.3)
    r = ns["fn_vec_ops"](a, b, s)
    half = F(1.0) / F(2.0)                                                           # N2: vector / scalar = vector * (1 / scalar)
    want = [((a.c[i] + b.c[i] * s) - a.c[j]) * half for i, j in ((0, 2), (1, 0), (2, 1))]
    assert list(r.c) == want
    r = ns["fn_swz"](W.Vec(F(x) for x in (1, 2, 3, 4)))
    assert list(r.c) == [F(1) + F(1), F(1) + F(3), F(2) + F(2)]


def test_block_scoping_and_shadowing(ns):
    # r = x; inner: x' = 2r; r = x'+1; innermost: x'' = r*r; r = x''   ->  (2x+1)^2 ; > 10 -> 0.5 + the PARAMETER x
    assert ns["fn_shadow"](F(1.0)) == F(9.0)
    assert ns["fn_shadow"](F(2.0)) == F(0.5) + F(2.0)


def test_structs_arrays_and_parameters_are_values(ns):
    assert ns["fn_copies"]() == F(2.0) + F(4.0) * F(10.0) + F(7) * F(100.0) + F(9) * F(1000.0) + F(0.25)
    assert ns["fn_param_copy"]() == F(5.0) * F(10.0) + F(4.0)
    assert ns["fn_arrays"]() == (0 + 1 + 4 + 9) + 100


def test_loops_break_and_compound_assignment(ns):
    assert ns["fn_loops"](3) == (0 + 1 + 2) * 100 + 3
    assert ns["fn_loops"](9) == (0 + 1 + 2 + 3 + 4) * 100 + 5


def test_builtins_follow_the_literal_conventions(ns):
    a, b, t = F(1.5), F(-0.5), F(0.25)
    r = ns["fn_mixes"](a, b, t)
    assert list(r.c) == [a * (F(1.0) - t) + b * t, F(1.0), F(0.25)]
    x = F(1.1)
    r = ns["fn_powers"](x)
    assert list(r.c) == [x * x, ((x * x) * (x * x)) * x, (x * x) * (x * x)]
    assert ns["fn_cmp"](W.Vec((3, 1)), W.Vec((3, 1))) is True and ns["fn_cmp"](W.Vec((0, 1)), W.Vec((0, 1))) is False


def test_a_local_named_like_a_builtin_does_not_hide_the_call(ns):
    a, b = v3(1, 2, 2), v3(0, 0, 0)
    assert ns["fn_distance_shadowed"](a, b) == F(3.0) + F(1.0)       # `let distance` in hit_torus2d (ray.wgsl:682, 693)


def test_a_loop_that_cannot_end_is_left_after_a_repeated_state():
    ns = W.compile_source("""
fn stuck(h0: f32) -> f32 { var h = h0; var n = 0; while true { n += 1; let t = 0.9 * h; if t > h { break; } h = max(t, h); if n > 1000 { break; } } return f32(n); }
""".replace("n += 1;", ""))
    # without the counter the state (h) repeats at once: the interpreter leaves the loop instead of spinning (D1)
    assert ns["fn_stuck"](F(1.0)) == F(0.0)


def test_syntax_outside_the_subset_is_refused():
    with pytest.raises(SyntaxError):
        W.compile_source("fn f() -> f32 { var i = 0; loop { continue; } return 1.0; }")
    with pytest.raises(NameError):
        W.compile_source("fn f() -> f32 { return undefined_thing; }")

"""CPU / Node: the GLSL-ES-subset interpreter that pins the oracle (oracle/ref_js/glsl2js.js) is itself pinned here.

The STEP_FS / RENDER_FS goldens under tests/golden/ were produced by running the reference's shader TEXT through that
interpreter; a construct it misreads in the same way as the oracle's authors would go unseen.  These tests run shader
snippets WRITTEN FOR THIS FILE (no reference text) through the interpreter and compare every output with the value the
GLSL ES 3.00 specification prescribes, computed by hand below (numpy float32, one rounding per operation):

  * clamp = min(max(x, minVal), maxVal); min(x,y) = y < x ? y : x; max(x,y) = x < y ? y : x   (§8.3; the sign of a zero
    result tells the argument order)
  * mix(x,y,a) = x*(1-a) + y*a                                                               (§8.3)
  * int(float) drops the fractional part (toward zero, §5.4.1); floor() rounds down
  * texture() with NEAREST / CLAMP_TO_EDGE as the page configures its textures (html:438-458): texel
    floor(u*w) clamped to [0, w-1]; an R8 texel reads value/255 in .x
  * float arrays, `for` loops with an int counter, compound assignment, operator precedence and left-to-right
    evaluation, one fp32 rounding per operation (no fused multiply-add), swizzles and constructors, sqrt / length
    — the forms html:324-356 uses.

Every case is then re-run with ONE builtin of the interpreter deliberately broken (tests/_glsl2js_driver.js); each
breakage must change at least one output, i.e. these tests fail if any builtin is perturbed.
"""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT

DRIVER = os.path.join(ROOT, "tests", "_glsl2js_driver.js")
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")

f32 = np.float32
HDR = "#version 300 es\nprecision highp float;\nprecision highp int;\n"

PERTURBATIONS = ["clamp_none", "clamp_lower_only", "max_is_mathmax", "min_is_mathmin", "minmax_swapped", "mix_lerp_form", "mix_reversed",
                 "int_rounds", "int_floors", "floor_truncs", "no_fp32_rounding", "length_unrounded", "tex_repeat", "tex_round",
                 "tex_flip_y", "tex_r8_unscaled"]


def run(src, uniforms=None, textures=None, frags=((0.5, 0.5),), mode="f32", perturb=None):
    job = {"src": HDR + src, "mode": mode, "uniforms": uniforms or {}, "textures": textures or {}, "frags": [list(f) for f in frags],
           "perturb": perturb}
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fh:
        json.dump(job, fh)
        path = fh.name
    try:
        proc = subprocess.run([NODE, DRIVER, path], capture_output=True, text=True, timeout=120)
    finally:
        os.unlink(path)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = proc.stdout
    res = json.loads(out)["results"]

    def dec(x):
        if isinstance(x, list):
            return [dec(v) for v in x]
        return {"-0": -0.0, "nan": float("nan")}.get(x, x) if isinstance(x, str) else float(x)
    return [{k: dec(v) for k, v in r.items()} for r in res]


def same(a, b):
    """Equal including the sign of zero; lists element-wise."""
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    a, b = float(a), float(b)
    if np.isnan(a) or np.isnan(b):
        return np.isnan(a) and np.isnan(b)
    return a == b and np.signbit(a) == np.signbit(b)


# ------------------------------------------------------------------------------------------------------------------
# the cases: (name, shader source, uniforms, textures, fragments, expected outputs per fragment)
# ------------------------------------------------------------------------------------------------------------------
def case_clamp_min_max():
    src = """
uniform float a; uniform float b; uniform float z; uniform float nz;
layout(location=0) out vec4 o0; layout(location=1) out vec4 o1; layout(location=2) out vec4 o2;
void main(){
  o0 = vec4(clamp(a, 0.5, 2.0), clamp(b, 0.5, 2.0), clamp(1.25, 0.5, 2.0), clamp(a, -4.0, -3.5));
  o1 = vec4(max(nz, z), max(z, nz), min(z, nz), min(nz, z));
  o2 = vec4(max(a, 2.0), min(a, 2.0), max(b, 2.0), min(b, 2.0));
}"""
    # clamp: a = -3 -> 0.5 ; b = 7 -> 2 ; inside stays ; both bounds below a=-3 -> min(max(-3,-4),-3.5) = -3.5
    # max(x,y) = x<y ? y : x : max(-0,+0) = -0 (x), max(+0,-0) = +0 (x); min(x,y) = y<x ? y : x : min(+0,-0) = +0, min(-0,+0) = -0
    exp = [{"o0": [0.5, 2.0, 1.25, -3.5], "o1": [-0.0, 0.0, 0.0, -0.0], "o2": [2.0, -3.0, 7.0, 2.0]}]
    return "clamp_min_max", src, {"a": -3.0, "b": 7.0, "z": 0.0, "nz": -0.0}, {}, [(0.5, 0.5)], exp


def case_mix():
    a, b, t = f32(0.1), f32(0.7), f32(0.61)
    spec = f32(f32(a * f32(f32(1.0) - t)) + f32(b * t))            # x*(1-a) + y*a, one rounding per operation
    lerp = f32(a + f32(t * f32(b - a)))
    assert spec != lerp, "pick operands for which the two formulas differ in fp32"
    v = [f32(f32(x * f32(f32(1.0) - t)) + f32(y * t)) for x, y in ((a, b), (b, a), (f32(5.0), f32(-3.0)))]
    src = """
uniform float a; uniform float b; uniform float t;
layout(location=0) out vec4 o0; layout(location=1) out vec4 o1;
void main(){
  vec3 p = vec3(a, b, 5.0); vec3 q = vec3(b, a, -3.0);
  vec3 m = mix(p, q, t);
  o0 = vec4(mix(a, b, t), mix(a, b, 0.0), mix(a, b, 1.0), 0.0);
  o1 = vec4(m, 1.0);
}"""
    exp = [{"o0": [float(spec), float(a), float(b), 0.0], "o1": [float(v[0]), float(v[1]), float(v[2]), 1.0]}]
    return "mix", src, {"a": float(a), "b": float(b), "t": float(t)}, {}, [(0.5, 0.5)], exp


def case_int_floor():
    src = """
uniform float p; uniform float n; uniform vec2 pn;
layout(location=0) out vec4 o0; layout(location=1) out vec4 o1;
void main(){
  int ip = int(p); int in_ = int(n);
  ivec2 iv = ivec2(pn);
  o0 = vec4(float(ip), float(in_), floor(p), floor(n));
  o1 = vec4(float(iv.x), float(iv.y), float(ip * 3 - 1), float(ip + in_ * 2));
}"""
    # int(2.7) = 2, int(-2.7) = -2 ; floor(2.7) = 2, floor(-2.7) = -3 ; ivec2((2.7,-2.7)) = (2,-2) ; 2*3-1 = 5 ; 2 + (-2)*2 = -2
    exp = [{"o0": [2.0, -2.0, 2.0, -3.0], "o1": [2.0, -2.0, 5.0, -2.0]}]
    return "int_floor", src, {"p": 2.7, "n": -2.7, "pn": [2.7, -2.7]}, {}, [(0.5, 0.5)], exp


def case_texture_edges():
    # 4 x 3 RGBA texture whose texel (ix,iy) holds (10*iy + ix, ...), and a 4 x 3 R8 texture of 0/255/51
    w, h = 4, 3
    rgba = []
    for iy in range(h):
        for ix in range(w):
            rgba += [10.0 * iy + ix, 100.0 + ix, 200.0 + iy, 1.0]
    r8 = [0, 255, 51, 0, 255, 0, 0, 51, 51, 51, 255, 255]
    src = """
uniform sampler2D ta; uniform sampler2D tm; uniform vec2 texel;
in vec2 vUV;
layout(location=0) out vec4 o0; layout(location=1) out vec4 o1; layout(location=2) out vec4 o2;
void main(){
  o0 = texture(ta, vUV);
  o1 = vec4(texture(ta, vUV - texel).x, texture(ta, vUV + texel).x, texture(ta, vUV + vec2(-texel.x, texel.y)).x, texture(ta, vUV + vec2(texel.x, -texel.y)).x);
  o2 = vec4(texture(tm, vUV).x, texture(tm, vUV + texel).x, texture(tm, vUV - texel).x, 0.0);
}"""
    tx, ty = f32(1.0 / w), f32(1.0 / h)

    def texel(u, v):               # NEAREST, CLAMP_TO_EDGE: floor(u*w) clamped
        ix = int(np.floor(f32(f32(u) * f32(w))))
        iy = int(np.floor(f32(f32(v) * f32(h))))
        return min(max(ix, 0), w - 1), min(max(iy, 0), h - 1)

    frags, exp = [], []
    centres = [((ix + 0.5) / w, (iy + 0.5) / h) for iy in range(h) for ix in range(w)]          # every texel, incl. 4 corners and 4 edges
    extra = [(0.0, 0.0), (1.0, 1.0), (1.0, 0.0), (0.0, 1.0), (-0.2, 0.4), (1.3, 0.4), (0.4, -0.7), (0.4, 1.9), (0.24999, 0.33334)]
    for (u, v) in centres + extra:
        u, v = f32(u), f32(v)
        frags.append((float(u), float(v)))

        def a_x(du, dv):
            ix, iy = texel(f32(u + du), f32(v + dv))
            return 10.0 * iy + ix
        ix, iy = texel(u, v)
        m = lambda du, dv: float(f32(r8[texel(f32(u + du), f32(v + dv))[1] * w + texel(f32(u + du), f32(v + dv))[0]] / 255.0))   # noqa: E731  (c/255, nearest float)
        exp.append({"o0": [10.0 * iy + ix, 100.0 + ix, 200.0 + iy, 1.0],
                    "o1": [a_x(-tx, -ty), a_x(tx, ty), a_x(-tx, ty), a_x(tx, -ty)],
                    "o2": [m(f32(0), f32(0)), m(tx, ty), m(-tx, -ty), 0.0]})
    tex = {"ta": {"w": w, "h": h, "ch": 4, "data": rgba, "scale": 1.0}, "tm": {"w": w, "h": h, "ch": 1, "data": r8, "scale": 1.0 / 255.0}}
    return "texture_edges", src, {"texel": [float(tx), float(ty)]}, tex, frags, exp


def case_loops_arrays_rounding():
    vals = [f32(x) for x in (0.44204444, 0.13231111, 0.11051111, 0.09231111, 0.11051111, 0.03307778, 0.02307778, 0.02307778, 0.03307778)]
    tau = f32(0.58)
    s = f32(0.0)
    for v in vals:
        s = f32(s + v)                                                  # rho += f[i], sequentially from 0.0
    mx = f32(f32(f32(f32(f32(vals[1] + vals[5]) + vals[8]) - vals[3]) - vals[6]) - vals[7])     # left to right
    relaxed = [f32(v - f32(f32(v - f32(0.1)) / tau)) for v in vals]    # f - (f - feq)/tau with feq = 0.1
    acc = f32(0.0)
    for i, r in enumerate(relaxed):
        if i >= 3:
            acc = f32(acc + r)
        else:
            acc = f32(acc - r)
    # one rounding per operation: a*b + c differs from fma(a,b,c) for these operands
    a, b, c = f32(1.0000001), f32(1.0000001), f32(-1.0000002)
    unfused = f32(f32(a * b) + c)
    fused = f32(np.float64(a) * np.float64(b) + np.float64(c))
    assert unfused != fused
    prec = f32(f32(f32(2.0) + f32(f32(3.0) * f32(4.0))) - f32(f32(10.0) / f32(4.0)))          # 2 + 3*4 - 10/4 = 11.5
    src = """
uniform float tau; uniform float a; uniform float b; uniform float c;
uniform float v0; uniform float v1; uniform float v2; uniform float v3; uniform float v4; uniform float v5; uniform float v6; uniform float v7; uniform float v8;
layout(location=0) out vec4 o0; layout(location=1) out vec4 o1;
float pick(int k){
  if (k == 0) return v0; if (k == 1) return v1; if (k == 2) return v2; if (k == 3) return v3; if (k == 4) return v4;
  if (k == 5) return v5; if (k == 6) return v6; if (k == 7) return v7; return v8;
}
void main(){
  float f[9];
  for (int i = 0; i < 9; i++) f[i] = pick(i);
  float rho = 0.0;
  for (int i = 0; i < 9; i++) rho += f[i];
  float mx = f[1] + f[5] + f[8] - f[3] - f[6] - f[7];
  float g[9];
  for (int i = 0; i < 9; i++) g[i] = f[i] - (f[i] - 0.1) / tau;
  float acc = 0.0;
  for (int i = 0; i < 9; i++) { if (i >= 3) { acc += g[i]; } else { acc -= g[i]; } }
  bool both = (rho > 0.9 && mx < 1.0) || false;
  float flag = 0.0;
  if (both) { flag = 1.0; } else { flag = 2.0; }
  if (!(rho >= 0.0) || mx != mx) flag = 3.0;
  o0 = vec4(rho, mx, acc, flag);
  o1 = vec4(a * b + c, 2.0 + 3.0 * 4.0 - 10.0 / 4.0, -a * b, g[8]);
}"""
    uni = {"tau": float(tau), "a": float(a), "b": float(b), "c": float(c)}
    uni.update({f"v{i}": float(v) for i, v in enumerate(vals)})
    exp = [{"o0": [float(s), float(mx), float(acc), 1.0], "o1": [float(unfused), float(prec), float(f32(f32(-a) * b)), float(relaxed[8])]}]
    return "loops_arrays_rounding", src, uni, {}, [(0.5, 0.5)], exp


def case_sqrt_length_swizzle():
    x, y = f32(0.2153276950120926), f32(0.8589195609092712)
    ln = f32(np.sqrt(f32(f32(x * x) + f32(y * y))))                      # length(vec2): sqrt(x*x + y*y), fp32 throughout
    hyp = f32(np.hypot(np.float64(x), np.float64(y)))
    assert ln != hyp
    src = """
uniform vec2 u;
in vec2 vUV;
layout(location=0) out vec4 o0; layout(location=1) out vec4 o1;
void main(){
  vec4 q = vec4(u, vUV);
  vec2 d = vec2(q.y, q.x);
  o0 = vec4(length(u), sqrt(u.x * u.x + u.y * u.y), q.z, q.w);
  o1 = vec4(d, d.x - d.y, q.r + q.g);
}"""
    exp = [{"o0": [float(ln), float(ln), 0.25, 0.75], "o1": [float(y), float(x), float(f32(y - x)), float(f32(x + y))]}]
    return "sqrt_length_swizzle", src, {"u": [float(x), float(y)]}, {}, [(0.25, 0.75)], exp


CASES = [case_clamp_min_max, case_mix, case_int_floor, case_texture_edges, case_loops_arrays_rounding, case_sqrt_length_swizzle]


@pytest.mark.parametrize("make", CASES, ids=lambda f: f.__name__[5:])
def test_interpreter_follows_glsl_es_300(make):
    name, src, uni, tex, frags, exp = make()
    got = run(src, uni, tex, frags)
    assert len(got) == len(exp)
    for g, e, uv in zip(got, exp, frags):
        for k in e:
            assert same(g[k], e[k]), f"{name} at uv={uv}: {k} = {g[k]}, GLSL ES 3.00 prescribes {e[k]}"


def _all_match(perturb):
    for make in CASES:
        name, src, uni, tex, frags, exp = make()
        got = run(src, uni, tex, frags, perturb=perturb)
        for g, e in zip(got, exp):
            for k in e:
                if not same(g[k], e[k]):
                    return False, name
    return True, None


@pytest.mark.parametrize("perturb", PERTURBATIONS)
def test_every_broken_builtin_is_caught(perturb):
    ok, _ = _all_match(perturb)
    assert not ok, f"the cases above do not notice the interpreter with '{perturb}'"


def test_f64_mode_keeps_doubles():
    """'f64' mode (the fp64 goldens): no rounding to binary32 anywhere."""
    src = """
uniform float a; uniform float b; uniform float c;
layout(location=0) out vec4 o0;
void main(){ o0 = vec4(a * b + c, a / b, sqrt(a), mix(a, b, c)); }"""
    a, b, c = 1.0000001, 3.0, 0.1
    got = run(src, {"a": a, "b": b, "c": c}, mode="f64")[0]["o0"]
    assert got == [a * b + c, a / b, float(np.sqrt(a)), a * (1 - c) + b * c]

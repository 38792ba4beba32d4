'use strict';
/*
 * glsl2js.js — TEST INFRASTRUCTURE (oracle side). Not part of the product path.
 *
 * A small transpiler from the GLSL ES 3.00 *subset* used by the reference's two
 * fragment shaders (STEP_FS_SRC, pages/airfoil_flow_lbm_aerolab.html:222-360 and
 * RENDER_FS_SRC, html:362-422) to JavaScript, so that the reference's own shader
 * TEXT can be executed headless under Node in the build container (no WebGL
 * exists there).  The shader text is read from /root/reference at golden-
 * generation time only; nothing from the reference is stored in this repository.
 *
 * Arithmetic model: every float operation is evaluated in IEEE double and, in
 * 'f32' mode, rounded to binary32 immediately (Math.fround).  For + - * / sqrt
 * this double rounding is innocuous, so 'f32' mode is bit-for-bit "IEEE fp32,
 * literal evaluation order, no FMA contraction" — the arithmetic contract the
 * CPU oracle (oracle/lbm_numpy.py, oracle/lbm_oracle.c) states.  'f64' mode
 * keeps doubles.
 *
 * Supported: global uniform/in/out/const declarations, layout(...) qualifiers,
 * functions, float/int/bool/vec2/vec3/vec4/ivec2 locals, fixed-size float
 * arrays, if/else, for, return, the operators + - * / < > <= >= == != && || !
 * = += -= *= /= ++, single-component swizzles, constructors, and the builtins
 * texture clamp sqrt max min floor mix length.  Anything else throws.
 */

const TYPES = new Set(['void', 'float', 'int', 'bool', 'vec2', 'vec3', 'vec4', 'ivec2', 'sampler2D']);
const VEC_N = { vec2: 2, vec3: 3, vec4: 4, ivec2: 2 };
const SWZ = { x: 0, y: 1, z: 2, w: 3, r: 0, g: 1, b: 2, a: 3 };

function tokenize(src) {
  const toks = [];
  const re = /\s+|\/\/[^\n]*|\/\*[\s\S]*?\*\/|#[^\n]*|(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)|(\d+)|([A-Za-z_]\w*)|(\+\+|--|\+=|-=|\*=|\/=|==|!=|<=|>=|&&|\|\||[-+*\/=<>!(){}\[\];,.?:])/gy;
  let m;
  re.lastIndex = 0;
  while (re.lastIndex < src.length) {
    const at = re.lastIndex;
    m = re.exec(src);
    if (!m) throw new Error('glsl2js: cannot tokenize at ' + at + ': ' + src.slice(at, at + 30));
    if (m[1] !== undefined) toks.push({ k: 'float', v: m[1] });
    else if (m[2] !== undefined) toks.push({ k: 'int', v: m[2] });
    else if (m[3] !== undefined) toks.push({ k: 'id', v: m[3] });
    else if (m[4] !== undefined) toks.push({ k: 'op', v: m[4] });
  }
  toks.push({ k: 'eof', v: '<eof>' });
  return toks;
}

class Scope {
  constructor(parent) { this.parent = parent; this.vars = new Map(); }
  declare(name, info) { this.vars.set(name, info); }
  lookup(name) {
    for (let s = this; s; s = s.parent) if (s.vars.has(name)) return s.vars.get(name);
    return null;
  }
}

function transpile(src) {
  const toks = tokenize(src);
  let p = 0;
  const peek = (o = 0) => toks[p + o];
  const next = () => toks[p++];
  const isOp = (v, o = 0) => peek(o).k === 'op' && peek(o).v === v;
  const isId = (v, o = 0) => peek(o).k === 'id' && peek(o).v === v;
  function expectOp(v) { const t = next(); if (t.k !== 'op' || t.v !== v) throw new Error(`glsl2js: expected '${v}' got '${t.v}'`); }
  function expectId() { const t = next(); if (t.k !== 'id') throw new Error(`glsl2js: expected identifier got '${t.v}'`); return t.v; }

  const globals = new Scope(null);   // name -> {t, code}
  const funcs = new Map();           // name -> {ret, params:[{t,name}]}
  const out = [];                    // generated JS lines
  const uniforms = [], ins = [], outs = [];
  let scope = globals;
  let tmpCounter = 0;

  const isVec = (t) => t in VEC_N;
  const isFloatVec = (t) => t === 'vec2' || t === 'vec3' || t === 'vec4';

  // ---------- expressions: return {c, t} ----------
  function arith(op, a, b) {
    if (a.t === 'float' && b.t === 'float') return { c: `fr(${a.c}${op}${b.c})`, t: 'float' };
    if (a.t === 'int' && b.t === 'int') {
      if (op === '/') throw new Error('glsl2js: integer division unsupported');
      return { c: `(${a.c}${op}${b.c})`, t: 'int' };
    }
    const name = { '+': 'add', '-': 'sub', '*': 'mul', '/': 'div' }[op];
    if (isFloatVec(a.t) && a.t === b.t) return { c: `V.${name}vv(${a.c},${b.c})`, t: a.t };
    if (isFloatVec(a.t) && b.t === 'float') return { c: `V.${name}vs(${a.c},${b.c})`, t: a.t };
    if (a.t === 'float' && isFloatVec(b.t)) return { c: `V.${name}sv(${a.c},${b.c})`, t: b.t };
    throw new Error(`glsl2js: no '${op}' for ${a.t},${b.t}`);
  }

  function constructor(t, args) {
    if (t === 'float') {
      if (args.length !== 1 || !(args[0].t === 'int' || args[0].t === 'float')) throw new Error('float() ctor');
      return { c: `(${args[0].c})`, t: 'float' };
    }
    if (t === 'int') {
      if (args.length !== 1) throw new Error('int() ctor');
      return { c: `Math.trunc(${args[0].c})`, t: 'int' };
    }
    const n = VEC_N[t];
    const parts = [];
    if (args.length === 1 && (args[0].t === 'float' || args[0].t === 'int')) {
      const tmp = `_s${tmpCounter++}`;
      // splat; evaluate once through an IIFE-free comma trick
      const conv = t === 'ivec2' ? `Math.trunc(${args[0].c})` : args[0].c;
      return { c: `V.splat(${n},${conv})`, t };
    }
    if (args.length === 1 && isVec(args[0].t) && VEC_N[args[0].t] === n) {
      return { c: t === 'ivec2' ? `V.trunc(${args[0].c})` : `V.copy(${args[0].c})`, t };
    }
    let count = 0;
    for (const a of args) {
      if (a.t === 'float' || a.t === 'int') { parts.push(t === 'ivec2' ? `Math.trunc(${a.c})` : a.c); count += 1; }
      else if (isFloatVec(a.t) && t !== 'ivec2') { parts.push(`...${a.c}`); count += VEC_N[a.t]; }
      else throw new Error(`glsl2js: ctor ${t} from ${a.t} unsupported`);
    }
    if (count !== n) throw new Error(`glsl2js: ctor ${t} needs ${n} components, got ${count}`);
    return { c: `[${parts.join(',')}]`, t };
  }

  function builtin(name, a) {
    const T = a.map((x) => x.t).join(',');
    switch (name) {
      case 'texture':
        if (T !== 'sampler2D,vec2') throw new Error('texture() args ' + T);
        return { c: `tex(${a[0].c},${a[1].c})`, t: 'vec4' };
      case 'clamp':
        if (T !== 'float,float,float') throw new Error('clamp() args ' + T);
        return { c: `V.clamp(${a[0].c},${a[1].c},${a[2].c})`, t: 'float' };
      case 'sqrt':
        if (T !== 'float') throw new Error('sqrt() args ' + T);
        return { c: `fr(Math.sqrt(${a[0].c}))`, t: 'float' };
      case 'floor':
        if (T !== 'float') throw new Error('floor() args ' + T);
        return { c: `Math.floor(${a[0].c})`, t: 'float' };
      case 'max':
      case 'min':
        if (T !== 'float,float') throw new Error(name + '() args ' + T);
        return { c: `V.${name}(${a[0].c},${a[1].c})`, t: 'float' };
      case 'length':
        if (T !== 'vec2') throw new Error('length() args ' + T);
        return { c: `V.length2(${a[0].c})`, t: 'float' };
      case 'mix':
        if (T === 'float,float,float') return { c: `V.mixs(${a[0].c},${a[1].c},${a[2].c})`, t: 'float' };
        if (isFloatVec(a[0].t) && a[1].t === a[0].t && a[2].t === 'float') return { c: `V.mixv(${a[0].c},${a[1].c},${a[2].c})`, t: a[0].t };
        throw new Error('mix() args ' + T);
      default:
        return null;
    }
  }

  function primary() {
    const t = next();
    if (t.k === 'float') {
      let v = t.v; if (v.startsWith('.')) v = '0' + v; if (v.endsWith('.')) v += '0';
      return { c: `fr(${v})`, t: 'float' };
    }
    if (t.k === 'int') return { c: t.v, t: 'int' };
    if (t.k === 'op' && t.v === '(') { const e = expr(); expectOp(')'); return { c: `(${e.c})`, t: e.t }; }
    if (t.k === 'id') {
      if (t.v === 'true' || t.v === 'false') return { c: t.v, t: 'bool' };
      if (isOp('(')) { // call or constructor
        next();
        const args = [];
        if (!isOp(')')) { do { args.push(assign()); } while (isOp(',') && next()); }
        expectOp(')');
        if (TYPES.has(t.v)) return constructor(t.v, args);
        const b = builtin(t.v, args);
        if (b) return b;
        const f = funcs.get(t.v);
        if (!f) throw new Error('glsl2js: unknown function ' + t.v);
        if (f.params.length !== args.length) throw new Error('glsl2js: arity ' + t.v);
        f.params.forEach((q, i) => { if (q.t !== args[i].t) throw new Error(`glsl2js: ${t.v} arg ${i}: ${args[i].t} vs ${q.t}`); });
        return { c: `${t.v}(${args.map((x) => x.c).join(',')})`, t: f.ret };
      }
      const v = scope.lookup(t.v);
      if (!v) throw new Error('glsl2js: undeclared ' + t.v);
      return { c: v.code, t: v.t, lvalue: true, arr: v.arr };
    }
    throw new Error(`glsl2js: unexpected token '${t.v}'`);
  }

  function postfix() {
    let e = primary();
    for (;;) {
      if (isOp('.')) {
        next(); const f = expectId();
        if (!isVec(e.t)) throw new Error('glsl2js: field of non-vector ' + e.t);
        if (f.length !== 1 || !(f in SWZ) || SWZ[f] >= VEC_N[e.t]) throw new Error('glsl2js: swizzle .' + f + ' unsupported');
        e = { c: `${e.c}[${SWZ[f]}]`, t: e.t === 'ivec2' ? 'int' : 'float', lvalue: e.lvalue };
      } else if (isOp('[')) {
        next(); const i = expr(); expectOp(']');
        if (i.t !== 'int') throw new Error('glsl2js: non-int index');
        if (!e.arr) throw new Error('glsl2js: indexing a non-array');
        e = { c: `${e.c}[${i.c}]`, t: e.t, lvalue: true };
      } else if (isOp('++')) {
        next();
        if (e.t !== 'int' || !e.lvalue) throw new Error('glsl2js: ++ on ' + e.t);
        e = { c: `${e.c}++`, t: 'int' };
      } else break;
    }
    return e;
  }

  function unary() {
    if (isOp('-')) {
      next(); const e = unary();
      if (e.t === 'float' || e.t === 'int') return { c: `(-${e.c})`, t: e.t };
      if (isFloatVec(e.t)) return { c: `V.neg(${e.c})`, t: e.t };
      throw new Error('glsl2js: unary - on ' + e.t);
    }
    if (isOp('!')) { next(); const e = unary(); if (e.t !== 'bool') throw new Error('! on ' + e.t); return { c: `(!${e.c})`, t: 'bool' }; }
    if (isOp('+')) { next(); return unary(); }
    return postfix();
  }
  function mulExpr() { let a = unary(); while (isOp('*') || isOp('/')) { const op = next().v; a = arith(op, a, unary()); } return a; }
  function addExpr() { let a = mulExpr(); while (isOp('+') || isOp('-')) { const op = next().v; a = arith(op, a, mulExpr()); } return a; }
  function relExpr() {
    let a = addExpr();
    while (isOp('<') || isOp('>') || isOp('<=') || isOp('>=')) {
      const op = next().v; const b = addExpr();
      if (a.t !== b.t || !(a.t === 'float' || a.t === 'int')) throw new Error(`glsl2js: compare ${a.t} ${op} ${b.t}`);
      a = { c: `(${a.c}${op}${b.c})`, t: 'bool' };
    }
    return a;
  }
  function eqExpr() {
    let a = relExpr();
    while (isOp('==') || isOp('!=')) {
      const op = next().v; const b = relExpr();
      if (a.t !== b.t || isVec(a.t)) throw new Error(`glsl2js: equality ${a.t} ${op} ${b.t}`);
      a = { c: `(${a.c}${op}=${b.c})`, t: 'bool' };
    }
    return a;
  }
  function andExpr() { let a = eqExpr(); while (isOp('&&')) { next(); const b = eqExpr(); if (a.t !== 'bool' || b.t !== 'bool') throw new Error('&& types'); a = { c: `(${a.c}&&${b.c})`, t: 'bool' }; } return a; }
  function orExpr() { let a = andExpr(); while (isOp('||')) { next(); const b = andExpr(); if (a.t !== 'bool' || b.t !== 'bool') throw new Error('|| types'); a = { c: `(${a.c}||${b.c})`, t: 'bool' }; } return a; }
  function assign() {
    const a = orExpr();
    if (isOp('=') || isOp('+=') || isOp('-=') || isOp('*=') || isOp('/=')) {
      const op = next().v;
      if (!a.lvalue) throw new Error('glsl2js: assignment to non-lvalue');
      const b = assign();
      if (op === '=') {
        if (a.t !== b.t) throw new Error(`glsl2js: assign ${b.t} to ${a.t}`);
        return { c: `${a.c}=${b.c}`, t: a.t };
      }
      const r = arith(op[0], { c: a.c, t: a.t }, b);
      if (r.t !== a.t) throw new Error('glsl2js: compound assign type');
      return { c: `${a.c}=${r.c}`, t: a.t };
    }
    return a;
  }
  function expr() { return assign(); }

  // ---------- statements ----------
  function zeroOf(t) {
    if (t === 'float' || t === 'int') return '0';
    if (t === 'bool') return 'false';
    return `V.splat(${VEC_N[t]},0)`;
  }

  function declaration(isGlobal, qualifiers) {
    // at: type name [ '[' n ']' ] [= expr] {, ...} ;
    const t = next().v;
    const lines = [];
    do {
      const name = expectId();
      let arr = 0;
      if (isOp('[')) { next(); const n = next(); if (n.k !== 'int') throw new Error('array size'); arr = parseInt(n.v, 10); expectOp(']'); }
      let init = null;
      if (isOp('=')) {
        next();
        if (arr) throw new Error('glsl2js: array initialisers unsupported');
        init = assign();
        if (init.t !== t) throw new Error(`glsl2js: init ${name}: ${init.t} vs ${t}`);
      }
      const code = isGlobal ? `G.${name}` : name;
      const rhs = arr ? `new Array(${arr}).fill(0)` : (init ? init.c : zeroOf(t));
      (isGlobal ? globals : scope).declare(name, { t, code, arr: arr > 0 });
      if (isGlobal) {
        if (qualifiers.has('uniform')) uniforms.push({ name, t });
        else if (qualifiers.has('in')) ins.push({ name, t });
        else if (qualifiers.has('out')) outs.push({ name, t });
        if (init) lines.push(`${code}=${rhs};`);
        else if (t !== 'sampler2D') lines.push(`${code}=${rhs};`);
      } else {
        lines.push(`let ${name}=${rhs};`);
      }
    } while (isOp(',') && next());
    expectOp(';');
    return lines.join(' ');
  }

  function isDeclStart() {
    if (isId('const')) return true;
    return peek().k === 'id' && TYPES.has(peek().v) && peek(1).k === 'id';
  }

  function statement() {
    if (isOp('{')) return block();
    if (isId('if')) {
      next(); expectOp('('); const c = expr(); expectOp(')');
      if (c.t !== 'bool') throw new Error('if condition type ' + c.t);
      const th = statement();
      let s = `if(${c.c})${th}`;
      if (isId('else')) { next(); s += `else ${statement()}`; }
      return s;
    }
    if (isId('for')) {
      next(); expectOp('(');
      const saved = scope; scope = new Scope(saved);
      let init;
      if (isDeclStart()) init = declaration(false, new Set()); else { init = expr().c + ';'; expectOp(';'); }
      const cond = expr(); expectOp(';');
      const step = expr(); expectOp(')');
      const body = statement();
      scope = saved;
      return `for(${init}${cond.c};${step.c})${body}`;
    }
    if (isId('return')) {
      next();
      if (isOp(';')) { next(); return 'return;'; }
      const e = expr(); expectOp(';');
      return `return ${e.c};`;
    }
    if (isDeclStart()) {
      if (isId('const')) next();
      return declaration(false, new Set());
    }
    const e = expr(); expectOp(';');
    return e.c + ';';
  }

  function block() {
    expectOp('{');
    const saved = scope; scope = new Scope(saved);
    const parts = [];
    while (!isOp('}')) parts.push(statement());
    expectOp('}');
    scope = saved;
    return `{${parts.join('\n')}}`;
  }

  // ---------- top level ----------
  while (peek().k !== 'eof') {
    if (isId('precision')) { while (!isOp(';')) next(); next(); continue; }
    const qualifiers = new Set();
    if (isId('layout')) { next(); expectOp('('); while (!isOp(')')) next(); next(); }
    while (isId('uniform') || isId('in') || isId('out') || isId('const') || isId('highp') || isId('flat')) qualifiers.add(next().v);
    if (!(peek().k === 'id' && TYPES.has(peek().v))) throw new Error(`glsl2js: unexpected '${peek().v}' at top level`);
    if (peek(1).k === 'id' && peek(2).k === 'op' && peek(2).v === '(') {
      // function definition
      const ret = next().v; const name = next().v; expectOp('(');
      const params = [];
      const fscope = new Scope(globals);
      if (!isOp(')')) {
        do { const t = next().v; const n = expectId(); params.push({ t, name: n }); fscope.declare(n, { t, code: n }); } while (isOp(',') && next());
      }
      expectOp(')');
      funcs.set(name, { ret, params });
      scope = fscope;
      const body = block();
      scope = globals;
      out.push(`function ${name}(${params.map((q) => q.name).join(',')})${body}`);
    } else {
      out.push(declaration(true, qualifiers));
    }
  }
  if (!funcs.has('main')) throw new Error('glsl2js: no main()');
  return { js: out.join('\n'), uniforms, ins, outs };
}

/* Runtime: componentwise helpers with per-operation rounding. */
function makeRuntime(mode) {
  const fr = mode === 'f32' ? Math.fround : (x) => x;
  const map2 = (f) => (a, b) => { const r = new Array(a.length); for (let i = 0; i < a.length; i++) r[i] = f(a[i], b[i]); return r; };
  const mapvs = (f) => (a, s) => { const r = new Array(a.length); for (let i = 0; i < a.length; i++) r[i] = f(a[i], s); return r; };
  const mapsv = (f) => (s, a) => { const r = new Array(a.length); for (let i = 0; i < a.length; i++) r[i] = f(s, a[i]); return r; };
  const ops = { add: (x, y) => fr(x + y), sub: (x, y) => fr(x - y), mul: (x, y) => fr(x * y), div: (x, y) => fr(x / y) };
  const V = {
    splat: (n, s) => new Array(n).fill(s),
    copy: (a) => a.slice(),
    trunc: (a) => a.map(Math.trunc),
    neg: (a) => a.map((x) => -x),
    clamp: (x, lo, hi) => Math.min(Math.max(x, lo), hi),   // GLSL ES 3.00 §8.3: min(max(x,minVal),maxVal)
    max: (x, y) => (x < y ? y : x),                         // GLSL: y if x<y else x
    min: (x, y) => (y < x ? y : x),
    length2: (a) => fr(Math.sqrt(fr(fr(a[0] * a[0]) + fr(a[1] * a[1])))),
    mixs: (x, y, t) => fr(fr(x * fr(1 - t)) + fr(y * t)),   // GLSL: x*(1-a)+y*a
    mixv: (x, y, t) => x.map((xi, i) => fr(fr(xi * fr(1 - t)) + fr(y[i] * t))),
  };
  for (const [n, f] of Object.entries(ops)) { V[n + 'vv'] = map2(f); V[n + 'vs'] = mapvs(f); V[n + 'sv'] = mapsv(f); }
  // NEAREST + CLAMP_TO_EDGE sampler (html:438-458). tex = {w,h,ch,data,scale}
  function tex(s, uv) {
    let ix = Math.floor(fr(uv[0] * s.w)), iy = Math.floor(fr(uv[1] * s.h));
    if (ix < 0) ix = 0; else if (ix > s.w - 1) ix = s.w - 1;
    if (iy < 0) iy = 0; else if (iy > s.h - 1) iy = s.h - 1;
    const o = (iy * s.w + ix) * s.ch;
    const d = s.data;
    if (s.ch === 4) return [d[o], d[o + 1], d[o + 2], d[o + 3]];
    return [fr(d[o] * s.scale), 0, 0, 1];   // unsigned normalized: c / 255, then the nearest float
  }
  return { fr, V, tex };
}

/* Compile a fragment shader to {G, main, meta}. */
function compileShader(src, mode) {
  const tr = transpile(src);
  const rt = makeRuntime(mode);
  const factory = new Function('fr', 'V', 'tex', `'use strict'; const G={};\n${tr.js}\nreturn {G, main};`);
  const mod = factory(rt.fr, rt.V, rt.tex);
  return { G: mod.G, main: mod.main, uniforms: tr.uniforms, ins: tr.ins, outs: tr.outs, fr: rt.fr, js: tr.js };
}

module.exports = { transpile, compileShader, makeRuntime };

'use strict';
/* Test driver for oracle/ref_js/glsl2js.js (tests/test_glsl2js_semantics.py): compiles a GLSL ES 3.00 snippet written
 * by the test, optionally with ONE builtin of the interpreter's runtime deliberately broken, runs main() for each
 * fragment of the job and prints the `out` variables as JSON.  No reference text is involved. */
const fs = require('fs');
const path = require('path');
const { transpile, makeRuntime } = require(path.join(__dirname, '..', 'oracle', 'ref_js', 'glsl2js.js'));

const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const tr = transpile(job.src);
const rt = makeRuntime(job.mode);
let js = tr.js;
let fr = rt.fr, V = Object.assign({}, rt.V), tex = rt.tex;
switch (job.perturb || '') {
  case '': break;
  case 'clamp_none': V.clamp = (x) => x; break;
  case 'clamp_lower_only': V.clamp = (x, lo) => Math.max(x, lo); break;
  case 'max_is_mathmax': V.max = (x, y) => Math.max(x, y); break;          // differs on (-0, +0)
  case 'min_is_mathmin': V.min = (x, y) => Math.min(x, y); break;
  case 'minmax_swapped': { const a = V.max; V.max = V.min; V.min = a; break; }
  case 'mix_lerp_form': V.mixs = (x, y, t) => fr(x + fr(t * fr(y - x))); V.mixv = (x, y, t) => x.map((xi, i) => fr(xi + fr(t * fr(y[i] - xi)))); break;
  case 'mix_reversed': V.mixs = (x, y, t) => fr(fr(y * fr(1 - t)) + fr(x * t)); V.mixv = (x, y, t) => x.map((xi, i) => fr(fr(y[i] * fr(1 - t)) + fr(xi * t))); break;
  case 'int_rounds': js = js.split('Math.trunc(').join('Math.round('); V.trunc = (a) => a.map(Math.round); break;
  case 'int_floors': js = js.split('Math.trunc(').join('Math.floor('); V.trunc = (a) => a.map(Math.floor); break;
  case 'floor_truncs': js = js.split('Math.floor(').join('Math.trunc('); break;
  case 'no_fp32_rounding': fr = (x) => x; { const r2 = makeRuntime('f64'); V = Object.assign({}, r2.V); tex = r2.tex; } break;
  case 'length_unrounded': V.length2 = (a) => fr(Math.hypot(a[0], a[1])); break;
  case 'tex_repeat': tex = (s, uv) => { let ix = Math.floor(fr(uv[0] * s.w)), iy = Math.floor(fr(uv[1] * s.h)); ix = ((ix % s.w) + s.w) % s.w; iy = ((iy % s.h) + s.h) % s.h;
                                       const o = (iy * s.w + ix) * s.ch, d = s.data; return s.ch === 4 ? [d[o], d[o + 1], d[o + 2], d[o + 3]] : [d[o] * s.scale, 0, 0, 1]; }; break;
  case 'tex_round': tex = (s, uv) => { let ix = Math.round(fr(uv[0] * s.w)), iy = Math.round(fr(uv[1] * s.h)); ix = Math.min(Math.max(ix, 0), s.w - 1); iy = Math.min(Math.max(iy, 0), s.h - 1);
                                      const o = (iy * s.w + ix) * s.ch, d = s.data; return s.ch === 4 ? [d[o], d[o + 1], d[o + 2], d[o + 3]] : [d[o] * s.scale, 0, 0, 1]; }; break;
  case 'tex_flip_y': tex = (s, uv) => rt.tex(s, [uv[0], fr(1 - uv[1])]); break;
  case 'tex_r8_unscaled': tex = (s, uv) => { const r = rt.tex(s, uv); return s.ch === 4 ? r : [r[0] / s.scale, 0, 0, 1]; }; break;
  default: throw new Error('unknown perturbation ' + job.perturb);
}
const mod = new Function('fr', 'V', 'tex', `'use strict'; const G={};\n${js}\nreturn {G, main};`)(fr, V, tex);
const G = mod.G;
for (const [k, v] of Object.entries(job.uniforms || {})) G[k] = v;
for (const [k, t] of Object.entries(job.textures || {})) G[k] = { w: t.w, h: t.h, ch: t.ch, data: (job.mode === 'f32' ? Float32Array : Float64Array).from(t.data), scale: t.scale };
const enc = (x) => (Array.isArray(x) ? x.map(enc) : (Object.is(x, -0) ? '-0' : (Number.isNaN(x) ? 'nan' : x)));
const results = [];
for (const uv of job.frags) {
  G.vUV = uv;
  mod.main();
  const o = {};
  for (const d of tr.outs) o[d.name] = enc(G[d.name]);
  results.push(o);
}
process.stdout.write(JSON.stringify({ outs: tr.outs.map((d) => d.name), results }));

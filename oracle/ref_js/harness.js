'use strict';
/*
 * harness.js — TEST INFRASTRUCTURE (oracle side). Not part of the product path.
 *
 * Runs the reference's OWN code headless under Node in the build container to
 * produce golden vectors (SURVEY.md §8c, Appendix C):
 *   - the pure-JS helpers of pages/airfoil_flow_lbm_aerolab.html are sliced out
 *     of the HTML text at run time (by marker, never stored here) and eval'd
 *     with the lattice size NX,NY overridden (html:76 makes them constants);
 *   - the two GLSL fragment shaders (html:222-360, 362-422) are transpiled by
 *     glsl2js.js and executed per lattice site with a NEAREST/CLAMP_TO_EDGE
 *     sampler model, reproducing simStep()/renderField() (html:510-545).
 * Only /root/reference is read; outputs are data (JSON + raw arrays).
 *
 * usage: node harness.js <reference.html> <job.json>      (results → stdout JSON,
 *        bulk arrays → files named in the job)
 */
const fs = require('fs');
const path = require('path');
const { compileShader } = require('./glsl2js.js');

function between(txt, startMarker, endMarker, what) {
  const a = txt.indexOf(startMarker);
  if (a < 0) throw new Error('harness: start marker not found for ' + what);
  const b = txt.indexOf(endMarker, a + startMarker.length);
  if (b < 0) throw new Error('harness: end marker not found for ' + what);
  return txt.slice(a, b);
}
function lineMatching(txt, re, what) {
  const m = txt.match(re);
  if (!m) throw new Error('harness: line not found for ' + what);
  return m[0];
}

function loadReference(htmlPath, nx, ny, opts) {
  const txt = fs.readFileSync(htmlPath, 'utf8');
  let domain = lineMatching(txt, /const DX0=[^;\n]*;/, 'domain (html:73)');
  if (opts && opts.dy_half !== undefined && opts.dy_half !== null) {
    // build extension for non-2:1 grids: only the y half-height changes (SURVEY §7)
    const before = domain;
    domain = domain.replace(/DY0=[-0-9.eE]+/, 'DY0=' + (-opts.dy_half)).replace(/DY1=[-0-9.eE]+/, 'DY1=' + opts.dy_half);
    if (domain === before && opts.dy_half !== 0.46) throw new Error('harness: DY override failed');
  }
  const consts = [
    lineMatching(txt, /const CHORD_L\s*=[^;\n]*;/, 'CHORD_L (html:77)'),
    lineMatching(txt, /const TAU\s*=[^;\n]*;/, 'TAU (html:78)'),
    lineMatching(txt, /const NU_L\s*=[^;\n]*;/, 'NU_L (html:79)'),
    lineMatching(txt, /const STEPS_PER_FRAME\s*=[^;\n]*;/, 'STEPS_PER_FRAME (html:80)'),
    lineMatching(txt, /const VORT_SCALE\s*=[^;\n]*;/, 'VORT_SCALE (html:528)'),
  ].join('\n');
  const geometry = between(txt, 'function naca4(', '// scanline polygon', 'geometry (html:99-157)') +
    between(txt, 'function rasterMask(', '// ================= WebGL setup', 'rasterMask (html:160-182)');
  const stepSrcDecl = between(txt, 'const STEP_FS_SRC=`', 'const RENDER_FS_SRC=`', 'STEP_FS_SRC (html:222-360)');
  const renderSrcDecl = between(txt, 'const RENDER_FS_SRC=`', 'let stepProg', 'RENDER_FS_SRC (html:362-422)');
  let init = between(txt, 'function equilibriumInitData(', 'function initSim(', 'equilibriumInitData (html:474-490)');
  if (opts && opts.init_f64) init = init.replace(/Float32Array/g, 'Float64Array');
  const buildGeometry = between(txt, 'function buildGeometry(', 'function applyGeometry(', 'buildGeometry (html:559-577)');
  const macroDecl = lineMatching(txt, /const macro=new Float32Array\([^;\n]*;/, 'macro (html:547)');
  const fields = between(txt, 'const Ufield=new Float32Array', 'function sampleScalar(', 'updateFieldsFromMacro (html:590-614)');
  const forces = between(txt, 'let CLsmooth=null', '// ================= colour maps', 'computeForces (html:641-700)');
  const cmaps = between(txt, 'function lerpScale(', '// ================= particle trail', 'colour maps (html:704-719)');
  const sampling = between(txt, 'function sampleScalar(', 'let CLsmooth=null', 'sampleScalar/sampleUV (html:616-639)');
  const particles = between(txt, 'let parts=[];', 'function w2cX(', 'particles (html:727-808)');

  const body = `
    ${domain}
    ${consts}
    ${geometry}
    ${stepSrcDecl}
    ${renderSrcDecl}
    ${init}
    let U0=0.06; let sol=null;
    ${buildGeometry}
    ${macroDecl}
    ${fields}
    ${forces}
    ${cmaps}
    ${sampling}
    ${particles}
    return {
      sampleUV, advect, spawn, initParts, getParts:()=>parts, STALL_SPEED2, STALL_DRAIN,
      DX0,DX1,DY0,DY1,CHORD_L,TAU,NU_L,STEPS_PER_FRAME,VORT_SCALE,NP,
      naca4,clarkY,SHAPES,rotate,panelise,rasterMask,buildGeometry,equilibriumInitData,
      STEP_FS_SRC,RENDER_FS_SRC,macro,
      setU0:(v)=>{U0=v;}, setSol:(s)=>{sol=s;},
      updateFieldsFromMacro, computeForces,
      ranges:()=>({maxS,cpMin,cpMax}), setRanges:(a,b,c)=>{maxS=a;cpMin=b;cpMax=c;},
      fieldsOut:()=>({Ufield,Vfield,CpField}),
      forceState:()=>({CLsmooth,CDsmooth,sepFrac}),
      resetForceState:()=>{CLsmooth=null;CDsmooth=null;sepFrac=0;},
      cmap,cmapCp,cmapVort
    };`;
  const userCoords = (opts && opts.user_coords) ? opts.user_coords : [];
  // eslint-disable-next-line no-new-func
  return new Function('NX', 'NY', 'USER_COORDS', body)(nx, ny, userCoords);
}

function writeArray(file, arr) {
  fs.writeFileSync(file, Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
}

/* Reproduces initSim + simStep (html:492-525) on top of the transpiled STEP_FS. */
function runLBM(R, nx, ny, mask, mode, u0, tau, steps, hooks) {
  const FA = mode === 'f32' ? Float32Array : Float64Array;
  const sh = compileShader(R.STEP_FS_SRC, mode);
  const fr = sh.fr;
  const init = R.equilibriumInitData(u0);
  const n = nx * ny;
  const mk = (d) => ({ w: nx, h: ny, ch: 4, data: FA.from(d) });
  let src = { A: mk(init.dA), B: mk(init.dB), C: mk(init.dC) };
  let dst = { A: mk(init.dA), B: mk(init.dB), C: mk(init.dC) };
  const G = sh.G;
  G.texMask = { w: nx, h: ny, ch: 1, data: mask, scale: 1 / 255 };
  G.texel = [fr(1 / nx), fr(1 / ny)];
  G.gridSize = [nx, ny];
  G.tau = fr(tau);
  G.U0 = fr(u0);
  const uvx = new Array(nx), uvy = new Array(ny);
  for (let ix = 0; ix < nx; ix++) uvx[ix] = fr((ix + 0.5) / nx);
  for (let iy = 0; iy < ny; iy++) uvy[iy] = fr((iy + 0.5) / ny);
  for (let s = 0; s < steps; s++) {
    G.texA = src.A; G.texB = src.B; G.texC = src.C;
    if (hooks && hooks.beforeStep) hooks.beforeStep(s, G);
    const dA = dst.A.data, dB = dst.B.data, dC = dst.C.data;
    for (let iy = 0; iy < ny; iy++) {
      for (let ix = 0; ix < nx; ix++) {
        G.vUV = [uvx[ix], uvy[iy]];
        sh.main();
        const o = (iy * nx + ix) * 4;
        const a = G.outA, b = G.outB, c = G.outC;
        dA[o] = a[0]; dA[o + 1] = a[1]; dA[o + 2] = a[2]; dA[o + 3] = a[3];
        dB[o] = b[0]; dB[o + 1] = b[1]; dB[o + 2] = b[2]; dB[o + 3] = b[3];
        dC[o] = c[0]; dC[o + 1] = c[1]; dC[o + 2] = c[2]; dC[o + 3] = c[3];
      }
    }
    const t = src; src = dst; dst = t;
    if (hooks && hooks.afterStep) hooks.afterStep(s + 1, src);
  }
  // de-interleave into SoA: f[9][ny][nx], rho, ux, uy
  const f = new FA(9 * n), rho = new FA(n), ux = new FA(n), uy = new FA(n);
  for (let i = 0; i < n; i++) {
    for (let k = 0; k < 4; k++) { f[k * n + i] = src.A.data[i * 4 + k]; f[(4 + k) * n + i] = src.B.data[i * 4 + k]; }
    f[8 * n + i] = src.C.data[i * 4];
    rho[i] = src.C.data[i * 4 + 1]; ux[i] = src.C.data[i * 4 + 2]; uy[i] = src.C.data[i * 4 + 3];
  }
  return { f, rho, ux, uy, texC: src.C };
}

/* renderField (html:530-545) with the transpiled RENDER_FS: returns RGB floats per site. */
function runRender(R, nx, ny, mask, mode, texC, fieldMode, u0, maxS, cpMin, cpMax) {
  const sh = compileShader(R.RENDER_FS_SRC, mode);
  const fr = sh.fr, G = sh.G;
  G.texC = texC;
  G.texMask = { w: nx, h: ny, ch: 1, data: mask, scale: 1 / 255 };
  G.texel = [fr(1 / nx), fr(1 / ny)];
  G.fieldMode = fieldMode;
  G.U0 = fr(u0); G.maxS = fr(maxS); G.cpMin = fr(cpMin); G.cpMax = fr(cpMax); G.vortScale = fr(R.VORT_SCALE);
  const FA = mode === 'f32' ? Float32Array : Float64Array;
  const rgb = new FA(nx * ny * 3);
  for (let iy = 0; iy < ny; iy++) for (let ix = 0; ix < nx; ix++) {
    G.vUV = [fr((ix + 0.5) / nx), fr((iy + 0.5) / ny)];
    sh.main();
    const o = (iy * nx + ix) * 3;
    rgb[o] = G.fragColor[0]; rgb[o + 1] = G.fragColor[1]; rgb[o + 2] = G.fragColor[2];
  }
  return rgb;
}

function lattice(R) {
  // e, w, opp as the reference's shader functions define them (html:238-264)
  const sh = compileShader(R.STEP_FS_SRC, 'f32');
  // functions are closed over inside the module; re-run a tiny probe shader built from the same text
  const probe = R.STEP_FS_SRC.replace(/void main\(\)\{[\s\S]*$/, 'void main(){ }');
  const tr = require('./glsl2js.js').transpile(probe);
  const rt = require('./glsl2js.js').makeRuntime('f32');
  // eslint-disable-next-line no-new-func
  const mod = new Function('fr', 'V', 'tex', `'use strict'; const G={};\n${tr.js}\nreturn {dir,wt,opp};`)(rt.fr, rt.V, rt.tex);
  const e = [], w = [], opp = [];
  for (let i = 0; i < 9; i++) { e.push(mod.dir(i)); w.push(mod.wt(i)); opp.push(mod.opp(i)); }
  return { e, w, opp, compiled_ok: !!sh.main };
}

function main() {
  const htmlPath = process.argv[2];
  const job = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
  const nx = job.nx, ny = job.ny;
  const R = loadReference(htmlPath, nx, ny, job.opts || {});
  const res = { nx, ny, consts: { DX0: R.DX0, DX1: R.DX1, DY0: R.DY0, DY1: R.DY1, CHORD_L: R.CHORD_L, TAU: R.TAU, NU_L: R.NU_L, STEPS_PER_FRAME: R.STEPS_PER_FRAME, VORT_SCALE: R.VORT_SCALE, NP: R.NP } };
  const outdir = job.outdir || '.';

  let sol = null;
  if (job.geometry) {
    const g = job.geometry;
    sol = R.buildGeometry(g.shape || 'naca2412', g.aoa);
    R.setSol(sol);
    let count = 0; for (let i = 0; i < sol.IN.length; i++) if (sol.IN[i]) count++;
    res.geometry = { solid_count: count, xp: Array.from(sol.xp), yp: Array.from(sol.yp) };
    if (g.base_coords) res.geometry.base = (job.opts && job.opts.user_coords && job.opts.user_coords.length) ? job.opts.user_coords : R.SHAPES[g.shape]();
    if (g.rotated) res.geometry.rotated = R.rotate(R.SHAPES[g.shape](), g.aoa);
    if (g.mask_file) writeArray(path.join(outdir, g.mask_file), sol.IN);
  }
  if (job.lattice) res.lattice = lattice(R);
  if (job.init) {
    const d = R.equilibriumInitData(job.init.u0);
    res.init = { f: [d.dA[0], d.dA[1], d.dA[2], d.dA[3], d.dB[0], d.dB[1], d.dB[2], d.dB[3], d.dC[0]], macro: [d.dC[1], d.dC[2], d.dC[3]] };
  }
  if (job.run) {
    const r = job.run;
    const hooks = {};
    const trace = [];
    if (r.aoa_schedule) {
      // html:943-947 → applyGeometry: mask replaced mid-run, flow state kept (Appendix A.9)
      hooks.beforeStep = (s, G) => {
        for (const ev of r.aoa_schedule) if (ev.step === s) {
          sol = R.buildGeometry(job.geometry.shape || 'naca2412', ev.aoa); R.setSol(sol);
          G.texMask = { w: nx, h: ny, ch: 1, data: sol.IN, scale: 1 / 255 };
        }
      };
    }
    if (r.u0_schedule) {
      const prev = hooks.beforeStep;
      hooks.beforeStep = (s, G) => { if (prev) prev(s, G); for (const ev of r.u0_schedule) if (ev.step === s) G.U0 = Math.fround(ev.u0); };
    }
    const t0 = Date.now();
    const o = runLBM(R, nx, ny, sol.IN, r.mode, r.u0, r.tau === undefined ? R.TAU : r.tau, r.steps, hooks);
    res.run = { seconds: (Date.now() - t0) / 1000, steps: r.steps, mode: r.mode };
    if (r.state_file) {
      const FA = r.mode === 'f32' ? Float32Array : Float64Array;
      const all = new FA(o.f.length + 3 * nx * ny);
      all.set(o.f, 0); all.set(o.rho, o.f.length); all.set(o.ux, o.f.length + nx * ny); all.set(o.uy, o.f.length + 2 * nx * ny);
      writeArray(path.join(outdir, r.state_file), all);
    }
    if (r.reduce && r.mode === 'f32') {
      // feed the reference's own JS reductions with the macro texture, exactly as readMacro would (html:547-552)
      R.macro.set(o.texC.data);
      const finalU0 = (r.u0_schedule && r.u0_schedule.length) ? r.u0_schedule[r.u0_schedule.length - 1].u0 : r.u0;
      R.setU0(finalU0);
      R.updateFieldsFromMacro();
      res.ranges = R.ranges();
      const fo = R.fieldsOut();
      if (r.fields_file) {
        const all = new Float32Array(3 * nx * ny);
        all.set(fo.Ufield, 0); all.set(fo.Vfield, nx * ny); all.set(fo.CpField, 2 * nx * ny);
        writeArray(path.join(outdir, r.fields_file), all);
      }
      R.resetForceState();
      R.computeForces();
      const f1 = R.forceState();
      R.computeForces();
      const f2 = R.forceState();
      res.forces = { first: f1, second: f2 };
      if (r.tracers) {
        // advect() (html:758-771) on the fields updateFieldsFromMacro just filled
        res.tracers = r.tracers.points.map((q) => { const a = R.advect({ x: q[0], y: q[1] }, r.tracers.dt); return a ? [a.nx, a.ny, a.speed] : null; });
        res.tracer_uv = r.tracers.points.map((q) => R.sampleUV(q[0], q[1]));
      }
      if (r.render_file) {
        const rg = res.ranges;
        const parts = [];
        for (let m = 0; m < 3; m++) parts.push(runRender(R, nx, ny, sol.IN, 'f32', o.texC, m, finalU0, rg.maxS, rg.cpMin, rg.cpMax));
        const all = new Float32Array(parts[0].length * 3);
        parts.forEach((q, i) => all.set(q, i * q.length));
        writeArray(path.join(outdir, r.render_file), all);
      }
    }
  }
  if (job.cmaps) {
    const ts = job.cmaps.t;
    res.cmaps = { speed: ts.map((t) => R.cmap(t)), cp: ts.map((t) => R.cmapCp(t)), vort: ts.map((t) => R.cmapVort(2 * t - 1)) };
  }
  process.stdout.write(JSON.stringify(res));
}

if (require.main === module) main();
module.exports = { loadReference, runLBM, runRender };

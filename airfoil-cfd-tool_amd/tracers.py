"""Tracer particles ("smoke") of the wind-tunnel page.

Host side of the reference's particle system, ``pages/airfoil_flow_lbm_aerolab.html``
727-808: ``spawn`` 730-736, ``initParts`` 737-753, ``stepParticles`` 780-808 (life drain, stall
drain, respawn).  The deterministic part — ``advect`` 758-771 over ``sampleUV`` 616-639 — runs
on the GPU (``wt_advect_tracers``).  The reference seeds with ``Math.random``; here a NumPy
``Generator`` takes its place, so only the deterministic part has exact goldens and the seeding
is compared statistically.  ``step`` returns the segments and their colour-map
arguments; ``draw`` strokes them onto a fading ``compose.TrailLayer`` the way html:781-803 strokes the
particle canvas.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import geometry as geo

STALL_SPEED2 = 3e-6      # html:777
STALL_DRAIN = 0.18       # html:778
NPART_DEFAULT = 2600     # html:728


class Tracers:
    def __init__(self, tunnel, n: int = NPART_DEFAULT, seed: Optional[int] = None):
        self.tunnel = tunnel
        self.rng = np.random.default_rng(seed)
        y_half = tunnel.y_half if tunnel.y_half is not None else geo.domain_y_half(tunnel.nx, tunnel.ny)
        self.window = (geo.DX0, geo.DX1, -y_half, y_half)
        self.x = np.empty(0)
        self.y = np.empty(0)
        self.life = np.empty(0)
        self.lane = np.empty(0)
        self.init_parts(n)

    # html:730-736
    def _spawn(self, edge: np.ndarray, lane: np.ndarray):
        dx0, dx1, dy0, dy1 = self.window
        n = lane.size
        r = self.rng
        at_edge = edge | (r.random(n) < 0.82)
        x = np.where(at_edge, dx0 + 0.001, dx0 + r.random(n) * (dx1 - dx0))
        y = np.where(at_edge, lane, dy0 + r.random(n) * (dy1 - dy0))
        life = np.where(at_edge, 220 + r.random(n) * 300, 150 + r.random(n) * 250)
        return x, y, life

    # html:737-753
    def init_parts(self, n: int) -> None:
        dx0, dx1, dy0, dy1 = self.window
        r = self.rng
        i = np.arange(n)
        centre = r.random(n) < 0.35
        c, half = (dy0 + dy1) / 2, (dy1 - dy0) / 6
        lane = np.where(centre, c + (r.random(n) - 0.5) * 2 * half,
                        dy0 + ((i + 0.5) / n) * (dy1 - dy0) + (r.random(n) - 0.5) * 0.003)
        _, y, life = self._spawn(np.ones(n, bool), lane)
        self.x = dx0 + r.random(n) * (dx1 - dx0) * 0.95
        self.y = y
        self.life = life * r.random(n)
        self.lane = lane

    def resize(self, n: int) -> None:
        """The trails slider (html:961-967)."""
        cur = self.x.size
        if n < cur:
            self.x, self.y, self.life, self.lane = (a[:n] for a in (self.x, self.y, self.life, self.lane))
        elif n > cur:
            dy0, dy1 = self.window[2], self.window[3]
            lane = dy0 + self.rng.random(n - cur) * (dy1 - dy0)
            x, y, life = self._spawn(np.zeros(n - cur, bool), lane)
            self.x, self.y = np.concatenate([self.x, x]), np.concatenate([self.y, y])
            self.life, self.lane = np.concatenate([self.life, life]), np.concatenate([self.lane, lane])

    def draw(self, layer, dt: float = 16.0):
        """stepParticles(dt) including the drawing (html:780-808): fade the layer, advance, stroke the moved particles."""
        from .compose import Canvas
        seg, t = self.step(dt)
        layer.fade()
        layer.stroke(Canvas(layer.s, alloc=False), seg, t, self.window[3])
        return seg, t

    # html:780-808
    def step(self, dt: float = 16.0):
        """One stepParticles(dt): returns (segments [m][4] = x0,y0,x1,y1, t [m]) of the particles that
        moved (t = min(speed/(0.92 maxS), 1), the colour-map argument of html:796-797)."""
        wt = self.tunnel
        xn, yn, speed, ok = wt.engine.advect_tracers(self.x, self.y, dt, wt.u0, self.window)
        stalled = ok & (speed * speed < STALL_SPEED2)
        self.life = self.life - dt * np.where(stalled, STALL_DRAIN, 0.06)
        dead = (~ok) | (self.life <= 0)
        alive = ~dead
        seg = np.stack([self.x[alive], self.y[alive], xn[alive], yn[alive]], axis=1)
        t = np.minimum(speed[alive] / (wt.max_s * 0.92), 1.0)
        self.x = np.where(alive, xn, self.x)
        self.y = np.where(alive, yn, self.y)
        if dead.any():
            x, y, life = self._spawn(np.ones(int(dead.sum()), bool), self.lane[dead])
            self.x[dead], self.y[dead], self.life[dead] = x, y, life
        return seg, t

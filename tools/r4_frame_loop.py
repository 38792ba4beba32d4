#!/usr/bin/env python3
"""Frames per second of the page's loop (frame(), html:902-930: 4 steps, ranges, forces every 3rd frame, tracers, the composited canvas) at the
reference's lattice (320x160, html:76), the build's page default (1024x512) and a large one (4096x2048), and where the time of a frame goes:
device stepping, the small read-backs (ranges, forces), the RGBA read-back, the tracers (device advect + host bookkeeping + strokes) and the
NumPy compositor.  VERDICT r3 item 7.

    python tools/r4_frame_loop.py [frames]         -> two tables per lattice on stdout (profiles/r04_*_frame_loop.txt): the host canvas of
                                                      round 3 (compose.py) and the device canvas (csrc/canvas.hpp)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import airfoil_cfd_tool_amd as pkg                                  # noqa: E402
from airfoil_cfd_tool_amd.compose import TrailLayer                 # noqa: E402
from airfoil_cfd_tool_amd.tracers import Tracers                    # noqa: E402


def timed(fn, sync):
    t0 = time.perf_counter()
    out = fn()
    sync()
    return (time.perf_counter() - t0) * 1e3, out


def run(nx, ny, frames, device_canvas):
    with pkg.WindTunnel(shape="naca2412", nx=nx, ny=ny, aoa_deg=6.0) as wt:
        tracers, layer = Tracers(wt, seed=1), (wt.trail_layer(1) if device_canvas else TrailLayer(1))
        sync = wt.engine.sync
        for _ in range(12):                                          # warm-up: plan tuned, clocks up, first macro emitted
            wt.frame(render=False)
            tracers.draw(layer, 16.0)
        wt.compose_frame(trails=layer)
        acc = {k: 0.0 for k in ("step", "ranges", "forces", "tracers", "rgba", "compose")}
        t_all = time.perf_counter()
        for f in range(frames):
            ms, _ = timed(lambda: wt.sim_step(4), sync); acc["step"] += ms
            ms, _ = timed(wt.update_fields_from_macro, lambda: None); acc["ranges"] += ms
            wt.stat_counter += 1
            if wt.stat_counter % 3 == 0:
                ms, _ = timed(wt.compute_forces, lambda: None); acc["forces"] += ms
            ms, _ = timed(lambda: tracers.draw(layer, 16.0), lambda: None); acc["tracers"] += ms
            if device_canvas:        # wt_canvas_compose: field colours, particle layer, foil, bar, labels per canvas pixel + the 1 MB read-back
                ms, _ = timed(lambda: wt.compose_frame(trails=layer), lambda: None); acc["compose"] += ms
            else:
                ms, img = timed(wt.render_rgba, lambda: None); acc["rgba"] += ms
                from airfoil_cfd_tool_amd import compose
                ms, _ = timed(lambda: compose.compose(img[::-1], wt.geometry.xp, wt.geometry.yp, wt.aoa_deg, 0, wt.y_half_world(), trails=layer, scale=1),
                              lambda: None)
                acc["compose"] += ms
        total = (time.perf_counter() - t_all) * 1e3
        per = {k: v / frames for k, v in acc.items()}
        print(f"{nx}x{ny}: {frames} frames, {total / frames:.2f} ms per frame = {1e3 * frames / total:.1f} frames/s "
              f"({4 * nx * ny * frames / total / 1e3:.0f} MLUPS inside the loop)" + ("  [device canvas]" if device_canvas else ""))
        for k in ("step", "ranges", "forces", "tracers", "rgba", "compose"):
            print(f"    {k:8s} {per[k]:8.3f} ms  {100 * per[k] / (total / frames):5.1f} %")
        # the stepping alone, as the library would run it without a host in the loop
        ms = wt.engine.step_timed(400, wt.tau, wt.u0)
        print(f"    (4 steps back to back on the device: {ms / 100:.3f} ms)")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    for nx, ny in ((320, 160), (1024, 512), (4096, 2048)):
        run(nx, ny, frames if nx < 4096 else max(10, frames // 3), False)      # round 3: NumPy compositor, per-particle strokes in Python
        run(nx, ny, 4 * frames, True)                                          # the canvas on the device (csrc/canvas.hpp)


if __name__ == "__main__":
    main()

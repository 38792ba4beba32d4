"""Streamlit wiring: the drop-in for ``build_lbm_component`` of the reference's analysis page.

``pages/Airfoil_Analysis.py:20-42`` injects ``coords_after`` into the WebGL template and embeds it
(call sites AA.py:1210-1213, 1413-1416, captions AA.py:1397-1432).  This module keeps that
function's name and inputs and renders the MI355X tunnel instead: the component's own controls
(html:20-58: angle of attack, field, flow speed, trails), the read-outs of html:53-57 (CL, CD,
Reynolds, separation) and the composited canvas (compose.py).  Like the page's requestAnimationFrame
loop (html:902-930, which never stops) the image keeps advancing WITHOUT a user action and WITHOUT an end:
where the installed Streamlit has ``st.fragment`` the controls and the canvas live in a fragment that
Streamlit re-runs every `run_every` seconds, for ever, each run advancing `frames_per_update` simulation
frames (a control change re-runs just that fragment and is picked up by the next update); older Streamlit
versions get a loop that streams updates into ``st.empty()`` placeholders until Streamlit stops the script
(a widget change or the end of the session — the rerun then picks the same tunnel up again).  `frames=N`
bounds the stream for batch use and tests.

``streamlit`` is imported lazily (it is not installed in the build image); everything else is the
package's ordinary host API, so a page needs only::

    from airfoil_cfd_tool_amd.streamlit_page import build_lbm_component
    build_lbm_component(result["coords_after"], airfoil_name)
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .compose import TrailLayer
from .tracers import Tracers
from .windtunnel import WindTunnel

_FIELD_LABELS = {"Velocity": "speed", "Pressure (Cp)": "cp", "Vorticity": "vort"}     # html:32-36


def _full_width_kwarg(image_fn) -> dict:
    """The keyword that stretches an image over its column: `use_container_width` where the installed Streamlit has it (1.40+; the older
    `use_column_width` is deprecated there), else `use_column_width` — the same sizing AA.py:42's `components.html(..., height=680)` gives the
    iframe.  Decided from the signature of the placeholder's own `image`, so that no Streamlit version is assumed."""
    import inspect
    try:
        params = inspect.signature(image_fn).parameters
    except (TypeError, ValueError):
        return {}
    if "use_container_width" in params:
        return {"use_container_width": True}
    if "use_column_width" in params:
        return {"use_column_width": True}
    return {}


def _tunnel_for(st, coords_after, airfoil_name: str, nx: int, ny: int, dtype: str) -> WindTunnel:
    """One tunnel per (coordinates, lattice) in the session, like the iframe's one GL context."""
    key = ("wt_amd", hash(tuple((round(float(x), 6), round(float(y), 6)) for x, y in coords_after)), nx, ny, dtype)
    store = st.session_state
    if store.get("wt_amd_key") != key:
        old = store.get("wt_amd_tunnel")
        if old is not None:
            old.close()
        store["wt_amd_tunnel"] = WindTunnel(coords_after, airfoil_name, nx=nx, ny=ny, dtype=dtype)
        store["wt_amd_tracers"] = None
        store["wt_amd_trails"] = None
        store["wt_amd_key"] = key
    return store["wt_amd_tunnel"]


def build_lbm_component(coords_after, airfoil_name: str = "", *, nx: int = 1024, ny: int = 512,
                        dtype: str = "float32", frames: Optional[int] = None, frames_per_update: int = 4, scale: int = 1,
                        run_every: float = 0.05, st=None) -> Optional[WindTunnel]:
    """Render the interactive LBM wind tunnel for the user's parsed coordinates (AA.py:20-42) and stream it live —
    without an end unless `frames` bounds the stream (see the module text)."""
    if st is None:
        import streamlit as st          # noqa: PLC0415  (lazy: absent in the build image)
    try:
        wt = _tunnel_for(st, coords_after, airfoil_name, nx, ny, dtype)
    except Exception as exc:            # the page shows a message instead of raising (AA.py:25-32, html:887-897)
        st.error(f"⚠️ LBM wind tunnel unavailable: {exc}")
        return None

    def controls():
        """The component's own controls (html:20-58) -> the tunnel; Streamlit evaluates them once per (fragment) run."""
        aoa = st.slider("Angle of attack", -20.0, 25.0, 6.0, 0.5)                   # html:26
        field = _FIELD_LABELS[st.selectbox("Field", list(_FIELD_LABELS))]          # html:32-36
        u0 = st.slider("Flow speed", 30, 100, 60, 2) / 1000.0                       # html:41, 957
        ntrails = st.slider("Trails", 800, 5000, 2600, 100)                         # html:47
        if aoa != wt.aoa_deg:
            wt.aoa_deg = aoa
        wt.set_field(field)
        wt.set_flow_speed(u0)
        tracers = st.session_state.get("wt_amd_tracers")
        if tracers is None:
            tracers = st.session_state["wt_amd_tracers"] = Tracers(wt, n=ntrails)
        elif tracers.x.size != ntrails:
            tracers.resize(ntrails)
        layer = st.session_state.get("wt_amd_trails")
        if layer is None or layer.s != scale:
            layer = st.session_state["wt_amd_trails"] = wt.trail_layer(scale)      # on the GPU for a whole-lattice tunnel
        canvas = st.empty()                                                          # the <canvas> of the component
        slots = [c.empty() for c in st.columns(4)]
        return tracers, layer, canvas, slots

    def advance_and_show(tracers, layer, canvas, slots):
        """`frames_per_update` passes of frame() (html:902-930), then one redraw of the canvas and the read-outs."""
        for _ in range(max(1, int(frames_per_update))):
            wt.frame(render=False)                                                   # 4 steps, ranges, forces every 3rd frame
            tracers.draw(layer, 16.0)                                                # stepParticles(dt), html:917
        st.session_state["wt_amd_frames"] = st.session_state.get("wt_amd_frames", 0) + max(1, int(frames_per_update))
        canvas.image(wt.compose_frame(trails=layer, scale=scale), caption="D2Q9 lattice-Boltzmann · MI355X · live unsteady solve",
                     **_full_width_kwarg(canvas.image))
        s = wt.stats()                                                               # updateStatsUI, html:862-885
        slots[0].metric("CL (approx)", "—" if s.cl is None else f"{s.cl:.3f}")
        slots[1].metric("CD (approx)", "—" if s.cd is None else f"{s.cd:.3f}")
        slots[2].metric("Reynolds", f"{round(s.reynolds):,}")
        slots[3].metric("Separation", s.separation)

    if frames is not None:                                                           # bounded stream (batch use, tests)
        state = controls()
        done = 0
        while done < int(frames):
            advance_and_show(*state)
            done += max(1, int(frames_per_update))
    elif hasattr(st, "fragment"):                                                    # Streamlit >= 1.33: re-run by Streamlit's own timer, for ever
        @st.fragment(run_every=run_every)
        def live():
            advance_and_show(*controls())
        live()
    else:                                                                            # the rAF loop itself: ends only when Streamlit stops the script
        state = controls()
        while True:
            advance_and_show(*state)
    return wt

"""Streamlit wiring: the drop-in for ``build_lbm_component`` of the reference's analysis page.

``pages/Airfoil_Analysis.py:20-42`` injects ``coords_after`` into the WebGL template and embeds it
(call sites AA.py:1210-1213, 1413-1416, captions AA.py:1397-1432).  This module keeps that
function's name and inputs and renders the MI355X tunnel instead: the component's own controls
(html:20-58: angle of attack, field, flow speed, trails), the read-outs of html:53-57 (CL, CD,
Reynolds, separation) and the field image; one rerun of the script = a batch of frames.

``streamlit`` is imported lazily (it is not installed in the build image); everything else is the
package's ordinary host API, so a page needs only::

    from airfoil_cfd_tool_amd.streamlit_page import build_lbm_component
    build_lbm_component(result["coords_after"], airfoil_name)
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .tracers import Tracers
from .windtunnel import WindTunnel

_FIELD_LABELS = {"Velocity": "speed", "Pressure (Cp)": "cp", "Vorticity": "vort"}     # html:32-36


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
        store["wt_amd_key"] = key
    return store["wt_amd_tunnel"]


def build_lbm_component(coords_after, airfoil_name: str = "", *, nx: int = 1024, ny: int = 512,
                        dtype: str = "float32", frames_per_rerun: int = 15, st=None) -> Optional[WindTunnel]:
    """Render the interactive LBM wind tunnel for the user's parsed coordinates (AA.py:20-42)."""
    if st is None:
        import streamlit as st          # noqa: PLC0415  (lazy: absent in the build image)
    try:
        wt = _tunnel_for(st, coords_after, airfoil_name, nx, ny, dtype)
    except Exception as exc:            # the page shows a message instead of raising (AA.py:25-32, html:887-897)
        st.error(f"⚠️ LBM wind tunnel unavailable: {exc}")
        return None

    aoa = st.slider("Angle of attack", -20.0, 25.0, 6.0, 0.5)                       # html:26
    field = _FIELD_LABELS[st.selectbox("Field", list(_FIELD_LABELS))]              # html:32-36
    u0 = st.slider("Flow speed", 30, 100, 60, 2) / 1000.0                           # html:41, 957
    ntrails = st.slider("Trails", 800, 5000, 2600, 100)                             # html:47
    if aoa != wt.aoa_deg:
        wt.aoa_deg = aoa
    wt.set_field(field)
    wt.set_flow_speed(u0)

    tracers = st.session_state.get("wt_amd_tracers")
    if tracers is None:
        tracers = st.session_state["wt_amd_tracers"] = Tracers(wt, n=ntrails)
    elif tracers.x.size != ntrails:
        tracers.resize(ntrails)

    segments = []
    for _ in range(int(frames_per_rerun)):
        wt.frame(render=False)                                                       # html:902-915
        segments.append(tracers.step(16.0))                                          # html:917
    image = wt.render_rgba()[::-1]                                                   # top row first for display

    st.image(np.ascontiguousarray(image), caption="D2Q9 lattice-Boltzmann · MI355X · live unsteady solve",
             use_column_width=True)
    s = wt.stats()
    c1, c2, c3, c4 = st.columns(4)
    c1.metric("CL (approx)", "—" if s.cl is None else f"{s.cl:.3f}")                 # html:863
    c2.metric("CD (approx)", "—" if s.cd is None else f"{s.cd:.3f}")                 # html:864
    c3.metric("Reynolds", f"{round(s.reynolds):,}")                                   # html:865-866
    c4.metric("Separation", s.separation)                                            # html:869-884
    st.session_state["wt_amd_segments"] = segments[-1]
    return wt

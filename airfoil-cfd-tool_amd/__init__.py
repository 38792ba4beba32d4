"""MI355X-native D2Q9 lattice-Boltzmann airfoil wind tunnel (hot path of
583phoenix-hue/Airfoil-CFD-Tool's ``pages/airfoil_flow_lbm_aerolab.html``).

Import as ``airfoil_cfd_tool_amd`` (the directory name carries a hyphen; the
sibling ``airfoil_cfd_tool_amd/`` package forwards here).
"""
from . import geometry, datfile, compose  # noqa: F401
from .datfile import DatParseError, load_dat, parse_dat_file, detect_and_merge_sections  # noqa: F401
from ._capi import Engine, WTError, load_library, LIB_PATH  # noqa: F401
from .windtunnel import (WindTunnel, build_lbm_component, Stats, stall_label, tau_from_reynolds,  # noqa: F401
                         reynolds, chord_cells, write_png, FIELD_MODES, TAU_DEFAULT, U0_DEFAULT, VORT_SCALE, STEPS_PER_FRAME)

from .tracers import Tracers  # noqa: F401
from .distributed import SlabWindTunnel, LocalSlabWindTunnel, slab_bounds, slab_edges, balanced_edges, balance_split, balance_over_group, measure_slab_cost, measure_slab_real  # noqa: F401

__all__ = ["WindTunnel", "build_lbm_component", "Engine", "WTError", "geometry", "load_library"]

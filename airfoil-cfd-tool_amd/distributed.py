"""Column-slab wind tunnel over ``torch.distributed`` — one process per GPU.

New design (the reference is a single browser GL context): the lattice is
split into column slabs, one per rank.  The data path — ghost-column refresh by RCCL
send/recv over xGMI, overlapped with the interior columns — lives INSIDE libwindtunnel
(``wt_create_slab`` / ``wt_comm_init_rank`` / ``wt_step``); ``torch.distributed`` is plumbing:
it carries the RCCL unique id, combines the P partial reductions and gathers fields for
read-back.  The engine is injected so that the host logic of this module (slab bounds,
reduction combination, gathers, the page's "keep previous range" and smoothing rules) is
exercised by world_size-2 ``gloo`` tests on CPU with a stand-in engine (tests/).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import geometry as geo
from .windtunnel import (FIELD_MODES, STEPS_PER_FRAME, TAU_DEFAULT, U0_DEFAULT, VORT_SCALE, Stats, chord_cells,
                         reynolds, stall_label, tau_from_reynolds)


def slab_bounds(nx: int, nranks: int) -> List[Tuple[int, int]]:
    """(x0, width) of every slab — the same split wt_create_slab makes: [r*NX/P, (r+1)*NX/P)."""
    edges = [r * nx // nranks for r in range(nranks + 1)]
    return [(edges[r], edges[r + 1] - edges[r]) for r in range(nranks)]


def _default_engine_factory(nx, ny, dtype, device, rank, nranks, halo):
    from ._capi import Engine
    return Engine(nx, ny, dtype=dtype, device=device, rank=rank, nranks=nranks, halo=halo)


class SlabWindTunnel:
    """The :class:`WindTunnel` surface for a lattice sharded over the ranks of a process group."""

    def __init__(self, coords=None, name: str = "", *, shape: str = "naca2412", nx: int = 4096, ny: int = 2048,
                 dtype="float32", aoa_deg: float = 6.0, u0: float = U0_DEFAULT, tau: Optional[float] = None,
                 re: Optional[float] = None, field: str = "speed", halo: int = 16, device: Optional[int] = None,
                 group=None, engine_factory: Callable = _default_engine_factory, y_half: Optional[float] = None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("SlabWindTunnel needs an initialised torch.distributed process group")
        if tau is not None and re is not None:
            raise ValueError("give tau or re, not both")
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.nx, self.ny = int(nx), int(ny)
        self.name = name or "Uploaded airfoil"
        self.user_coords = geo.round_coords(coords) if coords is not None and len(coords) else []
        self.shape, self.field, self.y_half = shape, field, y_half
        self.u0 = float(u0)
        self.tau = float(tau) if tau is not None else (tau_from_reynolds(re, self.u0, self.nx) if re is not None else TAU_DEFAULT)
        self.bounds = slab_bounds(self.nx, self.nranks)
        self.x0, self.width = self.bounds[self.rank]
        self.halo = int(halo) if self.nranks > 1 else 0
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.engine = engine_factory(self.nx, self.ny, dtype, device, self.rank, self.nranks, self.halo)
        if (self.engine.x0, self.engine.width) != (self.x0, self.width):
            raise RuntimeError("engine and host disagree on the slab bounds")
        self.dtype = np.dtype(dtype)
        if self.nranks > 1:
            ids = [self.engine.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ids, src=self._global_rank(0), group=group)
            self.engine.comm_init_rank(ids[0])
        self.max_s, self.cp_min, self.cp_max = 0.6, -1.0, 1.0          # html:593
        self.cl_smooth: Optional[float] = None
        self.cd_smooth: Optional[float] = None
        self.sep_frac = 0.0
        self.stat_counter = 0
        self.steps = 0
        self.geometry: Optional[geo.Geometry] = None
        self.init_sim(self.u0)
        self.apply_geometry(aoa_deg)

    # ---- plumbing ----------------------------------------------------------------------
    def _global_rank(self, group_rank: int) -> int:
        return group_rank if self.group is None else self._dist.get_global_rank(self.group, group_rank)

    def _tensor_device(self):
        import torch
        return torch.device("cuda") if self._dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _all_reduce(self, values, op):
        import torch
        t = torch.tensor(values, dtype=torch.float64, device=self._tensor_device())
        self._dist.all_reduce(t, op=op, group=self.group)
        return [float(v) for v in t.cpu()]

    # ---- the component's runtime (same names as WindTunnel) -------------------------------
    def init_sim(self, u0: float) -> None:
        self.engine.init_equilibrium(u0)
        self.steps = 0

    def apply_geometry(self, aoa_deg: float, shape: Optional[str] = None) -> None:
        if shape is not None:
            self.shape = shape
        # every rank rasterises the whole mask (O(NY x 160), deterministic) and hands it to its
        # slab, which keeps the owned + ghost columns (wt_set_mask)
        self.geometry = geo.build_geometry(self.nx, self.ny, aoa_deg, self.user_coords, self.shape, self.y_half)
        self.engine.set_mask(self.geometry.mask)

    @property
    def aoa_deg(self) -> float:
        return self.geometry.a_deg

    @aoa_deg.setter
    def aoa_deg(self, value: float) -> None:
        self.apply_geometry(value)

    def set_flow_speed(self, u0: float) -> None:
        self.u0 = float(u0)

    def set_field(self, field: str) -> None:
        if field not in FIELD_MODES:
            raise ValueError(f"field must be one of {sorted(FIELD_MODES)}")
        self.field = field

    def sim_step(self, n: int = 1) -> None:
        self.engine.step(n, self.tau, self.u0)
        self.steps += n

    def update_fields_from_macro(self):
        """html:596-614 over all slabs: max / min / max of the per-slab partials."""
        mx, cmin, cmax = self.engine.reduce_ranges(self.u0)
        ReduceOp = self._dist.ReduceOp
        mx, cmax = self._all_reduce([mx, cmax], ReduceOp.MAX)
        (cmin,) = self._all_reduce([cmin], ReduceOp.MIN)
        if mx > 0:
            self.max_s = mx
        if math.isfinite(cmin):
            self.cp_min = cmin
        if math.isfinite(cmax):
            self.cp_max = cmax
        return self.max_s, self.cp_min, self.cp_max

    def compute_forces(self):
        """html:650-700 over all slabs: sums of the per-slab face sums."""
        fx, fy, surf, rev = self.engine.forces()
        fx, fy, surf, rev = self._all_reduce([fx, fy, float(surf), float(rev)], self._dist.ReduceOp.SUM)
        surf, rev = int(round(surf)), int(round(rev))
        if surf == 0:
            return None
        q = 0.5 * self.u0 * self.u0 * chord_cells(self.nx)
        cl_raw, cd_raw = fy / q, fx / q
        self.cl_smooth = cl_raw if self.cl_smooth is None else self.cl_smooth * 0.9 + cl_raw * 0.1
        self.cd_smooth = cd_raw if self.cd_smooth is None else self.cd_smooth * 0.9 + cd_raw * 0.1
        self.sep_frac = self.sep_frac * 0.85 + (rev / surf) * 0.15
        return cl_raw, cd_raw, rev / surf

    def _gather_columns(self, local: np.ndarray, dst: int = 0) -> Optional[np.ndarray]:
        """Gathers [..., NY, width_r] slabs into [..., NY, NX] on group rank `dst`."""
        import torch
        dev = self._tensor_device()
        mine = torch.from_numpy(np.ascontiguousarray(np.moveaxis(local, -1, 0))).to(dev)   # column-major: [width, ..., NY]
        if self.rank == dst:
            bufs = [torch.empty((w,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=dev) for _, w in self.bounds]
            self._dist.gather(mine, bufs, dst=self._global_rank(dst), group=self.group)
            full = torch.cat(bufs, dim=0).cpu().numpy()
            return np.ascontiguousarray(np.moveaxis(full, 0, -1))
        self._dist.gather(mine, None, dst=self._global_rank(dst), group=self.group)
        return None

    def read_macro(self, dst: int = 0):
        """readMacro (html:547-552) of the whole tunnel on rank `dst` (None elsewhere)."""
        rho, ux, uy = self.engine.read_macro()
        full = self._gather_columns(np.stack([rho, ux, uy]), dst)
        return None if full is None else (full[0], full[1], full[2])

    def read_f(self, dst: int = 0):
        return self._gather_columns(self.engine.read_f(), dst)

    def render_field(self, field: Optional[str] = None, dst: int = 0):
        mode = FIELD_MODES[field or self.field]
        t = self.engine.field(mode, self.u0, self.max_s, self.cp_min, self.cp_max, VORT_SCALE)
        return self._gather_columns(t, dst)

    def frame(self, render: bool = False):
        self.sim_step(STEPS_PER_FRAME)
        t = self.render_field() if render else None
        self.update_fields_from_macro()
        self.stat_counter += 1
        if self.stat_counter % 3 == 0:
            self.compute_forces()
        return t

    def stats(self) -> Stats:
        return Stats(cl=self.cl_smooth, cd=None if self.cd_smooth is None else max(self.cd_smooth, 0.0),
                     reynolds=reynolds(self.u0, self.nx, self.tau), sep_frac=self.sep_frac,
                     separation=stall_label(self.sep_frac))

    def close(self) -> None:
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

"""Column-slab wind tunnel over ``torch.distributed`` — one process per GPU.

New design (the reference is a single browser GL context): the lattice is
split into column slabs, one per rank.  The data path — ghost-column refresh by RCCL
send/recv over xGMI, overlapped with the interior columns — lives INSIDE libwindtunnel
(``wt_create_slab`` / ``wt_comm_init_rank`` / ``wt_step``); ``torch.distributed`` is plumbing:
it carries the RCCL unique id, combines the P partial reductions and gathers fields for
read-back.  :class:`SlabWindTunnel` is :class:`WindTunnel` with those three hooks replaced, so
the page logic (sliders, frame loop, smoothing, stall label) is literally the same code.  The
engine is injected so that this host logic is exercised by world_size-2 ``gloo`` tests on CPU
with a stand-in engine (tests/_slab_standin.py).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import numpy as np

from .windtunnel import FIELD_MODES, VORT_SCALE, WindTunnel


def slab_edges(nx: int, nranks: int) -> List[int]:
    """Equal widths — the split wt_create_slab makes: slab r owns [r*NX/P, (r+1)*NX/P)."""
    return [r * nx // nranks for r in range(nranks + 1)]


def slab_bounds(nx: int, nranks: int, edges: Optional[List[int]] = None) -> List[Tuple[int, int]]:
    """(x0, width) of every slab of a split (equal widths unless `edges` is given)."""
    edges = slab_edges(nx, nranks) if edges is None else [int(e) for e in edges]
    return [(edges[r], edges[r + 1] - edges[r]) for r in range(nranks)]


DEFAULT_HALO = 61       # ghost columns per interior side: one refresh step + fifteen four-step passes between two exchanges.  29 until round 5: with the
                        # slabs on overlapping windows the measured AND the exchange-priced cost fall with the depth up to 61-77 (profiles/r05_z_halo_deep.txt)


def default_halo(bounds: List[Tuple[int, int]], halo: Optional[int] = None) -> int:
    """The hosts' halo depth: `halo` as given; None = DEFAULT_HALO clamped to the narrowest slab of the split (a halo cannot be deeper than
    the columns the neighbour owns — wt_create_slab refuses that, and an EXPLICIT value that is too deep still fails there, loudly)."""
    if halo is not None:
        return int(halo)
    return max(1, min(DEFAULT_HALO, min(w for _, w in bounds)))


def balanced_edges(edges: List[int], cost: List[float], min_width: int) -> List[int]:
    """One round of cutting the slabs by cost instead of by width.  `cost[r]` is what slab r of the split `edges` was measured to take per
    step (any unit): the cost is spread evenly over the slab's columns, and the new edges cut the running cost into equal parts.
    Slabs that hold the body come out narrower, plain ones wider; no slab gets fewer than `min_width` columns (>= the halo depth).
    A slab's cost is not linear in its columns (fixed costs per pass, body columns inside), so the caller repeats: measure the new split,
    cut again (bench.py --balance; two or three rounds settle within a few percent)."""
    P = len(edges) - 1
    if P < 2:
        return list(edges)
    dens = [float(cost[r]) / max(1, edges[r + 1] - edges[r]) for r in range(P)]
    total = sum(dens[r] * (edges[r + 1] - edges[r]) for r in range(P))
    new = [edges[0]]
    r, acc = 0, 0.0                       # acc: cost of the columns left of edges[r]
    for k in range(1, P):
        want = total * k / P
        while r < P - 1 and acc + dens[r] * (edges[r + 1] - edges[r]) < want:
            acc += dens[r] * (edges[r + 1] - edges[r])
            r += 1
        x = edges[r] + (want - acc) / dens[r] if dens[r] > 0 else edges[r + 1]
        new.append(int(round(x)))
    new.append(edges[-1])
    for k in range(1, P):                 # minimum widths, left to right, then right to left
        new[k] = max(new[k], new[k - 1] + min_width)
    for k in range(P - 1, 0, -1):
        new[k] = min(new[k], new[k + 1] - min_width)
    if any(new[k + 1] - new[k] < min_width for k in range(P)):
        raise ValueError(f"{P} slabs of at least {min_width} columns do not fit {edges[-1] - edges[0]} columns")
    return new


def measure_slab_cost(mask: np.ndarray, edges: List[int], rank: int, halo: int, dtype="float32", device: int = 0, tau: float = 0.58,
                      u0: float = 0.06, steps: int = 200, options=None, trimmed: bool = True) -> float:
    """Microseconds per step of slab `rank` of the split on its own: a stand-alone handle of the slab's owned + ghost columns on the mask
    columns it would hold (its cut edges act as inlet / outlet, which costs a little more than the ghost columns of the real slab).
    No neighbour and no communicator is involved, so the ranks of a process group measure their candidate slabs side by side.
    trimmed: the real slab marches, pass by pass, only the ghost columns that stay exact (library option trim_ghosts) — on average half of
    them — so the stand-in carries (halo + 1) // 2 ghost columns per interior side instead of all `halo`."""
    from ._capi import Engine
    nx = mask.shape[1]
    ghosts = (halo + 1) // 2 if trimmed else halo
    lo, hi = max(0, edges[rank] - ghosts), min(nx, edges[rank + 1] + ghosts)
    with Engine(hi - lo, mask.shape[0], dtype=dtype, device=device) as e:
        for k, v in (options or {}).items():
            e.set_option(k, v)
        # every slab of a tunnel takes the steps per pass its NARROWEST slab allows: plan this stand-in the same way
        e.set_option("plan_columns", min(b - a for a, b in zip(edges[:-1], edges[1:])) + halo)
        e.set_mask(np.ascontiguousarray(mask[:, lo:hi]))
        e.init_equilibrium(u0)
        e.step(200, tau, u0)                  # (the plan is timed and cut again here; and an idle GPU's clocks take some 20 ms of work to settle)
        return e.step_timed(steps, tau, u0) / steps * 1e3


def measure_slab_real(mask: np.ndarray, edges: List[int], rank: int, halo: int, dtype="float32", device: int = 0, tau: float = 0.58,
                      u0: float = 0.06, steps: int = 0, options=None) -> dict:
    """Slab `rank` of the split as a REAL slab handle, alone on the GPU and linked to ITSELF (wt_link_local with one handle: its ghost columns are
    refreshed from its own owned edges — a tunnel periodic in x over this slab): the whole slab state machine runs — trimmed ghost marching, the
    refresh mode of `options`, the exchange beside the interior — with a copy kernel as the exchange.  Returns microseconds per step and, per
    refresh, the time of the exchange (a copy here) and of the compute that runs beside it ("interior_us": the window an exchange over the links
    can hide in).  What measure_slab_cost approximates with a stand-alone lattice, measured on the thing itself; the exchange over xGMI is NOT in
    it (bench.py adds its stated model: exchange_model_us)."""
    from ._capi import Engine
    ny, nx = mask.shape
    with Engine(nx, ny, dtype=dtype, device=device, rank=rank, nranks=len(edges) - 1, halo=halo, edges=list(edges)) as e:
        for k, v in (options or {}).items():
            e.set_option(k, v)
        Engine.link_local([e])
        e.set_mask(mask)
        e.init_equilibrium(u0)
        cycle = max(halo, 1)
        n = steps or 14 * cycle
        Engine.step_group([e], 7 * cycle, tau, u0)          # (the plan is timed and cut again here; the clocks settle)
        e.set_option("exchange_timing", 1)
        ms = Engine.step_group_timed([e], n, tau, u0)[0]
        nex = max(1.0, e.get_option("exchanges"))
        return {"us_per_step": ms / n * 1e3, "steps": n, "exchanges": int(e.get_option("exchanges")),
                "exchange_us": e.get_option("exchange_ms") / nex * 1e3, "interior_us": e.get_option("interior_ms") / nex * 1e3,
                "exposed_us": e.get_option("exchange_exposed_ms") / nex * 1e3, "single_steps": int(e.get_option("single_steps")),
                "fused_renewals": int(e.get_option("fused_renewals")), "passes": int(e.get_option("passes")), "width": e.width,
                "sides": (1 if rank > 0 else 0) + (1 if rank < len(edges) - 2 else 0)}


def balance_split(nx: int, nranks: int, min_width: int, measure: Callable[[List[int]], List[float]], rounds: int = 3):
    """Equal widths first, then `rounds` times: measure every slab of the split (`measure(edges)` -> cost per slab, the same list on every
    caller), cut by cost (balanced_edges).  Returns (edges of the split whose SLOWEST slab was fastest, history of (edges, costs))."""
    edges = slab_edges(nx, nranks)
    history = []
    for _ in range(max(0, rounds) + 1):
        cost = [float(c) for c in measure(edges)]
        history.append((list(edges), cost))
        if len(history) > rounds:
            break
        nxt = balanced_edges(edges, cost, min_width)
        if nxt == edges:
            break
        edges = nxt
    best = min(history, key=lambda h: max(h[1]))
    return best[0], history


def balance_over_group(nx: int, min_width: int, measure_mine: Callable[[List[int], int], float], rounds: int = 3, group=None):
    """balance_split over a torch.distributed process group — COLLECTIVE: every rank calls it with the same arguments.  `measure_mine(edges,
    rank)` times this rank's slab of a candidate split (measure_slab_cost on its own GPU: no communicator involved); the P costs are
    all-gathered, so every rank cuts the same edges.  A rank whose measurement raises reports that, and then EVERY rank falls back to equal
    widths: returns (None, []) everywhere, never a split only some ranks know."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    class _Failed(RuntimeError):
        pass

    def measure(edges):
        ok, mine = 1.0, 0.0
        try:
            mine = float(measure_mine(edges, rank))
        except Exception as e:      # noqa: BLE001 - whatever it is, the other ranks must learn of it
            print(f"[balance rank {rank}] slab measurement failed: {e}", flush=True)
            ok = 0.0
        t = torch.tensor([mine, ok], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t, group=group)
        if min(float(g[1]) for g in got) < 1.0:
            raise _Failed("a rank could not measure its slab")
        return [float(g[0]) for g in got]

    try:
        return balance_split(nx, world, min_width, measure, rounds)
    except (_Failed, ValueError):       # (ValueError: the slabs do not fit — computed from gathered numbers, the same on every rank)
        return None, []


def _default_engine_factory(nx, ny, dtype, device, rank, nranks, halo, edges=None):
    from ._capi import Engine
    return Engine(nx, ny, dtype=dtype, device=device, rank=rank, nranks=nranks, halo=halo, edges=edges)


class SlabWindTunnel(WindTunnel):
    """The :class:`WindTunnel` surface for a lattice sharded over the ranks of a process group.
    Every method is collective: all ranks call it with the same arguments."""

    def __init__(self, coords=None, name: str = "", *, halo: Optional[int] = None, device: Optional[int] = None, group=None,
                 engine_factory: Callable = _default_engine_factory, nx: int = 4096, ny: int = 2048, edges: Optional[List[int]] = None,
                 **kwargs):
        """edges: the split (nranks + 1 rising column indices, the same on every rank; see balanced_edges), None for equal widths."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("SlabWindTunnel needs an initialised torch.distributed process group")
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.edges = None if edges is None else [int(e) for e in edges]
        self.bounds = slab_bounds(int(nx), self.nranks, self.edges)
        self.x0, self.width = self.bounds[self.rank]
        self.halo = default_halo(self.bounds, halo) if self.nranks > 1 else 0
        self._engine_factory = engine_factory
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = int(device)
        if dist.get_backend(group) == "nccl":
            # collectives of this class run on THIS rank's GPU whatever thread calls them and whether or not the
            # engine (whose hipSetDevice would otherwise be what selects the device) exists yet
            import torch
            torch.cuda.set_device(self.device)
        super().__init__(coords, name, nx=nx, ny=ny, device=device, **kwargs)

    # ---- hooks -------------------------------------------------------------------------
    def _make_engine(self, dtype, device):
        if self.edges is None:
            eng = self._engine_factory(self.nx, self.ny, dtype, device, self.rank, self.nranks, self.halo)
        else:
            eng = self._engine_factory(self.nx, self.ny, dtype, device, self.rank, self.nranks, self.halo, edges=self.edges)
        if (eng.x0, eng.width) != (self.x0, self.width):
            raise RuntimeError("engine and host disagree on the slab bounds")
        if self.nranks > 1:
            ids = [eng.comm_unique_id() if self.rank == 0 else None]
            self._dist.broadcast_object_list(ids, src=self._global_rank(0), group=self.group)
            eng.comm_init_rank(ids[0])
        return eng

    def _reduce_ranges(self):
        """html:596-614 over all slabs: max / min / max of the per-slab partials."""
        mx, cmin, cmax = self.engine.reduce_ranges(self.u0)
        ReduceOp = self._dist.ReduceOp
        mx, cmax = self._all_reduce([mx, cmax], ReduceOp.MAX)
        (cmin,) = self._all_reduce([cmin], ReduceOp.MIN)
        return mx, cmin, cmax

    def _forces(self):
        """html:650-698 over all slabs: sums of the per-slab face sums."""
        fx, fy, surf, rev = self.engine.forces()
        fx, fy, surf, rev = self._all_reduce([fx, fy, float(surf), float(rev)], self._dist.ReduceOp.SUM)
        return fx, fy, int(round(surf)), int(round(rev))

    def clamp_events(self):
        a, b = self.engine.clamp_events()
        a, b = self._all_reduce([float(a), float(b)], self._dist.ReduceOp.SUM)
        return int(round(a)), int(round(b))

    # ---- plumbing ----------------------------------------------------------------------
    def _global_rank(self, group_rank: int) -> int:
        return group_rank if self.group is None else self._dist.get_global_rank(self.group, group_rank)

    def _tensor_device(self):
        import torch
        return torch.device("cuda", self.device) if self._dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _all_reduce(self, values, op):
        import torch
        t = torch.tensor(values, dtype=torch.float64, device=self._tensor_device())
        self._dist.all_reduce(t, op=op, group=self.group)
        return [float(v) for v in t.cpu()]

    def _gather_columns(self, local: np.ndarray, dst: int = 0) -> Optional[np.ndarray]:
        """Gathers [..., NY, width_r] slabs into [..., NY, NX] on group rank `dst` (None elsewhere)."""
        import torch
        dev = self._tensor_device()
        mine = torch.from_numpy(np.ascontiguousarray(np.moveaxis(local, -1, 0))).to(dev)   # [width, ..., NY]
        wmax = max(w for _, w in self.bounds)
        if mine.shape[0] < wmax:                 # gather wants one size on every rank: slabs of unequal width are padded, then trimmed
            pad = torch.zeros((wmax - mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=dev)
            mine = torch.cat([mine, pad], dim=0)
        if self.rank == dst:
            bufs = [torch.empty_like(mine) for _ in self.bounds]
            self._dist.gather(mine, bufs, dst=self._global_rank(dst), group=self.group)
            full = torch.cat([b[:w] for b, (_, w) in zip(bufs, self.bounds)], dim=0).cpu().numpy()
            return np.ascontiguousarray(np.moveaxis(full, 0, -1))
        self._dist.gather(mine, None, dst=self._global_rank(dst), group=self.group)
        return None

    # ---- read-backs: whole tunnel on rank `dst` -------------------------------------------
    def read_macro(self, dst: int = 0):
        full = self._gather_columns(np.stack(self.engine.read_macro()), dst)
        self.macro = None if full is None else (full[0], full[1], full[2])
        return self.macro

    def read_f(self, dst: int = 0):
        return self._gather_columns(self.engine.read_f(), dst)

    def write_f(self, f: np.ndarray) -> None:
        self.engine.write_f(np.ascontiguousarray(f[:, :, self.x0:self.x0 + self.width]))

    def render_field(self, max_s=None, cp_min=None, cp_max=None, field: Optional[str] = None, dst: int = 0):
        mode = FIELD_MODES[field or self.field]
        t = self.engine.field(mode, self.u0, self.max_s if max_s is None else max_s,
                              self.cp_min if cp_min is None else cp_min, self.cp_max if cp_max is None else cp_max, VORT_SCALE)
        return self._gather_columns(t, dst)

    def render_rgba(self, field: Optional[str] = None, dst: int = 0):
        mode = FIELD_MODES[field or self.field]
        img = self.engine.render_rgba(mode, self.u0, self.max_s, self.cp_min, self.cp_max, VORT_SCALE)   # [NY][W][4]
        full = self._gather_columns(np.moveaxis(img, 2, 0), dst)                                          # [4][NY][NX]
        return None if full is None else np.ascontiguousarray(np.moveaxis(full, 0, 2))

    def frame(self, render: bool = False):
        return super().frame(render=render)


class _LocalSlabEngine:
    """Engine-shaped facade over P locally linked slab handles (one process, any mix of devices):
    ghost columns move by peer copies (wt_link_local), all slabs advance in lock-step (wt_step_group)."""

    def __init__(self, nx, ny, dtype, devices, halo, edges=None):
        from ._capi import Engine
        self._Engine = Engine
        P = len(devices)
        self.slabs = [Engine(nx, ny, dtype=dtype, device=d, rank=r, nranks=P, halo=halo, edges=edges) for r, d in enumerate(devices)]
        Engine.link_local(self.slabs)
        self.dtype = self.slabs[0].dtype
        self.nx_global, self.ny, self.x0, self.width = nx, ny, 0, nx

    def close(self):
        for s in self.slabs:
            s.close()

    def set_mask(self, mask):
        for s in self.slabs:
            s.set_mask(mask)

    def init_equilibrium(self, u0):
        for s in self.slabs:
            s.init_equilibrium(u0)

    def step(self, nsteps, tau, u0):
        self._Engine.step_group(self.slabs, nsteps, tau, u0)

    def read_f(self):
        return np.concatenate([s.read_f() for s in self.slabs], axis=2)

    def write_f(self, f):
        for s in self.slabs:
            s.write_f(np.ascontiguousarray(f[:, :, s.x0:s.x0 + s.width]))

    def read_macro(self):
        parts = [s.read_macro() for s in self.slabs]
        return tuple(np.concatenate([p[a] for p in parts], axis=1) for a in range(3))

    def reduce_ranges(self, u0):
        rr = [s.reduce_ranges(u0) for s in self.slabs]
        return max(r[0] for r in rr), min(r[1] for r in rr), max(r[2] for r in rr)

    def forces(self):
        ff = [s.forces() for s in self.slabs]
        return tuple(sum(x[a] for x in ff) for a in range(4))

    def field(self, *args):
        return np.concatenate([s.field(*args) for s in self.slabs], axis=1)

    def render_rgba(self, *args):
        return np.concatenate([s.render_rgba(*args) for s in self.slabs], axis=1)

    def advect_tracers(self, *args):
        raise NotImplementedError("tracers need the whole lattice on one handle")

    def sync(self):
        for s in self.slabs:
            s.sync()


class LocalSlabWindTunnel(WindTunnel):
    """:class:`WindTunnel` over several GPUs driven by ONE process (no torch.distributed, no RCCL):
    ``LocalSlabWindTunnel(coords, devices=[0, 1, 2, 3], halo=61, nx=8192, ny=4096)``."""

    def __init__(self, coords=None, name: str = "", *, devices=(0,), halo: Optional[int] = None, edges=None, **kwargs):
        self.devices = [int(d) for d in devices]
        self.edges = None if edges is None else [int(e) for e in edges]
        self.halo = default_halo(slab_bounds(int(kwargs.get("nx", 320)), max(1, len(self.devices)), self.edges), halo)
        if len(self.devices) < 2:
            raise ValueError("LocalSlabWindTunnel needs at least two slabs; use WindTunnel for one GPU")
        super().__init__(coords, name, device=self.devices[0], **kwargs)

    def _make_engine(self, dtype, device):
        return _LocalSlabEngine(self.nx, self.ny, dtype, self.devices, self.halo, self.edges)

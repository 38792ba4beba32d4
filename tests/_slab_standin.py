"""Stand-in slab engine for the CPU (gloo) tests of airfoil_cfd_tool_amd.distributed.

TEST CODE.  Implements the Engine interface the host logic uses, with the ORACLE as the per-slab
stepper and torch.distributed send/recv as the ghost-column transport, following the same protocol
as libwindtunnel's slab handles (csrc/windtunnel.hip: `halo` ghost columns per interior side, exact
replicas of the neighbour's edge columns, refreshed when exhausted, i.e. every `halo` steps).
"""
import numpy as np
import torch
import torch.distributed as dist

import lbm_numpy as oracle


class OracleSlabEngine:
    def __init__(self, nx, ny, dtype, device, rank, nranks, halo, edges=None):
        self.nx_global, self.ny, self.dtype = nx, ny, np.dtype(dtype)
        self.rank, self.nranks, self.halo = rank, nranks, halo if nranks > 1 else 0
        edges = [r * nx // nranks for r in range(nranks + 1)] if edges is None else list(edges)      # wt_create_slab / wt_create_slab_at
        self.x0, self.width = edges[rank], edges[rank + 1] - edges[rank]
        self.gl = self.halo if rank > 0 else 0
        self.gr = self.halo if rank < nranks - 1 else 0
        self.lo, self.hi = self.x0 - self.gl, self.x0 + self.width + self.gr      # global columns held locally
        self.ghost_valid = 0
        self.f = self.macro = self.mask = None

    # -- transport bootstrap (nothing to do on gloo) --
    @staticmethod
    def comm_unique_id():
        return b"\0" * 128

    def comm_init_rank(self, comm_id):
        assert len(comm_id) == 128

    def close(self):
        pass

    def set_mask(self, mask):
        assert mask.shape == (self.ny, self.nx_global)
        self.mask = np.ascontiguousarray(mask[:, self.lo:self.hi])

    def init_equilibrium(self, u0):
        self.f, self.macro = oracle.equilibrium_init(self.hi - self.lo, self.ny, u0, self.dtype)
        self.ghost_valid = self.halo

    def _exchange(self):
        ops, bufs = [], []
        own = slice(self.gl, self.gl + self.width)
        if self.gl:
            send = torch.from_numpy(np.ascontiguousarray(self.f[:, :, self.gl:self.gl + self.halo]))
            recv = torch.empty_like(send)
            ops += [dist.P2POp(dist.isend, send, self.rank - 1), dist.P2POp(dist.irecv, recv, self.rank - 1)]
            bufs.append((slice(0, self.gl), recv))
        if self.gr:
            send = torch.from_numpy(np.ascontiguousarray(self.f[:, :, own.stop - self.halo:own.stop]))
            recv = torch.empty_like(send)
            ops += [dist.P2POp(dist.isend, send, self.rank + 1), dist.P2POp(dist.irecv, recv, self.rank + 1)]
            bufs.append((slice(own.stop, own.stop + self.gr), recv))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for sl, t in bufs:
            self.f[:, :, sl] = t.numpy()
        self.ghost_valid = self.halo

    def step(self, nsteps, tau, u0):
        for _ in range(nsteps):
            if self.nranks > 1 and self.ghost_valid == 0:
                self._exchange()
            # the oracle treats the array's first/last column as inlet/outlet: right for the tunnel's
            # own ends; on a ghost edge it merely spoils the outermost ghost, which expires anyway
            self.f, self.macro = oracle.step(self.f, self.mask, tau, u0)
            self.ghost_valid -= 1

    def _own(self, a):
        return np.ascontiguousarray(a[..., self.gl:self.gl + self.width])

    def read_f(self):
        return self._own(self.f)

    def read_macro(self):
        return tuple(self._own(a) for a in self.macro)

    def reduce_ranges(self, u0):
        rho, ux, uy = self.read_macro()
        fluid = self._own(self.mask) == 0
        if not fluid.any():
            return 0.0, float("inf"), float("-inf")
        u = ux.astype(np.float64)[fluid] / u0
        v = uy.astype(np.float64)[fluid] / u0
        cp = (rho.astype(np.float64)[fluid] - 1) / (1.5 * u0 * u0)
        s = np.hypot(u, v)
        s = s[s < 4]
        q = cp[(cp > -4) & (cp < 1.2)]
        return (float(s.max()) if s.size else 0.0, float(q.min()) if q.size else float("inf"),
                float(q.max()) if q.size else float("-inf"))

    def forces(self):
        """Faces attributed to the slab that owns the FLUID cell (as wt_forces does)."""
        sol = self.mask != 0
        rho = self.macro[0].astype(np.float64)
        ux = self.macro[1]
        ny, nxl = sol.shape
        fx = fy = 0.0
        surf = rev = 0
        for dx, dy in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            for y in range(ny):
                for x in range(self.gl, self.gl + self.width):
                    if sol[y, x]:
                        continue
                    xs, ys = x + dx, y + dy
                    gxs = self.lo + xs
                    if gxs < 0 or gxs >= self.nx_global or ys < 0 or ys >= ny or not sol[ys, xs]:
                        continue
                    p = rho[y, x] / 3
                    fx += p * dx
                    fy += p * dy
                    surf += 1
                    rev += int(ux[y, x] < 0)
        return fx, fy, surf, rev

    def field(self, mode, u0, max_s, cp_min, cp_max, vort_scale):
        if mode == 2 and self.nranks > 1:
            raise NotImplementedError("stand-in: vorticity needs the macro ghost exchange")
        t = oracle.field_scalar(mode, *self.macro, self.mask, u0, max_s, cp_min, cp_max, vort_scale)
        return self._own(t)

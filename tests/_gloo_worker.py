"""One rank of the world_size-N gloo test (spawned by tests/test_distributed_gloo.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, halo = (int(v) for v in sys.argv[1:5])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lbm_numpy as oracle
    from _slab_standin import OracleSlabEngine
    from airfoil_cfd_tool_amd.distributed import SlabWindTunnel, slab_bounds

    nx, ny = 96, 48
    wt = SlabWindTunnel(shape="naca4412", nx=nx, ny=ny, aoa_deg=9.0, halo=halo, engine_factory=OracleSlabEngine)
    assert wt.bounds == slab_bounds(nx, world) and sum(w for _, w in wt.bounds) == nx
    frames = 7
    for _ in range(frames):
        wt.frame()
    wt.aoa_deg = 15.0                      # AoA slider on every rank
    wt.set_flow_speed(0.07)
    wt.sim_step(5)
    wt.update_fields_from_macro()
    wt.compute_forces()
    macro = wt.read_macro()
    f = wt.read_f()
    t_speed = wt.render_field(field="speed")
    ok = True
    if rank == 0:
        import airfoil_cfd_tool_amd.geometry as geo
        m1 = geo.build_geometry(nx, ny, 9.0, None, "naca4412").mask
        m2 = geo.build_geometry(nx, ny, 15.0, None, "naca4412").mask
        fr = None
        prev = (0.6, -1.0, 1.0)
        st = oracle.ForceState()
        for k in range(1, frames + 1):
            fr, mac = oracle.run(m1, 4, 0.58, 0.06, np.float32, f=fr)
            prev = oracle.ranges_from_macro(*mac, m1, 0.06, prev)
            if k % 3 == 0:
                st.update(*oracle.compute_forces_raw(mac[0], mac[1], m1), 0.06, nx)
        fr, mac = oracle.run(m2, 5, 0.58, 0.07, np.float32, f=fr)
        prev = oracle.ranges_from_macro(*mac, m2, 0.07, prev)
        st.update(*oracle.compute_forces_raw(mac[0], mac[1], m2), 0.07, nx)
        ok &= np.array_equal(f, fr) and all(np.array_equal(a, b) for a, b in zip(macro, mac))
        ok &= np.allclose([wt.max_s, wt.cp_min, wt.cp_max], prev, rtol=1e-13, atol=0)
        ok &= np.allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], [st.cl, st.cd, st.sep], rtol=1e-11, atol=1e-13)
        ref_t = oracle.field_scalar(0, *mac, m2, 0.07, *prev)
        ok &= np.array_equal(np.nan_to_num(t_speed), np.nan_to_num(ref_t))
        ok &= wt.stats().separation == oracle.stall_label(st.sep)
        print("rank0 checks", "PASS" if ok else "FAIL", flush=True)
    else:
        ok &= macro is None and f is None and t_speed is None
    # every rank must hold the same combined reductions
    import torch
    t = torch.tensor([wt.max_s, wt.cp_min, wt.cp_max, wt.cl_smooth, wt.cd_smooth, wt.sep_frac], dtype=torch.float64)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ok &= bool(torch.equal(lo, hi))

    # ---- slabs cut by cost: the collective balancing gives every rank the same split, a failing rank sends ALL ranks back to equal widths,
    #      and a tunnel on an uneven split equals the monolithic oracle
    from airfoil_cfd_tool_amd.distributed import balance_over_group, slab_edges
    dens = np.ones(4096); dens[1000:3000] = 2.5
    edges, hist = balance_over_group(4096, 32, lambda ed, r: 5.0 + 0.02 * dens[ed[r]:ed[r + 1]].sum(), rounds=3)
    every = [None] * world
    dist.all_gather_object(every, (edges, [h[1] for h in hist]))
    ok &= all(e == every[0] for e in every) and edges is not None and edges[0] == 0 and edges[-1] == 4096 and len(edges) == world + 1
    ok &= min(max(c) for _, c in hist) <= max(hist[0][1]) and hist[0][0] == slab_edges(4096, world)
    calls = {"n": 0}

    def flaky(ed, r):
        calls["n"] += 1
        if r == world - 1 and calls["n"] == 2:
            raise RuntimeError("no GPU here")
        return 1.0 + r
    e2, h2 = balance_over_group(4096, 32, flaky, rounds=3)
    ok &= e2 is None and h2 == []
    uneven = [0] + [int(nx * (0.2 + 0.5 * k / max(1, world - 1))) for k in range(world - 1)] + [nx]
    wt2 = SlabWindTunnel(shape="naca4412", nx=nx, ny=ny, aoa_deg=9.0, halo=halo, engine_factory=OracleSlabEngine, edges=uneven)
    ok &= [b[0] for b in wt2.bounds] == uneven[:-1]
    wt2.sim_step(2 * halo + 3)
    f2 = wt2.read_f()
    if rank == 0:
        import airfoil_cfd_tool_amd.geometry as geo
        fr2, _ = oracle.run(geo.build_geometry(nx, ny, 9.0, None, "naca4412").mask, 2 * halo + 3, 0.58, 0.06, np.float32)
        same = bool(np.array_equal(f2, fr2))
        print("rank0 uneven split", "PASS" if same else "FAIL", flush=True)
        ok &= same
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

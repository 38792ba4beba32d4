"""One rank of tests/test_gpu_rccl_stub.py::test_slab_wind_tunnel_host_class_over_the_stand_in_transport (launched through torch.distributed.run with
the RCCL stand-in LD_PRELOADed): the host class SlabWindTunnel on REAL slab engines, every rank on device 0 — the page's frame loop, the AoA slider,
the combined reductions, read-backs gathered to rank 0 and the vorticity field, whose ghost columns move over the transport (refresh_macro_ghosts,
TR_RCCL branch) — against one WindTunnel on the whole lattice."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import airfoil_cfd_tool_amd as pkg
    nx, ny, dtype = 1536, 768, sys.argv[1] if len(sys.argv) > 1 else "float32"
    kw = dict(shape="naca4412", nx=nx, ny=ny, aoa_deg=9.0, dtype=dtype)
    wt = pkg.SlabWindTunnel(device=0, **kw)                       # default halo (29, clamped to the narrowest slab), equal widths
    assert wt.halo == min(pkg.distributed.DEFAULT_HALO, nx // world) and wt.engine.get_option("comm_ranks") == world
    frames = 9
    for _ in range(frames):
        wt.frame()
    wt.aoa_deg = 14.0                                             # the AoA slider, on every rank (wt_set_mask is collective on slab handles)
    wt.set_flow_speed(0.07)
    wt.sim_step(33)
    wt.update_fields_from_macro()
    wt.compute_forces()
    macro = wt.read_macro()
    f = wt.read_f()
    fields = {m: wt.render_field(field=m) for m in ("speed", "cp", "vort")}
    rgba = wt.render_rgba(field="vort")
    clamp = wt.clamp_events()
    ok = True
    if rank == 0:
        with pkg.WindTunnel(device=0, **kw) as ref:
            for _ in range(frames):
                ref.frame()
            ref.aoa_deg = 14.0
            ref.set_flow_speed(0.07)
            ref.sim_step(33)
            ref.update_fields_from_macro()
            ref.compute_forces()
            same = lambda a, b: np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
            ok &= same(f, ref.read_f()) and all(same(a, b) for a, b in zip(macro, ref.read_macro()))
            ok &= np.allclose([wt.max_s, wt.cp_min, wt.cp_max], [ref.max_s, ref.cp_min, ref.cp_max], rtol=1e-13, atol=0)
            ok &= np.allclose([wt.cl_smooth, wt.cd_smooth, wt.sep_frac], [ref.cl_smooth, ref.cd_smooth, ref.sep_frac], rtol=1e-10, atol=1e-13)
            for m, t in fields.items():
                ok &= same(np.nan_to_num(t), np.nan_to_num(ref.render_field(field=m)))
            ok &= same(rgba, ref.render_rgba(field="vort")) and clamp == ref.clamp_events()
            ok &= wt.stats().separation == ref.stats().separation
        print(f"slab wind tunnel over the stand-in transport, {world} ranks, {dtype}: {'PASS' if ok else 'FAIL'}", flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    wt.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

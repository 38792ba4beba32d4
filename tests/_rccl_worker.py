"""One rank of the multi-GPU RCCL test (launched by tests/test_gpu_rccl_multi.py through torch.distributed.run, one process
per GPU): the column-slab tunnel over the library's own RCCL transport against the single lattice, bit for bit."""
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
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    same = os.environ.get("WT_RCCL_SAME_DEVICE") is not None      # experiment: every rank on ONE GPU (torch's group over gloo; only the library talks RCCL)
    if same:
        local = int(os.environ["WT_RCCL_SAME_DEVICE"])
    torch.cuda.set_device(local)
    if same:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import airfoil_cfd_tool_amd as pkg
    nx, ny, halo = 2048, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 16
    chunks = [1, 2, 3, 40, 17]
    mask = pkg.geometry.build_geometry(nx, ny, 9.0, None, "naca4412").mask
    ok = True
    uneven = [0] + [int(nx * (0.30 + 0.40 * k / (world - 1))) for k in range(world - 1)] + [nx]     # slabs cut by the caller (wt_create_slab_at)
    # (refresh 2: the ghost columns renewed inside a fused pass — the exchange beside its interior columns, over the transport under test)
    for dtype, depth, edges, refresh in (("float32", 0, None, 0), ("float32", 2, None, 0), ("float64", 0, None, 0), ("float32", 0, uneven, 0),
                                         ("float32", 0, None, 2), ("float64", 0, uneven, 2)):
        eng = pkg.Engine(nx, ny, dtype=dtype, device=local, rank=rank, nranks=world, halo=halo, edges=edges)
        if depth:
            eng.set_option("fuse_depth", depth)
        eng.set_option("fuse_steps", 2)
        eng.set_option("refresh", refresh)
        ids = [pkg.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        eng.comm_init_rank(ids[0])                          # (all-reduces the schedule fingerprint: a rank that would plan differently fails here)
        assert int(eng.get_option("comm_ranks")) == world
        eng.set_mask(mask); eng.init_equilibrium(0.06)
        # before the first exchange: every rank answers wt_plan_steps alike (the property the collective exchange rests on)
        plans = [None] * world
        dist.all_gather_object(plans, (eng.get_option("fuse_active"), eng.get_option("fuse_depth"), tuple(eng.plan_steps(sum(chunks), 0.58))))
        if len(set(plans)) != 1:
            print(f"rank {rank}: the ranks plan different schedules: {plans}", flush=True)
            ok = False
            eng.close()
            break
        for n in chunks:
            eng.step(n, 0.58, 0.06)
        if refresh == 2 and eng.get_option("fuse_depth") >= 3 and (eng.get_option("fused_renewals") < 1 or eng.get_option("boundary_exchanges") != 0):
            print(f"rank {rank}: refresh 2 took no fused renewal", flush=True)
            ok = False
        f = eng.read_f()                                    # this rank's owned columns
        rho, ux, uy = eng.read_macro()
        gathered = [None] * world
        dist.all_gather_object(gathered, (f, rho, ux, uy))
        eng.close()
        if rank == 0:
            with pkg.Engine(nx, ny, dtype=dtype, device=local) as ref:
                ref.set_option("fuse_steps", 0)
                ref.set_mask(mask); ref.init_equilibrium(0.06)
                for n in chunks:
                    ref.step(n, 0.58, 0.06)
                fr, (r0, u0, v0) = ref.read_f(), ref.read_macro()
            fa = np.concatenate([g[0] for g in gathered], axis=2)
            same = (np.array_equal(fa.view(np.uint8), fr.view(np.uint8))
                    and all(np.array_equal(np.concatenate([g[i] for g in gathered], axis=1).view(np.uint8), m.view(np.uint8))
                            for i, m in ((1, r0), (2, u0), (3, v0))))
            print(f"rccl slabs {dtype} depth={depth or 'auto'} world={world} edges={edges or 'equal'} refresh={refresh}: {'PASS' if same else 'FAIL'}", flush=True)
            ok &= bool(same)
    # a rank with another knob value is refused by EVERY rank inside wt_comm_init_rank — an error, not a hang at the first exchange
    eng = pkg.Engine(nx, ny, device=local, rank=rank, nranks=world, halo=halo)
    eng.set_option("fuse_depth", 3 if rank == world - 1 else 4)
    ids = [pkg.Engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    refused = False
    try:
        eng.comm_init_rank(ids[0])
    except pkg.WTError as e:
        refused = e.code == -5 and "disagree" in str(e) and "fuse_depth" in str(e)
    # ... and the refused handle is not left steppable: its communicator is gone, a stepping call says "no transport" instead of entering an exchange
    if refused:
        eng.set_mask(mask); eng.init_equilibrium(0.06)
        try:
            eng.step(2 * halo + 4, 0.58, 0.06)
            refused = False
        except pkg.WTError as e:
            refused = e.code == -5 and "no transport" in str(e) and int(eng.get_option("comm_ranks")) == 0
    eng.close()
    got = [None] * world
    dist.all_gather_object(got, refused)
    if rank == 0:
        print(f"rccl slabs: a rank with another fuse_depth is refused on every rank: {'PASS' if all(got) else 'FAIL'}", flush=True)
    ok &= all(got)
    flag = torch.tensor([1 if ok else 0], device=torch.device("cpu") if same else torch.device("cuda", local))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

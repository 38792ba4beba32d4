#!/usr/bin/env python3
"""bench.py — MLUPS of the D2Q9 wind-tunnel step on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one lattice update (simStep, html:510-525) of the whole tunnel.  Workload:
BASELINE.json configs[2] — 4096x4096 fp32, AoA 10 deg, U0 0.06, tau 0.58.  The S1223 coordinates
named there are not available offline (no network; the reference ships no .dat files), so the
body is the reference's own high-camber built-in shape NACA 6409 (html:127); the kernel's cost
depends on the body only through the fraction of wave-tiles that touch its surface.

N > 1: the SAME 4096x4096 lattice is split into N column slabs (strong scaling, as the metric
"MLUPS on 4096^2 ... at 1/2/4/8 MI355X" is quoted), one process per GPU, ghost columns
exchanged by RCCL send/recv inside libwindtunnel every `halo` steps.

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` and
`cpu_baseline` objects.  The CPU baseline is the straight NumPy transcription of the scheme
(oracle/lbm_numpy.py) timed on this host on a bounded sample — it is only ever the thing
measured BESIDE the product, never part of it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s float4-copy)
BYTES_PER_LUP = {"float32": 72, "float64": 144}   # 9 loads + 9 stores per site update


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--ny", type=int, default=4096)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--shape", default="naca6409")
    ap.add_argument("--aoa", type=float, default=10.0)
    ap.add_argument("--u0", type=float, default=0.06)
    ap.add_argument("--tau", type=float, default=0.58)
    ap.add_argument("--halo", type=int, default=16, help="ghost columns per interior slab side (exchange every `halo` steps)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="steps of the NumPy CPU baseline (0 = skip)")
    ap.add_argument("--fuse", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="two steps per pass over the lattice (csrc/step_fused.hpp; fp32; bit-identical): -1 library "
                         "default (on where it pays for one GPU, off for slabs), 0 off, 1 where it pays, 2 always")
    ap.add_argument("--fuse-chunk", type=int, default=0, help="columns per marching chunk (0 = chosen per mask)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: nx x ny split over N GPUs; weak: every GPU gets an nx x ny slab")
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(mask, steps, tau, u0, dtype):
    """NumPy transcription (the 'port') on the host cores; element-wise NumPy = 1 core."""
    import numpy as np
    import lbm_numpy
    ny, nx = mask.shape
    f, _ = lbm_numpy.equilibrium_init(nx, ny, u0, np.dtype(dtype))
    f, _ = lbm_numpy.step(f, mask, tau, u0)            # untimed first touch
    t0 = time.perf_counter()
    for _ in range(steps):
        f, _ = lbm_numpy.step(f, mask, tau, u0)
    dt = time.perf_counter() - t0
    return {
        "value": nx * ny * steps / dt / 1e6,
        "unit": "MLUPS",
        "cores": 1,
        "kind": "port",
        "sample": f"{steps} steps of the same {nx}x{ny} {dtype} lattice and mask, oracle/lbm_numpy.py "
                  f"(NumPy {np.__version__}, element-wise => single core; host: {cpu_model()}, {os.cpu_count()} cpus), {dt:.1f} s",
    }


def baseline_metric():
    """BASELINE.json's metric string, verbatim."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as fh:
            return json.load(fh)["metric"]
    except Exception:
        return "MLUPS on 4096\u00b2 fp32 D2Q9 at 1/2/4/8 MI355X; achieved % HBM3E peak"


def measured_traffic(workload_key):
    """HBM bytes per launch from the rocprofv3 PMC passes kept under profiles/ (None if absent)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import airfoil_cfd_tool_amd as wtpkg

    if os.environ.get("WT_BENCH_FORCE_DEVICE") is not None:      # plumbing tests on a 1-GPU box only
        local_rank = int(os.environ["WT_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    nx_total = args.nx if (args.scaling == "strong" or world == 1) else args.nx * world
    ny = args.ny
    geom = wtpkg.geometry.build_geometry(nx_total, ny, args.aoa, None, args.shape)
    mask = geom.mask

    if distributed:
        eng = wtpkg.Engine(nx_total, ny, dtype=args.dtype, device=local_rank, rank=rank, nranks=world, halo=args.halo)
        if args.fuse >= 0 and args.dtype == "float32":
            eng.set_option("fuse_chunk", args.fuse_chunk)
            eng.set_option("fuse_steps", args.fuse)
        ids = [wtpkg.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        eng.comm_init_rank(ids[0])
    else:
        eng = wtpkg.Engine(nx_total, ny, dtype=args.dtype, device=local_rank)
        if args.fuse >= 0 and args.dtype == "float32":
            eng.set_option("fuse_chunk", args.fuse_chunk)
            eng.set_option("fuse_steps", args.fuse)
    eng.set_mask(mask)
    eng.init_equilibrium(args.u0)

    def barrier():
        if distributed:
            dist.barrier()

    # warm-up (untimed)
    if args.warmup > 0:
        eng.step(args.warmup, args.tau, args.u0)
    eng.sync()
    torch.cuda.synchronize()
    barrier()

    # timed region: exactly K steps; HIP events on the library's compute stream give the device time
    t0 = time.perf_counter()
    dev_ms = eng.step_timed(args.steps, args.tau, args.u0)
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0

    if distributed:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    sites = nx_total * ny
    mlups = sites * args.steps / wall / 1e6
    bpl = BYTES_PER_LUP[args.dtype]
    fused = bool(eng.get_option("fuse_active"))
    launch_ms = dev_ms / args.steps                      # one step = one launch of k_step over the slab
    if fused:
        launch_ms *= 2.0                                 # one pass (k_step2 + the two list passes) = TWO steps
    sites_per_launch = eng.width * ny if not distributed else (nx_total // world) * ny
    achieved = bpl * sites_per_launch * (2 if fused else 1) / (launch_ms * 1e-3) / 1e9
    workload = (f"{args.shape.upper()} {nx_total}x{ny} {args.dtype} D2Q9, AoA={args.aoa:g} deg, U0={args.u0:g}, "
                f"tau={args.tau:g} (BASELINE configs[2]; S1223 coordinates unavailable offline -> NACA 6409)")
    out = {
        "metric": baseline_metric(),
        "value": mlups,
        "unit": "MLUPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32" if args.dtype == "float32" else "f64",
        "data": "synthetic",
        "config": {"workload": workload, "nx": nx_total, "ny": ny, "slabs": world,
                   "halo": args.halo if distributed else 0, "fuse_steps": int(fused),
                   "fuse_chunk": int(eng.get_option("fuse_chunk")) if fused else 0,
                   "solid_sites": int((mask != 0).sum())},
        "roofline": {
            "bound": "hbm",
            "kernel": "wt::k_step2 (+ k_step_list x2 on the body zone), two steps per pass" if fused else "wt::k_step",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": measured_traffic(f"{nx_total}x{ny}_{args.dtype}" + ("_fused" if fused else "")) if not distributed else None,
            "algorithmic_bytes_per_launch": bpl * sites_per_launch * (2 if fused else 1),
            "launch_ms": launch_ms,
        },
    }
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(mask, args.cpu_steps, args.tau, args.u0, args.dtype)
    elif rank == 0:
        out["cpu_baseline"] = None
    eng.close()
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

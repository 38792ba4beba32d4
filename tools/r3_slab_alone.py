"""Runs ON THE GPU BOX: one INTERIOR slab handle of an 8-way split (no inlet / outlet column: every body-free block can chain), stepped
alone for as long as its ghost columns stay exact (16 steps per init); chain blocks on / off, steps per pass 3 / 4."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airfoil_cfd_tool_amd as pkg

def run(nx, ny, nranks, rank, depth, chain, halo=16, reps=30):
    with pkg.Engine(nx, ny, rank=rank, nranks=nranks, halo=halo) as e:
        e.set_option("chain", chain)
        e.set_option("fuse_depth", depth)
        e.set_mask(np.zeros((ny, nx), np.uint8))
        n = (halo // depth) * depth
        ts = []
        for r in range(reps):
            e.init_equilibrium(0.06)
            ts.append(e.step_timed(n, 0.58, 0.06) / n * 1e3)
        ts = sorted(ts[5:])
        return ts[len(ts) // 2], int(e.get_option("fuse_units")), int(e.get_option("chain_units")), e.info().width

for nranks in (8, 4):
    for depth in (3, 4):
        for chain in (0, 1, 0, 1):
            us, units, cu, w = run(4096, 4096, nranks, nranks // 2, depth, chain)
            print(f"interior slab of {nranks}: width {w}+32, depth {depth}, chain {chain}: {units} units ({cu} in chain blocks), {us:.2f} us/step", flush=True)

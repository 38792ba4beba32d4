"""Developer experiment: where does a slab of the 2-way split of 4096^2 lose against the whole lattice?  Stand-alone lattices of its columns against the real slab.
    python3 tools/r5_n2_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import airfoil_cfd_tool_amd as pkg
full = pkg.geometry.build_geometry(4096, 4096, 10.0, None, "naca6409").mask
def alone(mask, opts=None):
    best = 1e9
    for rep in range(2):
        with pkg.Engine(mask.shape[1], mask.shape[0]) as e:
            for k, v in (opts or {}).items():
                e.set_option(k, v)
            e.set_mask(mask); e.init_equilibrium(0.06); e.step(600, 0.58, 0.06); e.sync()
            best = min(best, min(e.step_timed(408, 0.58, 0.06) for _ in range(3)) / 408 * 1e3)
            info = (int(e.get_option("fuse_depth")), int(e.get_option("fuse_units")), int(e.get_option("window_overlap")))
    return best, info
for name, m in (("whole 4096 columns", full), ("columns 0..2109 (slab 0 of 2 + its ghosts)", np.ascontiguousarray(full[:, :2109])), ("columns 0..2048", np.ascontiguousarray(full[:, :2048])),
                ("2048 plain columns", np.zeros((4096, 2048), np.uint8))):
    us, info = alone(m)
    print(f"stand-alone {name}: {us:.2f} us/step = {us / m.shape[1] * 1e3:.2f} ns per column  plan {info}", flush=True)
for halo in (61, 29):
    r = pkg.measure_slab_real(full, [0, 2048, 4096], 0, halo, options={"refresh": 0})
    print(f"real slab 0 of 2, halo {halo}: {r['us_per_step']:.2f} us/step; interior beside the exchange {r['interior_us']:.1f} us; passes {r['passes']}, single steps {r['single_steps']}, exchanges {r['exchanges']} in {r['steps']} steps", flush=True)

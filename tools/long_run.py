#!/usr/bin/env python3
"""Long run of the benchmark workload: the state evolves (particles clump, contact chains
get deeper), so the solver's iteration and round counts and the step time are printed as it goes.
usage: long_run.py [steps] [workload] [library]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from psim_amd.particle_system import ParticleSystem

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
label, n, bounds, state, mask = bench.workload(sys.argv[2] if len(sys.argv) > 2 else "c2")
ps = ParticleSystem(n, dict(bench.SIM), np.random.default_rng(42), bounds, stage_mask=mask, state=state,
                    library=sys.argv[3] if len(sys.argv) > 3 else None)
ps._calculate_gravity()
e0 = ps.get_energies()
t0 = time.time()
for k in range(1, steps + 1):
    ps.update()
    if k % 50 == 0 or k == steps:
        c = ps.counts()
        dt = (time.time() - t0) / 50 * 1e3
        t0 = time.time()
        print("step %4d: %.2f ms/step  particles %d contacts %d iterations %d rounds %d  stages %s" % (
            k, dt, c["particles"], c["contacts"], c["solver_iterations"], c["solver_rounds"],
            {a: round(b, 2) for a, b in ps.stage_times_ms().items() if b > 0.05}), flush=True)
e1 = ps.get_energies()
print("energies (KE, PE, TE) before", e0, "after", e1)
assert np.isfinite(e1).all() and np.isfinite(ps.positions).all()
print("ok")

#!/usr/bin/env python3
"""Robustness run at sizes far above the benchmark's: N particles at the reference's
default density, a few full steps, stage times and counts printed (no oracle at this
size; the invariants checked are the ones that do not need one)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd")):
    sys.path.insert(0, p)
import numpy as np
from bench import SIM
from psim_amd.particle_system import ParticleSystem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dense_hot = len(sys.argv) > 3 and sys.argv[3] == "c4"
t0 = time.time()
if dense_hot:
    # the dense cold-collapse stress case: a uniform disk at 4x the default density (64 Mi
    # particles: R = 73078 in 150000 x 150000), at rest, temperatures U[0.5, 0.9] so that
    # collision heating pushes particles past the explosion threshold
    import bench
    label, n, bounds, state, mask = bench.workload("c4", n)
    sim = dict(SIM)
    ps = ParticleSystem(n, sim, np.random.default_rng(42), bounds, state=state)
else:
    side = float(np.sqrt(n / 1e-3))            # 1e-3 particles per square pixel, as config.json
    bounds = (side * np.sqrt(2.0), side / np.sqrt(2.0))
    sim = dict(SIM)
    if len(sys.argv) > 3 and sys.argv[3] == "hot":
        # a tenth of the particles start at or above the explosion threshold (1.0): mass
        # explosions, compaction and the event log at scale
        sim.update(min_temp=0.8, max_temp=1.022)
    ps = ParticleSystem(n, sim, np.random.default_rng(42), bounds)
print("init %.1f s" % (time.time() - t0), flush=True)
m0 = float(ps.masses.sum())
ps._calculate_gravity()
e0 = ps.get_energies()
for k in range(steps):
    t = time.time()
    ps.update()
    c = ps.counts()
    print("step %d: %.1f ms  %s" % (k, (time.time() - t) * 1e3, c), flush=True)
    print("   stages", {a: round(b, 2) for a, b in ps.stage_times_ms().items()}, flush=True)
    assert c["tree_nodes"] == 1 + 4 * c["internal_nodes"]
    n_now = c["particles"]
e1 = ps.get_energies()
print("energies before", e0, "after", e1)
assert np.isfinite(e1).all()
assert abs(float(ps.masses.sum()) - m0) <= 1e-9 * m0 or c["particles"] < n
assert ps.num_particles == n_now and ps.positions.shape == (n_now, 2)
acc = ps.accelerations
assert np.isfinite(acc).all() and np.isfinite(ps.positions).all()
print("ok")

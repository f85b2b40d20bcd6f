#!/usr/bin/env python3
"""One step of a uniform cloud at a multiple of the default density (default config: 2000
particles in 2000 x 1000), timed: where the parallel contact solver stops coping.
usage: dense_blob.py PARTICLES [WIDTH HEIGHT]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from psim_amd.particle_system import ParticleSystem

n = int(sys.argv[1])
bounds = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (2000.0, 1000.0)
s = ParticleSystem(n, dict(bench.SIM, particle_count=n), np.random.default_rng(0), bounds)
s._calculate_gravity()
for k in range(3):
    t = time.time()
    s.update()
    _ = s.positions
    c = s.counts()
    print("n=%d step %d: %.3f s  contacts %d iterations %d launches of rounds %d" % (
        n, k, time.time() - t, c["contacts"], c["solver_iterations"], c["solver_rounds"]), flush=True)

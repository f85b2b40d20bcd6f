#!/usr/bin/env python3
"""Random small states of every density against the oracle (host-loop build by default, the HIP
library with --hip), full physics, a few steps each; and the same states split over emulated
ranks against the unsplit step.  usage: fuzz_contacts.py [first_seed] [count] [--hip]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import parity
from psim_amd import _capi
from psim_amd.particle_system import ParticleSystem
from psim_amd.distributed import ShardedParticleSystem

args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(args[0]) if args else 0
count = int(args[1]) if len(args) > 1 else 20
lib = _capi.HIP_LIBRARY if "--hip" in sys.argv else os.environ.get("PSIM_FUZZ_LIBRARY", os.path.join(ROOT, "tests", "hostloop", "libpsim_hostloop.so"))
BASE = dict(particle_count=0, min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10,
            barnes_hut_theta=0.5, min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1,
            coefficient_of_restitution=0.5, grid_cell_size_multiplier=10.0, position_correction_factor=1.05,
            collision_solver_iterations=1, explosion_efficiency=0.5, thermal_damping_factor=0.0005)
for seed in range(first, first + count):
    r = np.random.default_rng(1000 + seed)
    n = int(r.integers(300, 6000))
    area_per = float(np.exp(r.uniform(np.log(25.0), np.log(2500.0))))     # px^2 per particle: packed .. sparse
    w = float(np.sqrt(n * area_per * 1.5))
    bounds = (int(w) + 50, int(w / 1.5) + 50)
    sim = dict(BASE, collision_solver_iterations=int(r.integers(1, 4)),
               grid_cell_size_multiplier=float(r.choice([2.0, 4.0, 10.0])),
               position_correction_factor=float(r.choice([0.0, 0.5, 1.05])))
    if r.random() < 0.4:
        sim.update(min_temp=0.7, max_temp=1.02)
    steps = 4
    parity.run_against_oracle(n, sim, seed, bounds, steps, lib)
    # the same state split over 3 ranks (one process plays them) against the unsplit step
    a = ParticleSystem(n, dict(sim), np.random.default_rng(seed), bounds, library=lib)
    fall = 0
    b = [ShardedParticleSystem(n, dict(sim), np.random.default_rng(seed), bounds, library=lib, emulate=(3, k)) for k in (0, 2)]
    for s in [a] + b:
        s._calculate_gravity()
    for _ in range(steps):
        a.update()
        for s in b:
            s.update()
            assert s.num_particles == a.num_particles
            for f in ("positions", "velocities", "temperatures"):
                assert np.array_equal(getattr(a, f), getattr(s, f)), (seed, f)
            assert s.counts()["contacts"] == a.counts()["contacts"], seed
    fall = b[0].fallbacks
    for s in [a] + b:
        s.close()
    print("seed %d ok: n %d, %.0f px^2 per particle, sweeps %d, cell x%g, correction %g, hot %s, fallbacks %d" % (
        seed, n, area_per, sim["collision_solver_iterations"], sim["grid_cell_size_multiplier"],
        sim["position_correction_factor"], sim["min_temp"] > 0.5, fall), flush=True)
print("all ok")

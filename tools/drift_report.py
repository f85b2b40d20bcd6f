#!/usr/bin/env python3
"""Energy and momentum drift of the GPU step on the reference's default
configuration (config.json verbatim, seed 42, 2000 particles, 100 steps), next to
the energies recorded from the reference itself (tests/golden/default2000.npz)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from conftest import load_golden
from psim_amd.particle_system import ParticleSystem

g = load_golden("default2000")
s = ParticleSystem(g.n, dict(g.sim), g.rng(), g.bounds)
s._calculate_gravity()
ke0, pe0, te0 = s.get_energies()
p0 = s.get_total_momentum()
exact0 = s.exact_potential_energy()
print("default config.json, seed 42, N=2000, bounds 2000x1000, FP64")
print("step      KE            PE(tree)        TE           total        |  reference total   rel.diff")
print("%4d %13.6f %15.6f %13.6f %13.6f  |  %15.6f  %.2e" % (0, ke0, pe0, te0, ke0 + pe0 + te0,
      g["prime_energies"].sum(), abs(ke0 + pe0 + te0 - g["prime_energies"].sum()) / abs(g["prime_energies"].sum())))
for step in range(1, 101):
    s.update()
    if step in (1, 2, 10, 50, 100):
        ke, pe, te = s.get_energies()
        ref = g["energies"][step - 1]
        print("%4d %13.6f %15.6f %13.6f %13.6f  |  %15.6f  %.2e" % (step, ke, pe, te, ke + pe + te, ref.sum(),
              abs(ke + pe + te - ref.sum()) / abs(ref.sum())))
px, py = s.get_total_momentum()
ke, pe, te = s.get_energies()
print("total energy drift over 100 steps: %+.6f (%.3e relative); the reference's own: %+.6f" % (
    ke + pe + te - (ke0 + pe0 + te0), (ke + pe + te - (ke0 + pe0 + te0)) / abs(ke0 + pe0 + te0),
    g["energies"][99].sum() - g["prime_energies"].sum()))
print("total momentum: step 0 (%+.6e, %+.6e) -> step 100 (%+.6e, %+.6e)" % (p0[0], p0[1], px, py))
ref_p = (g["s100_velocities"] * g["init_masses"]).sum(axis=0)
print("reference momentum at step 100: (%+.6e, %+.6e)" % (ref_p[0], ref_p[1]))
print("max |x - x_ref| after 100 steps: %.3e px; max |T - T_ref|: %.3e" % (
    np.abs(s.positions - g["s100_positions"]).max(), np.abs(s.temperatures - g["s100_temperatures"]).max()))
exact1 = s.exact_potential_energy()
print("with the exact softened pairwise potential energy instead of the sibling-cell sum: PE %.6f -> %.6f, "
      "KE + TE + PE_exact %.6f -> %.6f (%+.3e relative)" % (
          exact0, exact1, ke0 + te0 + exact0, ke + te + exact1,
          (ke + te + exact1 - (ke0 + te0 + exact0)) / abs(ke0 + te0 + exact0)))

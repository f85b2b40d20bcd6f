import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "particle-simulator_amd"))
import numpy as np
import bench
from psim_amd.particle_system import ParticleSystem
label, n, bounds, state, mask = bench.workload("c2")
ps = ParticleSystem(n, dict(bench.SIM), np.random.default_rng(42), bounds, device=0, stage_mask=mask, state=state)
ps._calculate_gravity()
for _ in range(3): ps.update()
for w in (0, 8, 7.5, 7, 6.5, 6, 5, 4):
    print(w, round(ps.time_walk(3, w), 3), flush=True)

#!/usr/bin/env python3
"""Lane statistics of the walk's loop on a benchmark tree (GPU): how many trips are uniform
over the group's active lanes, and how many of those the bounding box of the group's 64
targets would settle -- what a test of 64 nodes per instruction against that box could save.
    python tools/walk_profile.py [c2|c1|c3]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd")):
    sys.path.insert(0, p)
import numpy as np

import bench
from psim_amd.particle_system import ParticleSystem

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
label, n, bounds, state, mask = bench.workload(wl)
ps = ParticleSystem(n, dict(bench.SIM), np.random.default_rng(42), bounds, state=state, stage_mask=mask)
ps._calculate_gravity()
ps.update(3)
p = ps.walk_profile()
groups = (n + 63) // 64
T, F = p["trips"], p["force_trips"]
per = {k: v / groups for k, v in p.items()}
A, O = p["box_accept_trips"], p["box_open_trips"]
UA, UO = p["uniform_accept_trips"], p["uniform_open_trips"]
now = 6.0 * T + 11.0 * F
# What a test of 64 elements per instruction against the group's bounding box could save.  The
# loop's 6 "test" instructions are 2 subtractions, a multiply and a fused multiply-add for
# d^2, and two compares; an accepted element needs dx, dy and d^2 for its force all the same,
# so a trip the box settles as "every lane accepts" saves the two compares only, one it
# settles as "every lane opens" saves all six.  Against that: one slot of a 64-wide box test
# per visited element (nearest and farthest corner, two compares: about 14 vector
# instructions per 64 elements, at best), before any cost of keeping a frontier of elements.
def with_box(acc, opn):
    return now - 2.0 * acc - 6.0 * opn + 14.0 * T / 64.0
out = {"workload": label, "groups": groups, "per_group": per,
       "fraction_of_trips": {k: p[k] / T for k in p if k.endswith("_trips")},
       "vector_issues_per_group": {"now (6 per trip + 11 per force trip)": now / groups,
                                   "box test settles what the box can": with_box(A, O) / groups,
                                   "bound: every uniform trip settled by the box": with_box(UA, UO) / groups},
       "saving": {"box": 1.0 - with_box(A, O) / now, "bound": 1.0 - with_box(UA, UO) / now},
       "verdict": "below the 15 % at which a rewrite of the walk around a per-group frontier would be worth building"}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Random states through the HIP library and the oracle: whole tree (topology, masses, centres
of mass bit for bit), accelerations, and one full step each.  Sizes around the tile and
workgroup boundaries of the cooperative kernels, domains of several aspect ratios, uniform /
clumped / tight-pair states.
usage: fuzz_tree.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-simulator_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import parity
from psim_amd import _capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sizes = [1, 2, 3, 5, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, 4096, 8191, 8193, 20000, 65537]
for case in range(cases):
    n = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 30000))
    aspect = float(rng.choice([1.0, 2.0, 0.5, 3.7, 0.21]))
    w = float(rng.uniform(300.0, 4000.0))
    bounds = (w, w / aspect)
    kind = int(rng.integers(0, 4))
    seed = int(rng.integers(1 << 30))
    if kind == 0 or n < 8:
        st = dict(positions=np.random.default_rng(seed).random((n, 2)) * np.array(bounds),
                  masses=np.random.default_rng(seed + 1).uniform(5.0, 20.0, (n, 1)),
                  temperatures=np.full((n, 1), 0.07))
        name = "uniform"
    elif kind == 1:
        # (not denser than a few particles per contact area: a blob in which everything overlaps
        # everything has dependency chains as long as its pair list, which no parallel schedule shortens)
        sigma = max(min(bounds) / float(rng.uniform(4, 40)), 3.0 * float(np.sqrt(n)))
        st = parity.gaussian_cluster_state(n, bounds, sigma, seed, 0.05, 0.1)
        name = "blob"
    elif kind == 2:
        # (the couples' square holds them at no more than about two discs deep, for the same reason)
        pairs = max(1, n // int(rng.integers(4, 60)))
        box = max(float(rng.uniform(5.0, min(bounds) / 2)), float(np.sqrt(30.0 * pairs)))
        while box > min(bounds) / 2:
            pairs = max(1, pairs // 2)
            box = max(5.0, float(np.sqrt(30.0 * pairs)))
        st = parity.tight_pairs_state(n, bounds, seed, pairs=pairs, gap=10.0 ** -float(rng.uniform(1, 5)), box=box)
        name = "pairs"
    else:
        st = parity.clustered_state(n, bounds, seed, clusters=max(1, n // 500), per=int(rng.integers(2, 9)), spread=10.0 ** -float(rng.uniform(1, 4)))
        name = "knots"
    try:
        parity.check_tree_against_oracle(_capi.HIP_LIBRARY, st, n, bounds)
    except RuntimeError as e:
        # (the oracle, like the reference, gives up on more subdivided cells than its pool holds)
        print("case %d n=%d %s %s: oracle/library refused: %s" % (case, n, name, bounds, str(e)[:80]), flush=True)
        continue
    print("case %d n=%d %s bounds=(%.0f, %.0f): ok" % (case, n, name, *bounds), flush=True)
print("done")

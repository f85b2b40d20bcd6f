"""Shared parity checks: a `ParticleSystem` of this build (bound to whichever
library the caller names) against the golden fixtures recorded from the
reference and against the C oracle.

Bars (north star): Morton keys / tree topology / grid / contact pair lists
bit-exact; COM and per-contact energies bit-exact while the state feeding them
is bit-exact; accelerations, positions, velocities, temperatures within the
relative tolerance written at each assert (the reference's gravity, conduction
and PE kernels are `fastmath`, so they have no bit-level definition)."""
import numpy as np

from conftest import load_golden
from psim_amd.particle_system import ParticleSystem

TREE_KEYS_EXACT = ["int_depth", "int_path", "leaf_depth", "leaf_path", "leaf_particle"]
TREE_KEYS_COM = ["int_mass", "int_comx", "int_comy"]

# relative tolerance on accelerations: FP64 walk, summation order differs from the
# reference's recursion (whose own result is only defined up to reassociation)
ACC_RTOL = 2e-12


def make_system(g, library, walk_mode=0, debug=True, **kw):
    s = ParticleSystem(g.n, dict(g.sim), g.rng(), g.bounds, library=library, walk_mode=walk_mode,
                       debug=debug, **kw)
    v = g.init_velocities()
    if v is not None:
        s.velocities = v
    return s


def check_tree(s, g, prefix, com_exact=True):
    ct = s.canonical_tree()
    assert ct["tree_nodes"] == int(g[prefix + "tree_nodes"]), prefix
    for k in TREE_KEYS_EXACT:
        np.testing.assert_array_equal(ct[k], g[prefix + k], err_msg=prefix + k)
    for k in TREE_KEYS_COM:
        if com_exact:
            np.testing.assert_array_equal(ct[k], g[prefix + k], err_msg=prefix + k)
        else:
            np.testing.assert_allclose(ct[k], g[prefix + k], rtol=1e-11, err_msg=prefix + k)


def acc_close(got, want, rtol=ACC_RTOL):
    """relative tolerance per component, plus an absolute floor of 5e-14 of the largest
    acceleration for components whose several hundred terms nearly cancel (their rounding
    error is set by the size of the terms, not of the sum)"""
    scale = np.abs(want).max() if want.size else 1.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(rtol * 1e-3, 5e-14) * scale)


def run_golden_stepwise(name, library, walk_mode=0):
    """Prime + every recorded step of one fixture, checking each stage output."""
    g = load_golden(name)
    s = make_system(g, library, walk_mode)
    np.testing.assert_array_equal(s.positions, g["init_positions"])
    np.testing.assert_array_equal(s.radii, g["init_radii"])
    s._calculate_gravity()
    check_tree(s, g, "init_")
    acc_close(s.accelerations, g["prime_accelerations"])
    np.testing.assert_allclose(s.get_energies(), g["prime_energies"], rtol=1e-11)
    exact = True            # state still bit-identical to the reference's
    for step in range(1, g.steps + 1):
        s.update()
        p = "s%d_" % step
        cand, ncont, nodes, n_after, n_expl = (int(x) for x in g["counts"][step - 1])
        c = s.counts()
        assert c["particles"] == n_after, (step, "n")
        assert c["tree_nodes"] == nodes, (step, "nodes")
        assert c["contacts"] == ncont, (step, "contacts")
        assert c["candidates"] == cand, (step, "candidates")
        assert c["exploded"] == n_expl, (step, "exploded")
        if (p + "grid_offsets") not in g:
            continue
        # grid + tree are built from post-drift positions: exact while the state is
        np.testing.assert_array_equal(s.debug_array("grid_offsets"), g[p + "grid_offsets"])
        np.testing.assert_array_equal(s.debug_array("grid_indices"), g[p + "grid_indices"])
        check_tree(s, g, p, com_exact=False)
        ij = s.debug_array("contact_ij")
        np.testing.assert_array_equal(ij, g[p + "contact_ij"], err_msg="step %d contact pair list" % step)
        np.testing.assert_allclose(s.debug_array("contact_vals"), g[p + "contact_vals"], rtol=1e-7, atol=1e-13)
        np.testing.assert_array_equal(s.debug_array("exploded"), g[p + "exploded"])
        d = np.zeros(len(g[p + "delta_temps"]))
        dd = s.debug_array("delta_temps")
        d[:len(dd)] = dd
        np.testing.assert_allclose(d, g[p + "delta_temps"].ravel(), rtol=1e-9, atol=1e-16)
        np.testing.assert_allclose(s.positions, g[p + "positions"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(s.velocities, g[p + "velocities"], rtol=1e-9, atol=1e-13)
        acc_close(s.accelerations, g[p + "accelerations"], rtol=1e-10)
        np.testing.assert_allclose(s.temperatures, g[p + "temperatures"], rtol=1e-10, atol=1e-15)
        np.testing.assert_allclose(s.get_energies(), g["energies"][step - 1], rtol=1e-9)
        tot = np.array([s.pe_gain_collisions, s.ke_loss_collisions, s.heat_generated_collisions])
        np.testing.assert_allclose(tot, g["collision_totals"][step - 1], rtol=1e-7, atol=1e-12)
    s.close()


def run_against_oracle(n, sim, seed, bounds, steps, library, walk_mode=0, state=None, stage_mask=31,
                       check_contacts=True, oracle_threads=1, oracle_fast_pow=False, system_class=None,
                       prime=True, min_contacts=0, min_exploded=0, log_cap=1 << 22, debug=True):
    """Same seeded inputs through this build and through the C oracle.
    oracle_threads / oracle_fast_pow only make the oracle's gravity loop affordable at
    full benchmark sizes (threads do not change its results; the rsqrt form of x**-1.5
    changes them by a few ulp, far inside the tolerances below).
    system_class: ParticleSystem (default) or its sharded subclass driven through the staged
    entry points.  prime=False: no priming walk, the first drift uses the given
    accelerations (zero) -- both sides then drift to bit-identical positions, so the
    position-only outputs of the first step's contacts (pair list, PE change) are compared
    bit for bit.  min_contacts / min_exploded: the case must actually exercise those paths."""
    from oracle.oracle import OracleSystem
    rng_a, rng_b = np.random.default_rng(seed), np.random.default_rng(seed)
    o = OracleSystem(n, sim, rng_a, bounds, state=state, stage_mask=stage_mask)
    o.set_threads(oracle_threads)
    o.set_fast_pow(oracle_fast_pow)
    s = (system_class or ParticleSystem)(n, dict(sim), rng_b, bounds, library=library, walk_mode=walk_mode, debug=debug,
                                         state=state, stage_mask=stage_mask)
    o.enable_contact_log(log_cap)
    if prime:
        o._calculate_gravity()
        s._calculate_gravity()
        assert s.counts()["tree_nodes"] == o.num_active_nodes
        acc_close(s.accelerations, o.accelerations)
    seen_contacts = seen_exploded = 0
    for step in range(1, steps + 1):
        o.update()
        s.update()
        c = s.counts()
        assert c["particles"] == o.num_particles, step
        assert c["tree_nodes"] == o.num_active_nodes, step
        seen_contacts += c["contacts"]
        seen_exploded += c["exploded"]
        if stage_mask & 1:
            assert c["contacts"] == o.n_contacts, step
            assert c["candidates"] == o.n_candidates, step
            if check_contacts:
                ij, vals = o.contact_log()
                np.testing.assert_array_equal(s.debug_array("contact_ij"), ij, err_msg="step %d" % step)
                if not prime and step == 1:
                    # positions are still bit-identical on both sides: so is everything the
                    # contacts derive from positions alone (particle_system.py:207-237)
                    np.testing.assert_array_equal(s.debug_array("contact_vals")[:, 0], vals[:, 0],
                                                  err_msg="PE change of the first step's contacts")
        np.testing.assert_array_equal(s.debug_array("grid_offsets"), o.grid_offsets)
        np.testing.assert_array_equal(s.debug_array("grid_indices"), o.grid_indices)
        np.testing.assert_array_equal(s.debug_array("exploded"), o.exploded)
        np.testing.assert_allclose(s.positions, o.positions, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(s.velocities, o.velocities, rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(s.temperatures, o.temperatures, rtol=1e-9, atol=1e-15)
    assert seen_contacts >= min_contacts and seen_exploded >= min_exploded, (seen_contacts, seen_exploded)
    e_s = np.array(s.get_energies())
    e_o = np.array([o.get_total_kinetic_energy(), o.get_total_potential_energy(), o.get_total_thermal_energy()])
    np.testing.assert_allclose(e_s, e_o, rtol=1e-8)
    s.close()
    return s, o


def check_explosion_log(name, library, walk_mode=0):
    """The log records of the explosion stage (particle_system.py:683-689, 727-730) against
    the ones the reference's own `particle_sim` logger produced for the same fixture: level,
    text and order, step by step."""
    import json
    import logging
    g = load_golden(name)
    want = [tuple(r) for r in json.loads(bytes(g["log_json"]).decode())]
    got = []
    state = {"step": 0}

    class Capture(logging.Handler):
        def emit(self, record):
            got.append((state["step"], record.levelname, record.getMessage()))

    lg = logging.getLogger("particle_sim")
    handler, old_level, old_prop = Capture(level=logging.DEBUG), lg.level, lg.propagate
    s = make_system(g, library, walk_mode)       # (its constructor logs too: not captured)
    s._calculate_gravity()
    lg.addHandler(handler)
    lg.setLevel(logging.DEBUG)
    lg.propagate = False
    try:
        for step in range(1, g.steps + 1):
            state["step"] = step
            s.update()
    finally:
        lg.removeHandler(handler)
        lg.setLevel(old_level)
        lg.propagate = old_prop
    s.close()
    assert len(want) > 0
    assert got == want


def c4_state(n, seed=42):
    """the dense stress recipe at n particles: a uniform disk at 4x the default density
    (64 Mi particles: R = 73078 in 150000 x 150000), at rest, temperatures U[0.5, 0.9] so
    that collision heating pushes particles past the explosion threshold"""
    import bench
    label, n, bounds, state, mask = bench.workload("c4", n)
    return bounds, state


def clustered_state(n, bounds, seed, clusters, per, spread):
    """uniform particles plus `clusters` knots of `per` particles within `spread`
    pixels of each other (they share all coarse quadtree levels: deep chains,
    long runs of equal high key bits, contact chains)"""
    rng = np.random.default_rng(seed)
    pos = rng.random((n, 2)) * np.array(bounds)
    masses = rng.uniform(5.0, 20.0, (n, 1))
    temps = rng.uniform(0.05, 0.1, (n, 1))
    k = 0
    for c in range(clusters):
        centre = rng.random(2) * (np.array(bounds) - 40) + 20
        for j in range(per):
            pos[k] = centre + rng.random(2) * spread
            k += 1
    perm = rng.permutation(n)          # knots scattered through the index space
    return dict(positions=pos[perm], masses=masses[perm], temperatures=temps[perm])


def tight_pairs_state(n, bounds, seed, pairs, gap, box):
    """uniform particles plus `pairs` couples `gap` px apart, all couples inside one square of
    `box` px: consecutive in Morton order, each couple owning a chain of a dozen or more
    internal nodes (one per level between the depth that isolates the couple and the depth
    that parts it), so a few hundred consecutive particles own thousands of nodes"""
    rng = np.random.default_rng(seed)
    b = np.array(bounds, dtype=np.float64)
    pos = rng.random((n, 2)) * b
    corner = b * 0.37
    for k in range(pairs):
        a = corner + rng.random(2) * box
        pos[2 * k] = a
        pos[2 * k + 1] = a + gap * (0.5 + rng.random(2))
    perm = rng.permutation(n)
    return dict(positions=pos[perm], masses=rng.uniform(5.0, 20.0, (n, 1))[perm],
                temperatures=rng.uniform(0.05, 0.1, (n, 1))[perm])


def check_tree_against_oracle(library, state, n, bounds, expect_path=None):
    """whole tree (topology, masses, centres of mass: bit-exact) and the accelerations of one
    state against the oracle.  expect_path: 1 = the elements must have come from the
    cooperative kernels, 2 = from the per-element ones (the HIP library reports which)."""
    from oracle.oracle import OracleSystem
    sim = dict(particle_count=n, min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10,
               barnes_hut_theta=0.5, min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1,
               coefficient_of_restitution=0.5, grid_cell_size_multiplier=10.0, position_correction_factor=1.05,
               collision_solver_iterations=1, explosion_efficiency=0.5, thermal_damping_factor=0.0005)
    o = OracleSystem(n, sim, np.random.default_rng(0), bounds, state=state)
    s = ParticleSystem(n, dict(sim), np.random.default_rng(0), bounds, library=library, state=state, debug=True)
    o._calculate_gravity()
    s._calculate_gravity()
    if expect_path is not None:
        assert int(s.debug_array("tree_path")[0]) == expect_path
    want, got = o.canonical_tree(), s.canonical_tree()
    assert got["tree_nodes"] == want["tree_nodes"]
    for k in TREE_KEYS_EXACT + TREE_KEYS_COM:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    acc_close(s.accelerations, o.accelerations)
    s.update()
    o.update()
    assert s.counts()["tree_nodes"] == o.num_active_nodes
    acc_close(s.accelerations, o.accelerations)
    s.close()


def out_of_bounds_state(n, bounds, seed, outside):
    """dense uniform particles, `outside` of them moved up to 30 px beyond the four edges
    (walls only act at the end of a step, so a drifted particle can sit outside
    the box when the grid is built: the grid clips it into an edge cell, whose region is
    therefore unbounded outwards), each next to an in-bounds partner it overlaps"""
    rng = np.random.default_rng(seed)
    b = np.array(bounds, dtype=np.float64)
    pos = rng.random((n, 2)) * b
    for k in range(outside):
        side = k % 4
        t = 0.02 + 0.96 * rng.random()
        d = 1.0 + rng.random() * 29.0
        # (not beyond a corner: two particles outside both ranges share all 32 quadtree levels,
        # which the reference's insertion does not survive)
        edge = [(-d, t * b[1]), (b[0] + d, t * b[1]), (t * b[0], -d), (t * b[0], b[1] + d)][side]
        pos[2 * k] = edge
        pos[2 * k + 1] = np.array(edge) + rng.normal(0.0, 1.5, 2)      # overlapping partner, maybe inside, maybe not
    perm = rng.permutation(n)
    return dict(positions=pos[perm], masses=rng.uniform(5.0, 20.0, (n, 1))[perm],
                temperatures=rng.uniform(0.05, 0.1, (n, 1))[perm])


def gaussian_cluster_state(n, bounds, sigma, seed, temp_lo=0.3, temp_hi=0.9):
    """one Gaussian blob about the centre (dense-core stress: many
    particles per grid cell, long contact chains, many solver rounds)"""
    rng = np.random.default_rng(seed)
    b = np.array(bounds, dtype=np.float64)
    pos = b / 2 + rng.normal(0.0, sigma, (n, 2))
    out = ((pos < 5.0) | (pos > b - 5.0)).any(axis=1)
    pos[out] = 5.0 + rng.random((int(out.sum()), 2)) * (b - 10.0)     # tails re-drawn uniformly (no coincident points)
    return dict(positions=pos, masses=rng.uniform(5.0, 20.0, (n, 1)),
                temperatures=rng.uniform(temp_lo, temp_hi, (n, 1)))


def check_render_feed_and_pair_potential(library):
    """device-side draw() feed against the reference's per-particle expressions
    (particle_system.py:565-624), and the exact pairwise potential energy against NumPy"""
    from psim_amd import constants
    n, bounds = 700, (400.0, 300.0)
    rng = np.random.default_rng(11)
    temps = rng.uniform(-0.2, 1.3, (n, 1))
    temps[:5, 0] = [np.nan, np.inf, -np.inf, 0.0, 1.0]
    temps[5:12, 0] = [0.1, 0.25, 0.4, 0.55, 0.7, 0.0999999999, 0.7000000001]     # keyframe boundaries
    st = dict(positions=rng.random((n, 2)) * np.array(bounds), masses=rng.uniform(5, 20, (n, 1)), temperatures=temps)
    sim = dict(particle_count=n, min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10,
               barnes_hut_theta=0.5, min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1,
               coefficient_of_restitution=0.5, grid_cell_size_multiplier=10.0, position_correction_factor=1.05,
               collision_solver_iterations=1, explosion_efficiency=0.5, thermal_damping_factor=0.0005)
    s = ParticleSystem(n, sim, rng, bounds, library=library, state=st)
    xy, rgba = s.render_feed()
    np.testing.assert_array_equal(xy, st["positions"].astype(np.int64))        # int(): toward zero
    sane = np.nan_to_num(temps[:, 0], nan=constants.COLOR_MIN_TEMP, posinf=constants.COLOR_MAX_TEMP,
                         neginf=constants.COLOR_MIN_TEMP)
    t = np.clip((sane - constants.COLOR_MIN_TEMP) / (constants.COLOR_MAX_TEMP - constants.COLOR_MIN_TEMP), 0, 1)
    want = np.zeros((n, 3), dtype=np.int64)
    keys = constants.COLOR_GRADIENT_KEYFRAMES
    for i in range(n):                      # the reference's own loop, particle_system.py:570-585
        col = keys[-1][1]
        for k in range(len(keys) - 1):
            p1, c1 = keys[k]
            p2, c2 = keys[k + 1]
            if p1 <= t[i] <= p2:
                u = (t[i] - p1) / (p2 - p1)
                col = tuple(int(c1[c] * (1 - u) + c2[c] * u) for c in range(3))
                break
        want[i] = col
    np.testing.assert_array_equal(rgba[:, :3], want)
    assert (rgba[:, 3] == 255).all()
    pos, m = st["positions"], st["masses"][:, 0]
    d = pos[:, None, :] - pos[None, :, :]
    d2 = (d ** 2).sum(-1)
    iu = np.triu_indices(n, 1)
    pe = -(0.01 * m[iu[0]] * m[iu[1]] / np.sqrt(d2[iu] + 10)).sum()
    np.testing.assert_allclose(s.exact_potential_energy(), pe, rtol=1e-12)
    s.close()


def check_exactly_coincident_particles(library):
    """Two particles at the very same position: the reference subdivides for ever (its node
    pool overflows, quadtree.py:46-74 -- the oracle reports that); this build keeps them as
    one bucket below depth 31 and carries on: they exert no force on each other (d^2 = 0,
    particle_system.py:105), are no contact (:217), exchange heat (:322), and everything
    stays finite."""
    import pytest
    from oracle.oracle import OracleSystem
    n, bounds = 500, (600, 400)
    rng = np.random.default_rng(21)
    pos = rng.random((n, 2)) * np.array(bounds)
    pos[77] = pos[401]
    st = dict(positions=pos, masses=rng.uniform(5, 20, (n, 1)), temperatures=rng.uniform(0.05, 0.9, (n, 1)))
    sim = dict(particle_count=n, min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10,
               barnes_hut_theta=0.5, min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1,
               coefficient_of_restitution=0.5, grid_cell_size_multiplier=10.0, position_correction_factor=1.05,
               collision_solver_iterations=1, explosion_efficiency=0.5, thermal_damping_factor=0.0005)
    with pytest.raises(RuntimeError):
        OracleSystem(n, sim, np.random.default_rng(0), bounds, state=st)
    s = ParticleSystem(n, sim, np.random.default_rng(0), bounds, library=library, state=st, debug=True)
    s._calculate_gravity()
    assert np.isfinite(s.accelerations).all()
    # the pair feels the same field (same position, and no mutual force)
    np.testing.assert_allclose(s.accelerations[77], s.accelerations[401], rtol=1e-12)
    t0 = s.temperatures.copy()
    s.update()
    assert s.counts()["particles"] == n
    for name in ("positions", "velocities", "accelerations", "temperatures"):
        assert np.isfinite(getattr(s, name)).all(), name
    # the coincident pair conducts heat towards each other (full-sweep conduction path)
    hot, cold = (77, 401) if t0[77, 0] > t0[401, 0] else (401, 77)
    d = s.debug_array("delta_temps")
    assert d[hot] < 0 < d[cold]
    assert np.isfinite(s.get_energies()).all()
    s.close()

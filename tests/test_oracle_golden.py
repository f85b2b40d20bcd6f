"""Pin the C oracle (oracle/psim_oracle.c) against the fixtures recorded from the
reference's own source (tests/golden/make_golden.py).  CPU only.

Bit-exact where the reference function is strict IEEE (grid, tree topology and
COM, contact pair lists and per-contact energies, post-step state while no
fastmath stage feeds it differently); the reference's fastmath stages are
compared at 1e-12 relative."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.oracle import OracleSystem

CASES = ["tiny64", "dense256", "dense400", "hot300", "fast200", "hotall60"]
TREE_KEYS = ["int_depth", "int_path", "int_mass", "int_comx", "int_comy",
             "leaf_depth", "leaf_path", "leaf_particle"]


def make_oracle(g):
    s = OracleSystem(g.n, g.sim, g.rng(), g.bounds)
    v = g.init_velocities()
    if v is not None:
        s.velocities[:] = v
    return s


def assert_tree_equal(ct, g, prefix, exact=True):
    """Topology always bit-exact; COM bit-exact unless `exact` is False (after an
    explosion the velocities carry an order-dependent sum, see oracle header)."""
    assert ct["tree_nodes"] == int(g[prefix + "tree_nodes"])
    for k in TREE_KEYS:
        if exact or not k.startswith("int_com") and k != "int_mass":
            np.testing.assert_array_equal(ct[k], g[prefix + k], err_msg=prefix + k)
        else:
            np.testing.assert_allclose(ct[k], g[prefix + k], rtol=1e-12, err_msg=prefix + k)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_stepwise(name):
    g = load_golden(name)
    s = make_oracle(g)
    s.enable_contact_log(1 << 16)
    np.testing.assert_array_equal(s.positions, g["init_positions"])
    np.testing.assert_array_equal(s.masses, g["init_masses"])
    np.testing.assert_array_equal(s.radii, g["init_radii"])
    assert_tree_equal(s.canonical_tree(), g, "init_")
    s._calculate_gravity()
    np.testing.assert_allclose(s.accelerations, g["prime_accelerations"], rtol=1e-12, atol=1e-18)
    e = np.array([s.get_total_kinetic_energy(), s.get_total_potential_energy(),
                  s.get_total_thermal_energy()])
    np.testing.assert_allclose(e, g["prime_energies"], rtol=1e-12)
    exploded_before = False
    for step in range(1, g.steps + 1):
        s.update()
        p = "s%d_" % step
        cand, ncont, nodes, n_after, n_expl = g["counts"][step - 1]
        assert s.n_candidates == cand, (step, "candidates")
        assert s.n_contacts == ncont, (step, "contacts")
        assert s.num_active_nodes == nodes, (step, "nodes")
        assert s.num_particles == n_after, (step, "n")
        np.testing.assert_array_equal(s.grid_offsets, g[p + "grid_offsets"])
        np.testing.assert_array_equal(s.grid_indices, g[p + "grid_indices"])
        ij, vals = s.contact_log()
        np.testing.assert_array_equal(ij, g[p + "contact_ij"], err_msg="step %d pair list" % step)
        np.testing.assert_allclose(vals, g[p + "contact_vals"], rtol=1e-9, atol=1e-15)
        np.testing.assert_array_equal(s.exploded, g[p + "exploded"])
        assert_tree_equal(s.canonical_tree(), g, p, exact=not exploded_before)
        exploded_before = exploded_before or n_expl > 0
        np.testing.assert_allclose(s.delta_temps, g[p + "delta_temps"].ravel(), rtol=1e-11, atol=1e-18)
        np.testing.assert_allclose(s.positions, g[p + "positions"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(s.velocities, g[p + "velocities"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(s.accelerations, g[p + "accelerations"], rtol=1e-11, atol=1e-16)
        np.testing.assert_allclose(s.temperatures, g[p + "temperatures"], rtol=1e-11, atol=1e-16)
        e = np.array([s.get_total_kinetic_energy(), s.get_total_potential_energy(),
                      s.get_total_thermal_energy()])
        np.testing.assert_allclose(e, g["energies"][step - 1], rtol=1e-10)
        tot = np.array([s.pe_gain_collisions, s.ke_loss_collisions, s.heat_generated_collisions])
        np.testing.assert_allclose(tot, g["collision_totals"][step - 1], rtol=1e-9, atol=1e-13)


def test_oracle_tree_gravity_20k():
    g = load_golden("tree20k")
    s = make_oracle(g)
    assert_tree_equal(s.canonical_tree(), g, "init_")
    s._calculate_gravity()
    np.testing.assert_allclose(s.accelerations, g["prime_accelerations"], rtol=1e-11, atol=1e-18)
    np.testing.assert_allclose(s.get_total_potential_energy(), g["prime_energies"][1], rtol=1e-11)


def test_oracle_default_run_100_steps():
    """The reference's own default configuration (config.json verbatim), 100 steps."""
    g = load_golden("default2000")
    s = make_oracle(g)
    s._calculate_gravity()
    np.testing.assert_allclose(s.get_total_potential_energy(), g["prime_energies"][1], rtol=1e-12)
    for step in range(1, 101):
        s.update()
        cand, ncont, nodes, n_after, n_expl = g["counts"][step - 1]
        assert (s.n_candidates, s.n_contacts, s.num_active_nodes, s.num_particles) == \
            (cand, ncont, nodes, n_after), step
        if step in (1, 2, 10, 50, 100):
            p = "s%d_" % step
            np.testing.assert_allclose(s.positions, g[p + "positions"], rtol=1e-11, atol=1e-10)
            np.testing.assert_allclose(s.velocities, g[p + "velocities"], rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(s.temperatures, g[p + "temperatures"], rtol=1e-10)
    e = np.array([s.get_total_kinetic_energy(), s.get_total_potential_energy(),
                  s.get_total_thermal_energy()])
    np.testing.assert_allclose(e, g["energies"][99], rtol=1e-9)

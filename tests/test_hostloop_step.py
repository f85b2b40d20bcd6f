"""CPU checks of the step's host orchestration and per-element kernel bodies:
the same sources as the HIP library, compiled by tests/hostloop/Makefile into a
test-only host-loop build (each kernel body runs as a plain for loop), checked
stage by stage against the golden fixtures and the C oracle.  The cooperative
HIP primitives (radix sort, scan, reduction, wave-cooperative walk) are NOT
covered here -- they only exist on the GPU and are covered by the gpu tests."""
import os
import subprocess

import numpy as np
import pytest

import parity
from psim_amd.particle_system import ParticleSystem

HOSTLOOP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostloop")
HOSTLOOP = os.path.join(HOSTLOOP_DIR, "libpsim_hostloop.so")


@pytest.fixture(scope="session", autouse=True)
def build_hostloop():
    subprocess.check_call(["make", "-s", "-C", HOSTLOOP_DIR])


@pytest.mark.parametrize("name", ["tiny64", "dense256", "dense400", "hot300", "fast200", "hotall60"])
def test_golden_stepwise(name):
    parity.run_golden_stepwise(name, HOSTLOOP, walk_mode=1)


@pytest.mark.parametrize("name", ["hot300", "hotall60"])
def test_explosion_log_lines_match_the_reference(name):
    """info line per explosion in ascending particle index, warning line when every
    neighbour explodes too, one "removed" line per step (particle_system.py:683-689, 727-730)"""
    parity.check_explosion_log(name, HOSTLOOP, walk_mode=1)


def test_first_step_contact_outputs_bit_exact_without_priming():
    n = 20000
    bounds, st = parity.c4_state(n)
    st = dict(st, velocities=np.random.default_rng(3).standard_normal((n, 2)) * 2.0)
    parity.run_against_oracle(n, SIM, 1, bounds, 2, HOSTLOOP, walk_mode=1, state=st, prime=False, min_contacts=2000)


def test_assigned_arrays_reach_the_next_step_and_views_are_read_only():
    """`ps.velocities = v` is seen by the next update() (the reference's attributes are plain
    arrays); an in-place edit of a handed-out view raises instead of being lost"""
    n, bounds = 800, (700, 500)
    a = ParticleSystem(n, dict(SIM), np.random.default_rng(2), bounds, library=HOSTLOOP, walk_mode=1)
    b = ParticleSystem(n, dict(SIM), np.random.default_rng(2), bounds, library=HOSTLOOP, walk_mode=1)
    c = ParticleSystem(n, dict(SIM), np.random.default_rng(2), bounds, library=HOSTLOOP, walk_mode=1)
    with pytest.raises(ValueError):
        a.velocities[0, 0] = 1.0
    with pytest.raises(ValueError):
        a.velocities = np.zeros((n + 1, 2))
    v = np.random.default_rng(9).standard_normal((n, 2))
    for s in (a, b, c):
        s._calculate_gravity()
    a.velocities = v
    with b.editing() as arrays:
        arrays["velocities"][:] = v
    for s in (a, b, c):
        s.update()
    assert np.array_equal(a.positions, b.positions)
    assert not np.array_equal(a.positions, c.positions)
    assert abs(a.get_total_kinetic_energy() - b.get_total_kinetic_energy()) == 0.0


def test_walk_counts_after_explosions():
    """the counting walk runs over the tree as it was built (before the explosions of the
    same step removed particles)"""
    g = parity.load_golden("hot300")
    s = parity.make_system(g, HOSTLOOP, 1)
    s._calculate_gravity()
    s.update()
    assert s.counts()["exploded"] > 30
    visits, inter = s.walk_counts()
    assert visits > inter > g.n * 20


def test_default_run_first_steps():
    parity.run_golden_stepwise("default2000", HOSTLOOP, walk_mode=1)


def test_tree_gravity_20k():
    g = parity.load_golden("tree20k")
    s = parity.make_system(g, HOSTLOOP, walk_mode=1)
    s._calculate_gravity()
    parity.check_tree(s, g, "init_")
    parity.acc_close(s.accelerations, g["prime_accelerations"])
    np.testing.assert_allclose(s.get_energies()[1], g["prime_energies"][1], rtol=1e-11)


SIM = dict(min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10, barnes_hut_theta=0.5,
           min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1, coefficient_of_restitution=0.5,
           grid_cell_size_multiplier=10.0, position_correction_factor=1.05, collision_solver_iterations=1,
           explosion_efficiency=0.5, thermal_damping_factor=0.0005)


@pytest.mark.parametrize("clusters,per,spread", [(12, 3, 1e-3), (2, 70, 2e-3), (4, 20, 2e-5)])
def test_near_coincident_knots_vs_oracle(clusters, per, spread):
    """knots of particles that share the 16 coarsest quadtree levels: exercises the
    run fix-up of the Morton sort (short runs, and runs long enough to fall back to
    the full radix sort), deep chains of single-child nodes and contact chains"""
    n, bounds = 3000, (1500, 900)
    st = parity.clustered_state(n, bounds, 77, clusters, per, spread)
    parity.run_against_oracle(n, SIM, 1, bounds, 3, HOSTLOOP, walk_mode=1, state=st)


def test_two_collision_sweeps_hot_start_vs_oracle():
    """collision_solver_iterations = 2 at 4x default density with a hot start: the second
    sweep's position half follows the first sweep's velocity half (contact chains, explosions)"""
    sim = dict(SIM, min_temp=0.8, max_temp=1.02, collision_solver_iterations=2)
    parity.run_against_oracle(2500, sim, 6, (1060, 600), 3, HOSTLOOP, walk_mode=1)


@pytest.mark.parametrize("walls", [True, False])
def test_particles_outside_the_box_vs_oracle(walls):
    """contacts among particles the grid clips into its edge cells (with the walls stage off they
    stay outside for every step): the re-test's skipping of out-of-reach neighbour cells
    must treat those cells as unbounded"""
    n, bounds = 1500, (700, 500)
    st = parity.out_of_bounds_state(n, bounds, 11, 120)
    parity.run_against_oracle(n, SIM, 1, bounds, 3, HOSTLOOP, walk_mode=1, state=st, stage_mask=31 if walls else 31 - 8)


@pytest.mark.parametrize("mask", [0, 2 | 4 | 8, 1 | 8, 31 - 2])
def test_stage_masks_vs_oracle(mask):
    """stage switches (the reference has none; the oracle mirrors them): gravity only;
    conduction without collisions (full neighbour sweep instead of the sweep around
    touched slots); collisions without conduction; everything but conduction"""
    sim = dict(SIM, min_temp=0.2, max_temp=0.9)
    parity.run_against_oracle(4000, sim, 9, (1100, 800), 3, HOSTLOOP, walk_mode=1, stage_mask=mask)


def test_large_hypothesis_path_of_the_contact_solver_vs_oracle():
    """the contact solver's path for large contact lists (changed slots compacted into a work
    list, pending contacts checked before the re-test) forced on at a small size: a dense
    Gaussian blob, many fixed-point iterations, contact pair lists bit-exact"""
    n, bounds = 6000, (900, 900)
    st = parity.gaussian_cluster_state(n, bounds, 90.0, 3)
    parity.run_against_oracle(n, SIM, 1, bounds, 3, HOSTLOOP, walk_mode=1, state=st, debug=3, min_contacts=3000)
    sim = dict(SIM, min_temp=0.8, max_temp=1.02, collision_solver_iterations=2)
    parity.run_against_oracle(2500, sim, 6, (1060, 600), 3, HOSTLOOP, walk_mode=1, debug=3)


def test_full_retest_mode_of_the_contact_solver_vs_oracle():
    """debug bit 2: the re-test trusts no per-slot digest (every touched slot counts as changed
    in every iteration, nothing is carried over unexamined).  The contact lists of the dense
    cases must equal the oracle's exactly as they do with the digests' shortcut, i.e. the
    shortcut changed none of them (small and large-hypothesis paths)"""
    n, bounds = 6000, (900, 900)
    st = parity.gaussian_cluster_state(n, bounds, 90.0, 3)
    parity.run_against_oracle(n, SIM, 1, bounds, 3, HOSTLOOP, walk_mode=1, state=st, debug=5, min_contacts=3000)
    parity.run_against_oracle(n, SIM, 1, bounds, 2, HOSTLOOP, walk_mode=1, state=st, debug=7, min_contacts=2000)
    sim = dict(SIM, min_temp=0.8, max_temp=1.02, collision_solver_iterations=2)
    parity.run_against_oracle(2500, sim, 6, (1060, 600), 3, HOSTLOOP, walk_mode=1, debug=5)


def test_tree_walk_against_all_pairs():
    """theta = 0 opens every node, so the walk must reproduce the all-pairs sum (to
    summation order); theta = 0.5 must stay within Barnes-Hut's approximation error"""
    n, bounds = 1500, (700, 500)
    for theta, tol in ((0.0, 1e-11), (0.5, 5e-2)):
        s = ParticleSystem(n, dict(SIM, barnes_hut_theta=theta), np.random.default_rng(4), bounds,
                           library=HOSTLOOP, walk_mode=1)
        s._calculate_gravity()
        brute, _ = s.brute_force_accelerations()
        scale = np.abs(brute).max()
        assert np.abs(s.accelerations - brute).max() <= tol * scale, theta


def test_exactly_coincident_particles_form_a_bucket():
    parity.check_exactly_coincident_particles(HOSTLOOP)


def test_tight_pairs_tree_vs_oracle():
    n, bounds = 8000, (2000.0, 1000.0)
    st = parity.tight_pairs_state(n, bounds, 3, pairs=160, gap=1e-4, box=20.0)
    parity.check_tree_against_oracle(HOSTLOOP, st, n, bounds, expect_path=2)


def test_reference_attributes_and_the_diagnostic_getters():
    """`grid_offsets` / `grid_indices` / `qtree.boundary` as the reference's attributes
    (particle_system.py:393-401): zeros before the first step, afterwards the CSR grid of the particles
    the step started with; update counts and stage times of a multi-step call; the lane statistics of
    the walk add up"""
    n, bounds = 1500, (900, 600)
    s = ParticleSystem(n, dict(SIM), np.random.default_rng(8), bounds, library=HOSTLOOP)
    cells = s.grid_width * s.grid_height
    assert s.grid_offsets.shape == (cells + 1,) and not s.grid_offsets.any()
    assert s.grid_indices.shape == (n,) and not s.grid_indices.any()
    box = s.qtree.boundary
    assert (box.x, box.y, box.width, box.height) == (0, 0, bounds[0], bounds[1])
    s._calculate_gravity()
    s.update()
    off, idx = s.grid_offsets, s.grid_indices
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) >= 0)
    assert sorted(idx.tolist()) == list(range(n))
    with pytest.raises(ValueError):
        off[0] = 1
    cell_of = np.repeat(np.arange(cells), np.diff(off))
    assert cell_of.shape == (n,)
    for k in range(0, n, 97):
        members = idx[off[cell_of[k]]:off[cell_of[k] + 1]]
        assert np.all(np.diff(members) > 0)                  # ascending particle index inside a cell
    s.update(3)
    assert s.particle_updates() == 3 * s.num_particles
    t = s.stage_times_ms()
    assert set(t) >= {"tree", "gravity", "collisions"} and all(v >= 0.0 for v in t.values())
    p = s.walk_profile()
    visits, inter = s.walk_counts()
    assert (p["visits"], p["interactions"]) == (visits, inter)
    assert p["uniform_accept_trips"] + p["uniform_open_trips"] + p["box_accept_trips"] + p["box_open_trips"] \
        + p["full_trips"] <= p["trips"]
    s.close()


def test_settled_list_with_pairs_that_did_not_touch_vs_oracle():
    """a clumped state switches the first hypothesis' margin on; the sweep then settles on a list
    that still holds the pairs that did not touch (rows of -1 in the raw log): contacts, counts and
    states must be the oracle's all the same, and such rows must really occur in this case"""
    n, bounds = 4000, (1100, 800)
    parity.run_against_oracle(n, SIM, 3, bounds, 4, HOSTLOOP)
    s = ParticleSystem(n, dict(SIM), np.random.default_rng(3), bounds, library=HOSTLOOP, debug=1)
    s._calculate_gravity()
    held = 0
    for _ in range(4):
        s.update()
        raw = s.debug_array("contact_ij", holes=True)
        kept = s.debug_array("contact_ij")
        assert len(kept) == s.counts()["contacts"] == len(s.debug_array("contact_vals"))
        assert np.all(kept >= 0)
        held += len(raw) - len(kept)
    assert held > 0
    s.close()

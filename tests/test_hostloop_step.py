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


@pytest.mark.parametrize("name", ["tiny64", "dense256", "dense400", "hot300", "fast200"])
def test_golden_stepwise(name):
    parity.run_golden_stepwise(name, HOSTLOOP, walk_mode=1)


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

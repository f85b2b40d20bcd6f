"""The collision stage split over bands of grid rows (world sizes 2..5), every rank played in
turn by one process on the test-only host-loop build: bit-identical to the unsharded step,
with and without the whole-grid fallback.  tests/test_gpu_slab.py runs the same cases on the
HIP library."""
import os
import subprocess
import types

import numpy as np
import pytest

import slab_cases as sc

HERE = os.path.dirname(os.path.abspath(__file__))
HOSTLOOP_DIR = os.path.join(HERE, "hostloop")
HOSTLOOP = os.path.join(HOSTLOOP_DIR, "libpsim_hostloop.so")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", HOSTLOOP_DIR])


@pytest.mark.parametrize("name,world,steps", [("default2000", 3, 4), ("hot300", 2, 6), ("fast200", 3, 5),
                                              ("dense400", 2, 5), ("hotall60", 2, 2), ("tiny64", 4, 3)])
def test_golden_cases_by_bands(name, world, steps):
    sc.run_emulated(HOSTLOOP, sc.case(name), world, steps)


def test_two_sweeps_and_the_fallback():
    """dense256: two solver sweeps on 16-px cells, where chains of dependent contacts are longer
    than two halo rows in most steps: the seam records disagree and the step is solved again on
    the whole grid"""
    fallbacks, contacts = sc.run_emulated(HOSTLOOP, sc.case("dense256"), 2, 6)
    assert fallbacks > 0 and contacts > 0


def test_wider_halo_avoids_the_fallback():
    fallbacks, _ = sc.run_emulated(HOSTLOOP, sc.case("dense256"), 2, 6, ranks=[0], halo_rows=10)
    assert fallbacks == 0


@pytest.mark.parametrize("world,rank", [(4, 1), (5, 2), (5, 4)])
def test_dense_hot_disk_by_bands(world, rank):
    """the stress recipe at 60 000 particles: thousands of contacts in every band"""
    fallbacks, contacts = sc.run_emulated(HOSTLOOP, sc.case("c4:60000"), world, 3, ranks=[rank])
    assert contacts > 3000 and fallbacks == 0


def test_export_block_too_small():
    fallbacks, contacts = sc.run_emulated(HOSTLOOP, sc.case("c4:20000"), 3, 3, ranks=[1], shrink_blocks_at=2)
    assert contacts > 500


def _chain_case(lo, hi, multiplier=10.0):
    n, state = sc.chain_state(3000, lo, hi)
    c = sc.case("default2000")
    c.sim['grid_cell_size_multiplier'] = multiplier
    return types.SimpleNamespace(n=n, sim=c.sim, bounds=(2000, 1000), rng=lambda: np.random.default_rng(1),
                                 init_velocities=lambda: None, state=state)


def test_contact_chain_across_a_seam():
    """a chain of overlapping particles from y = 440 to y = 520 across the seam at y = 480 (row 6
    of 80-px cells; rank 0 owns rows 0-5): contacts on both sides of the seam and across it
    depend on one another, the chain lies inside the two halo rows, so both ranks record the
    same seam contacts and no fallback is needed"""
    fallbacks, contacts = sc.run_emulated(HOSTLOOP, _chain_case(440.0, 520.0), 2, 3, row_bounds=[0, 6, 13])
    assert fallbacks == 0 and contacts >= 2 * 15


def test_contact_chain_longer_than_the_halo():
    """the same chain from y = 100 to y = 900, every contact depending on the one above it: it
    crosses both seams of a three-rank split and is longer than any halo, so what a rank does
    not see above its band reaches the seam: the ranks' seam records disagree, and the fallback
    solves the step on the whole grid (16-px cells, then the default 80-px cells)"""
    fallbacks, contacts = sc.run_emulated(HOSTLOOP, _chain_case(100.0, 900.0, 2.0), 3, 2, row_bounds=[0, 20, 40, 63])
    assert fallbacks > 0 and contacts > 200
    fallbacks, contacts = sc.run_emulated(HOSTLOOP, _chain_case(100.0, 900.0), 3, 2, ranks=[1], row_bounds=[0, 4, 9, 13])
    assert fallbacks > 0 and contacts > 100


@pytest.mark.parametrize("mask", [31 - 1, 31 - 2, 8 + 4, 1, 1 + 2 + 4 + 8])
def test_stage_masks_by_bands(mask):
    """collisions off (conduction then sweeps the whole grid on every rank), conduction off,
    cooling and walls only, collisions only, explosions off"""
    sc.run_emulated(HOSTLOOP, sc.case("dense400"), 2, 3, ranks=[1], stage_mask=mask)


def test_coincident_pair_forces_the_full_conduction_sweep():
    """two particles at exactly the same place are no contact for the collision test
    (particle_system.py:217) and touching for conduction (:322): every rank then does the
    conduction sweep on the whole grid"""
    c = sc.case("dense400")
    g = sc.load_golden("dense400")
    rng = g.rng()
    pos = rng.random((g.n, 2)) * np.array(g.bounds)
    masses = rng.uniform(g.sim['min_mass'], g.sim['max_mass'], (g.n, 1))
    temps = rng.uniform(g.sim['min_temp'], g.sim['max_temp'], (g.n, 1))
    pos[17] = pos[230]
    c.state = dict(positions=pos, masses=masses, temperatures=temps)
    sc.run_emulated(HOSTLOOP, c, 2, 2, stage_mask=31 - 16)

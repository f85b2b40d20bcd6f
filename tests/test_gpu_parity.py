"""GPU parity tests (MI355X): the HIP library, called through the C ABI, against
the golden fixtures recorded from the reference and against the C oracle on the
same seeded inputs."""
import ctypes

import numpy as np
import pytest

import parity
from conftest import load_golden
from psim_amd import _capi
from psim_amd.particle_system import ParticleSystem

pytestmark = pytest.mark.gpu

HIP = _capi.HIP_LIBRARY


def test_library_is_the_hip_build():
    L = _capi.load(HIP)
    assert b"HIP gfx950" in L.psim_version()


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 16384, 16385, 100003, 1 << 20, (1 << 22) + 77])
def test_device_primitives(n):
    """radix sort (u64 full range with duplicates, u32 sub-range, stable with
    payload), exclusive scan, ordered reduction vs host results"""
    L = _capi.load(HIP)
    rc = L.psim_selftest(0, n, 1234 + n)
    assert rc == 0, L.psim_last_error().decode()


@pytest.mark.parametrize("walk_mode", [0, 1, 2])
@pytest.mark.parametrize("name", ["tiny64", "dense256", "dense400", "hot300", "fast200", "hotall60"])
def test_golden_stepwise(name, walk_mode):
    parity.run_golden_stepwise(name, HIP, walk_mode=walk_mode)


def test_default_run_100_steps():
    parity.run_golden_stepwise("default2000", HIP, walk_mode=0)
    g = load_golden("default2000")
    s = parity.make_system(g, HIP, 0, debug=False)
    s._calculate_gravity()
    s.update(100)
    np.testing.assert_allclose(s.positions, g["s100_positions"], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(s.temperatures, g["s100_temperatures"], rtol=1e-9)
    np.testing.assert_allclose(s.get_energies(), g["energies"][99], rtol=1e-8)


def test_tree_gravity_20k():
    g = load_golden("tree20k")
    for walk_mode in (0, 1, 2):
        s = parity.make_system(g, HIP, walk_mode=walk_mode)
        s._calculate_gravity()
        parity.check_tree(s, g, "init_")
        parity.acc_close(s.accelerations, g["prime_accelerations"])
        np.testing.assert_allclose(s.get_energies()[1], g["prime_energies"][1], rtol=1e-11)


SIM = dict(particle_count=0, min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10,
           barnes_hut_theta=0.5, min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1,
           coefficient_of_restitution=0.5, grid_cell_size_multiplier=10.0, position_correction_factor=1.05,
           collision_solver_iterations=1, explosion_efficiency=0.5, thermal_damping_factor=0.0005)


def test_oracle_50k_full_physics():
    """default density, 50k particles, 3 full steps vs the oracle"""
    parity.run_against_oracle(50000, SIM, 5, (10000, 5000), 3, HIP)


def _host_threads():
    import os
    return max(1, min(16, os.cpu_count() or 1))


def test_full_size_1Mi_disk_gravity_only_vs_oracle():
    """the 1Mi uniform-disk gravity-only configuration, two steps"""
    import bench
    label, n, bounds, state, mask = bench.workload("c1")
    parity.run_against_oracle(n, SIM, 42, bounds, 2, HIP, state=state, stage_mask=mask,
                              oracle_threads=_host_threads(), oracle_fast_pow=True)


@pytest.mark.parametrize("n,sigma", [(100000, 1060.0), (30000, 250.0)])
def test_dense_cluster_vs_oracle(n, sigma):
    """a Gaussian blob: up to a few hundred particles per grid cell; the tighter one is a
    single connected lump of contacts (tens of fixed-point iterations, a thousand
    dependency rounds per step).  Two steps, contact pair lists bit-exact."""
    bounds = (4000, 4000)
    st = parity.gaussian_cluster_state(n, bounds, sigma, 3)
    parity.run_against_oracle(n, SIM, 1, bounds, 2, HIP, state=st, oracle_threads=_host_threads())


def test_large_hypothesis_path_forced_vs_oracle():
    """the contact solver's large-list path (compacted list of changed slots, pending check
    before the re-test) forced on at sizes the small path normally takes"""
    bounds = (4000, 4000)
    st = parity.gaussian_cluster_state(30000, bounds, 250.0, 3)
    parity.run_against_oracle(30000, SIM, 1, bounds, 2, HIP, state=st, oracle_threads=_host_threads(), debug=3)
    sim = dict(SIM, min_temp=0.8, max_temp=1.02, collision_solver_iterations=2)
    parity.run_against_oracle(20000, sim, 6, (3000, 1700), 3, HIP, debug=3)


def test_full_retest_mode_of_the_contact_solver_vs_oracle():
    """debug bit 2: the re-test trusts no per-slot digest (every touched slot counts as changed
    in every iteration, nothing is carried over unexamined).  The contact lists of the dense
    cases must equal the oracle's exactly as they do with the digests' shortcut, i.e. the
    shortcut changed none of them"""
    bounds = (4000, 4000)
    st = parity.gaussian_cluster_state(30000, bounds, 250.0, 3)
    parity.run_against_oracle(30000, SIM, 1, bounds, 2, HIP, state=st, oracle_threads=_host_threads(), debug=5)
    sim = dict(SIM, min_temp=0.8, max_temp=1.02, collision_solver_iterations=2)
    parity.run_against_oracle(20000, sim, 6, (3000, 1700), 3, HIP, debug=5)


def test_oracle_dense_hot_20k():
    """4x default density, hot start: contact chains and explosions"""
    sim = dict(SIM, min_temp=0.8, max_temp=1.02, collision_solver_iterations=2)
    parity.run_against_oracle(20000, sim, 6, (3000, 1700), 3, HIP)


def test_side_stream_on_and_off_give_identical_states():
    """The grid build and the position half of the contact solver run on a second stream
    underneath the walk (overlap=True, the default) or on the main stream (overlap=False):
    same kernels, same order per stream, so every array must come out bit-identical."""
    n, bounds = 300000, (24000, 12500)
    a = ParticleSystem(n, dict(SIM), np.random.default_rng(5), bounds, library=HIP, overlap=True)
    b = ParticleSystem(n, dict(SIM), np.random.default_rng(5), bounds, library=HIP, overlap=False)
    for s in (a, b):
        s._calculate_gravity()
    for _ in range(4):
        a.update()
        b.update()
        assert a.counts()["contacts"] == b.counts()["contacts"] > 0
        for name in ("positions", "velocities", "accelerations", "temperatures"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert (a.pe_gain_collisions, a.ke_loss_collisions, a.heat_generated_collisions) == \
               (b.pe_gain_collisions, b.ke_loss_collisions, b.heat_generated_collisions)
    ms_whole, ms_resident = a.time_walk(2, 0.0), a.time_walk(2, 6.0)
    assert 0.0 < ms_whole < 1000.0 and 0.0 < ms_resident < 1000.0
    a.close()
    b.close()


@pytest.mark.parametrize("walls", [True, False])
def test_particles_outside_the_box_vs_oracle(walls):
    """contacts among particles the grid clips into its edge cells: the re-test's skipping of
    out-of-reach neighbour cells must treat those cells as unbounded"""
    n, bounds = 20000, (2600, 1800)
    st = parity.out_of_bounds_state(n, bounds, 11, 1500)
    parity.run_against_oracle(n, SIM, 1, bounds, 3, HIP, state=st, stage_mask=31 if walls else 31 - 8)


def test_walk_variants_agree_and_counts():
    g = load_golden("tree20k")
    a = parity.make_system(g, HIP, walk_mode=0)
    b = parity.make_system(g, HIP, walk_mode=1)
    a._calculate_gravity()
    b._calculate_gravity()
    parity.acc_close(a.accelerations, b.accelerations, rtol=1e-12)
    visits, inter = a.walk_counts()
    from oracle.oracle import OracleSystem
    o = OracleSystem(g.n, g.sim, g.rng(), g.bounds)
    o._calculate_gravity()
    # the oracle counts empty leaves it never enters? no: it only enters children with mass > 0
    assert inter == o.last_interactions


def test_sharded_step_over_rccl_single_rank():
    """the staged (multi-GPU) entry points + RCCL all-gather on one rank: bit-identical
    to psim_step on the golden cases (world_size > 1 is covered on CPU with gloo)"""
    from test_distributed_gloo import run_ranks
    run_ranks(1, "nccl", "hip", "dense256", 6)
    run_ranks(1, "nccl", "hip", "hot300", 6)


@pytest.mark.parametrize("clusters,per,spread", [(12, 3, 1e-3), (2, 70, 2e-3), (4, 20, 2e-5)])
def test_near_coincident_knots_vs_oracle(clusters, per, spread):
    """knots of particles sharing the 16 coarsest quadtree levels: run fix-up of the
    Morton sort (short runs; runs long enough to fall back to the full radix sort),
    deep single-child chains, contact chains"""
    n, bounds = 3000, (1500, 900)
    st = parity.clustered_state(n, bounds, 77, clusters, per, spread)
    parity.run_against_oracle(n, SIM, 1, bounds, 3, HIP, state=st)


@pytest.mark.parametrize("mask", [0, 2 | 4 | 8, 1 | 8, 31 - 2])
def test_stage_masks_vs_oracle(mask):
    """stage switches (the reference has none; the oracle mirrors them): gravity only;
    conduction without collisions (full neighbour sweep instead of the sweep around
    touched slots); collisions without conduction; everything but conduction"""
    sim = dict(SIM, min_temp=0.2, max_temp=0.9)
    parity.run_against_oracle(4000, sim, 9, (1100, 800), 3, HIP, stage_mask=mask)


def test_tree_walk_against_all_pairs():
    """theta = 0 opens every node: the walk must reproduce the all-pairs FP64 sum up
    to summation order; theta = 0.5 stays within Barnes-Hut's approximation error"""
    from psim_amd.particle_system import ParticleSystem
    n, bounds = 20000, (3000, 2000)
    for theta, tol in ((0.0, 1e-11), (0.5, 5e-2)):
        s = ParticleSystem(n, dict(SIM, barnes_hut_theta=theta), np.random.default_rng(4), bounds)
        s._calculate_gravity()
        brute, ms = s.brute_force_accelerations()
        scale = np.abs(brute).max()
        assert np.abs(s.accelerations - brute).max() <= tol * scale, theta
        assert ms > 0


def test_render_feed_and_pair_potential():
    parity.check_render_feed_and_pair_potential(HIP)


def test_exactly_coincident_particles_form_a_bucket():
    parity.check_exactly_coincident_particles(HIP)


def test_whole_tree_from_the_cooperative_kernels_vs_oracle():
    # 391 tiles; hundreds of nodes span tiles, up to the root
    n, bounds = 100003, (3000.0, 2000.0)
    st = dict(positions=np.random.default_rng(5).random((n, 2)) * np.array(bounds),
              masses=np.random.default_rng(6).uniform(5.0, 20.0, (n, 1)),
              temperatures=np.full((n, 1), 0.07))
    parity.check_tree_against_oracle(HIP, st, n, bounds, expect_path=1)


def test_tile_with_too_many_nodes_takes_the_per_element_kernels():
    n, bounds = 8000, (2000.0, 1000.0)
    st = parity.tight_pairs_state(n, bounds, 3, pairs=160, gap=1e-4, box=20.0)
    parity.check_tree_against_oracle(HIP, st, n, bounds, expect_path=2)


def test_deep_chains_across_tiles_vs_oracle():
    # couples spread over the whole domain: deep chains, few per tile, the cooperative path
    n, bounds = 30000, (2000.0, 1000.0)
    st = parity.tight_pairs_state(n, bounds, 4, pairs=60, gap=1e-3, box=900.0)
    parity.check_tree_against_oracle(HIP, st, n, bounds, expect_path=1)


def test_settled_list_with_pairs_that_did_not_touch_vs_oracle():
    """a clumped state: the first hypothesis reaches beyond contact distance and the sweep settles on a
    list that still holds the pairs that did not touch (rows of -1 in the raw log); contacts, counts
    and states are the oracle's all the same, and such rows do occur"""
    n, bounds = 40000, (3400, 2500)
    parity.run_against_oracle(n, SIM, 3, bounds, 4, HIP)
    s = ParticleSystem(n, dict(SIM), np.random.default_rng(3), bounds, library=HIP, debug=1)
    s._calculate_gravity()
    held = 0
    for _ in range(4):
        s.update()
        raw = s.debug_array("contact_ij", holes=True)
        kept = s.debug_array("contact_ij")
        assert len(kept) == s.counts()["contacts"] == len(s.debug_array("contact_vals"))
        held += len(raw) - len(kept)
    assert held > 0
    s.close()

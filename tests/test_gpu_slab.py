"""The collision stage split over bands of grid rows, on the HIP library (MI355X): one process
plays every rank of a world in turn with exactly the entry points and ranges the real ranks
use, and the state must be bit-identical to the unsharded step of the same library.  This is
the N > 1 path with the GPU kernels inside the collected suite; the transport between real
ranks (RCCL) moves these same blocks and is covered by the one-rank RCCL tests of
tests/test_gpu_scale.py and, on CPU, by the gloo tests."""
import types

import numpy as np
import pytest

import bench
import slab_cases as sc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,world,steps", [("default2000", 3, 4), ("hot300", 2, 6), ("dense400", 2, 5),
                                              ("dense256", 2, 6), ("tiny64", 4, 3)])
def test_golden_cases_by_bands(name, world, steps):
    sc.run_emulated(None, sc.case(name), world, steps)


def test_two_sweeps_and_the_fallback():
    fallbacks, contacts = sc.run_emulated(None, sc.case("dense256"), 2, 6, ranks=[1])
    assert fallbacks > 0 and contacts > 0


@pytest.mark.parametrize("world,rank", [(4, 1), (8, 6)])
def test_dense_hot_disk_by_bands(world, rank):
    """the stress recipe at 300 000 particles: contacts and explosions in every band"""
    fallbacks, contacts = sc.run_emulated(None, sc.case("c4:300000"), world, 3, ranks=[rank])
    assert contacts > 30000


def test_export_block_too_small():
    fallbacks, contacts = sc.run_emulated(None, sc.case("c4:100000"), 3, 3, ranks=[1], shrink_blocks_at=2)
    assert contacts > 5000


def _chain_case(lo, hi, multiplier=10.0):
    n, state = sc.chain_state(3000, lo, hi)
    c = sc.case("default2000")
    c.sim['grid_cell_size_multiplier'] = multiplier
    return types.SimpleNamespace(n=n, sim=c.sim, bounds=(2000, 1000), rng=lambda: np.random.default_rng(1),
                                 init_velocities=lambda: None, state=state)


def test_contact_chain_across_a_seam():
    fallbacks, contacts = sc.run_emulated(None, _chain_case(440.0, 520.0), 2, 3, row_bounds=[0, 6, 13])
    assert fallbacks == 0 and contacts >= 2 * 15


def test_contact_chain_longer_than_the_halo():
    fallbacks, contacts = sc.run_emulated(None, _chain_case(100.0, 900.0, 2.0), 3, 2, row_bounds=[0, 20, 40, 63])
    assert fallbacks > 0 and contacts > 200


@pytest.mark.parametrize("mask", [31 - 1, 31 - 2, 1 + 2 + 4 + 8])
def test_stage_masks_by_bands(mask):
    sc.run_emulated(None, sc.case("dense400"), 2, 3, ranks=[1], stage_mask=mask)


def test_uniform_1Mi_eight_bands():
    """1,048,576 particles at the default density, 8 ranks, rank 3 played: 20 000 contacts per
    step spread over all bands, seams agree"""
    fallbacks, contacts = sc.run_emulated(None, sc.case("uni:1048576"), 8, 2, ranks=[3])
    assert fallbacks == 0 and contacts > 30000


def test_c3_16Mi_eight_bands_equal_the_unsharded_step():
    """the 16,777,216-particle rotating disk (the multi-GPU configuration) as 8 ranks, rank 5
    played, one step: bit-identical to the unsharded step (which the next module checks against
    the oracle)"""
    label, n, bounds, state, mask = bench.workload("c3")
    c = types.SimpleNamespace(n=n, sim=dict(bench.SIM), bounds=bounds, rng=lambda: np.random.default_rng(42),
                              init_velocities=lambda: None, state=state)
    fallbacks, contacts = sc.run_emulated(None, c, 8, 1, ranks=[5])
    assert fallbacks == 0 and contacts > 100000


def test_staged_entry_points_refuse_ordinary_host_pointers():
    """the staged calls take device (or page-locked) pointers; an ordinary host array must come back
    as an error before any kernel is handed its address"""
    import ctypes
    from psim_amd import _capi
    from psim_amd.particle_system import ParticleSystem
    n = 2000
    # (first the check by itself, on buffers no kernel touches)
    assert _capi.load().psim_selftest(0, 1000, 1) == 0, _capi.load().psim_last_error().decode()
    s = ParticleSystem(n, dict(bench.SIM), np.random.default_rng(1), (2000, 1000))
    s._calculate_gravity()
    L, ctx = s._lib, s._ctx
    _capi.check(L, L.psim_stage_build(ctx))
    host = np.zeros((n, 2))
    rc = L.psim_stage_gravity(ctx, 0, n, ctypes.c_void_p(host.ctypes.data))
    assert rc == -1 and b"page-locked" in L.psim_last_error()
    with pytest.raises(ValueError):
        _capi.check(L, L.psim_stage_kick(ctx, ctypes.c_void_p(host.ctypes.data)))
    s.close()

"""GPU tests at the sizes of the benchmark configurations (MI355X): the 16 Mi rotating
disk through the staged multi-GPU entry points against the oracle, the dense 64 Mi stress
case through size-independent properties (and its recipe at 1 Mi against the oracle),
several full-size steps of the 4 Mi configuration, clumped clouds, mass explosions, the
wide-offset walk kernels, and the headless runner / checkpoint / log lines on the HIP
library."""
import json
import os
import socket

import numpy as np
import pytest

import bench
import parity
from psim_amd import _capi
from psim_amd.particle_system import ParticleSystem

pytestmark = pytest.mark.gpu

HIP = _capi.HIP_LIBRARY
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = dict(bench.SIM)


def _host_threads():
    return max(1, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def one_rank_rccl():
    """a one-rank RCCL process group in this process (the sharded class needs one)"""
    import torch
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_c3_16Mi_rotating_disk_staged_vs_oracle(one_rank_rccl):
    """the 16,777,216-particle rotating disk (the multi-GPU configuration), prime + one full
    step through the staged entry points (ShardedParticleSystem, one rank, RCCL) against the
    threaded oracle: node count, grid, ordered contact list (about 2 % of the particles),
    exploded set bit-exact; state within the tolerances of run_against_oracle"""
    from psim_amd.distributed import ShardedParticleSystem
    label, n, bounds, state, mask = bench.workload("c3")
    assert n == 1 << 24
    parity.run_against_oracle(n, SIM, 42, bounds, 1, None, state=state, stage_mask=mask,
                              oracle_threads=_host_threads(), oracle_fast_pow=True,
                              system_class=ShardedParticleSystem, min_contacts=100000)


def test_staged_step_with_more_results_than_the_export_block(one_rank_rccl):
    """a first step whose touched particles exceed the sharded class's initial export block:
    the export reports what it needs, the block is regrown and exported again; the contact
    list must equal the unsharded step's"""
    from psim_amd.distributed import ShardedParticleSystem
    n = 60000
    bounds, state = parity.c4_state(n)
    state = dict(state, positions=np.array(bounds) / 2 + (state["positions"] - np.array(bounds) / 2) * 0.6)   # denser still
    a = ParticleSystem(n, dict(SIM), None, bounds, state=state, debug=True)
    b = ShardedParticleSystem(n, dict(SIM), None, bounds, state=state, debug=True)
    before = b._block_words
    for s in (a, b):
        s._calculate_gravity()
        s.update()
    assert a.counts()["contacts"] > 20000 and b._block_words > before
    np.testing.assert_array_equal(a.debug_array("contact_ij"), b.debug_array("contact_ij"))
    for name in ("positions", "velocities", "temperatures"):
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
    a.close()
    b.close()


def test_c4_recipe_1Mi_vs_oracle():
    """the dense stress recipe (4x default density, hot) at 1,048,576 particles, three full
    steps against the oracle: a quarter of the particles are in contact in the first step"""
    n = 1 << 20
    bounds, state = parity.c4_state(n)
    parity.run_against_oracle(n, SIM, 42, bounds, 3, HIP, state=state, oracle_threads=_host_threads(),
                              oracle_fast_pow=True, min_contacts=300000)


def test_c4_64Mi_properties():
    """the 67,108,864-particle dense stress case itself, two full steps, through properties
    that need no oracle: particle and mass conservation, node count = 1 + 4 internal, finite
    state, and the first step's ordered contact list -- duplicate-free, i < j, visited in
    non-decreasing home cell of i (the reference's cell-major order, particle_system.py:148-197),
    and every contact whose two particles have no earlier contact in the list is a true overlap
    at the drifted positions (0 < d^2 < (r_i + r_j)^2, particle_system.py:217)"""
    n = 1 << 26
    label, n, bounds, state, mask = bench.workload("c4", n)
    s = ParticleSystem(n, dict(SIM), None, bounds, state=state, debug=True)
    del state
    m0 = float(s.masses.sum())
    s._calculate_gravity()
    drifted = s.positions + (s.velocities + 0.5 * s.accelerations)      # particle_system.py:528
    radii = s.radii[:, 0].copy()
    s.update()
    c = s.counts()
    assert c["particles"] == n and c["exploded"] == 0
    assert c["tree_nodes"] == 1 + 4 * c["internal_nodes"]
    assert c["contacts"] > n // 8
    ij = s.debug_array("contact_ij").astype(np.int64)
    assert len(ij) == c["contacts"]
    i, j = ij[:, 0], ij[:, 1]
    assert (i < j).all() and i.min() >= 0 and j.max() < n
    pair_id = i * n + j
    assert len(np.unique(pair_id)) == len(pair_id)
    cell = s.cell_size
    cx = np.clip((drifted[i, 0] / cell).astype(np.int32), 0, s.grid_width - 1)        # :745-749
    cy = np.clip((drifted[i, 1] / cell).astype(np.int32), 0, s.grid_height - 1)
    home = cy.astype(np.int64) * s.grid_width + cx
    assert (np.diff(home) >= 0).all()
    # first appearance of each particle in the list (as either member)
    order = np.arange(len(ij), dtype=np.int64)
    first = np.full(n, len(ij), dtype=np.int64)
    np.minimum.at(first, i, order)
    np.minimum.at(first, j, order)
    fresh = (first[i] == order) & (first[j] == order)
    assert fresh.sum() > len(ij) // 10
    d = drifted[j[fresh]] - drifted[i[fresh]]
    d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
    reach = (radii[i[fresh]] + radii[j[fresh]]).astype(np.float64)
    assert ((d2 > 0) & (d2 < reach * reach)).all()
    del drifted, ij, pair_id, first, order
    s.update()
    c = s.counts()
    assert c["particles"] + c["exploded"] == n
    assert c["tree_nodes"] == 1 + 4 * c["internal_nodes"]
    assert abs(float(s.masses.sum()) - m0) <= 1e-9 * m0 or c["particles"] < n
    for name in ("positions", "velocities", "accelerations", "temperatures"):
        assert np.isfinite(getattr(s, name)).all(), name
    assert np.isfinite(s.get_energies()).all()
    s.close()


def test_full_size_4Mi_three_steps_vs_oracle():
    """the benchmark's single-GPU configuration (4,194,304 particles, 90000x45000, seed 42,
    every stage), three consecutive steps against the oracle: the contact solver's carry-over
    state (round hint, buffer capacities) is exercised at scale"""
    parity.run_against_oracle(1 << 22, SIM, 42, (90000, 45000), 3, HIP, oracle_threads=_host_threads(),
                              oracle_fast_pow=True, min_contacts=200000)


def test_clumped_cloud_1Mi_vs_oracle():
    """a Gaussian cluster of 1,048,576 particles tight enough that more than a tenth of them
    are in contact: two steps, contact pair lists bit-exact (the regime a long run of the
    benchmark workload drifts into)"""
    n, bounds = 1 << 20, (40000, 40000)
    st = parity.gaussian_cluster_state(n, bounds, 5500.0, 5)
    s, o = parity.run_against_oracle(n, SIM, 1, bounds, 2, HIP, state=st, oracle_threads=_host_threads(),
                                     oracle_fast_pow=True, min_contacts=200000)


def test_mass_explosion_1Mi_vs_oracle():
    """hot start at 1,048,576 particles (a tenth at or above the explosion threshold): the
    exploded set, the compaction of the seven state arrays and the pushed neighbours against
    the oracle, two steps"""
    sim = dict(SIM, min_temp=0.8, max_temp=1.022)
    parity.run_against_oracle(1 << 20, sim, 8, (45000, 22500), 2, HIP, oracle_threads=_host_threads(),
                              oracle_fast_pow=True, min_exploded=50000)


def test_first_step_contact_outputs_bit_exact_without_priming():
    """no priming walk: both sides drift to bit-identical positions, so the first step's
    contact list AND the contacts' PE changes (functions of positions only) are bit-exact"""
    n = 200000
    bounds, st = parity.c4_state(n)
    st = dict(st, velocities=np.random.default_rng(3).standard_normal((n, 2)) * 2.0)
    parity.run_against_oracle(n, SIM, 1, bounds, 2, HIP, state=st, prime=False, oracle_threads=_host_threads(),
                              min_contacts=20000)


@pytest.mark.parametrize("wide,narrow,theta", [(3, 0, 0.5), (4, 2, 0.5), (5, 0, 0.5), (5, 0, 0.7), (3, 0, 0.3), (3, 5, 0.5),
                                               (3, 5, 0.7)])
def test_wide_offset_walk_equals_narrow(wide, narrow, theta):
    """the walk kernels' 64-bit element offsets (taken for trees of 4 GiB and more, 2^27
    elements) forced on at a small size: bit-identical accelerations and states.  (5, 0): the
    default walk's hand-scheduled loop against the same kernel with the compiler's loop, with
    theta^2 a power of two (folded into the elements' tags) and not (scaled acceptance test);
    (3, 5): the hand-scheduled loop with 64-bit offsets against the compiler's with 32-bit ones."""
    n, bounds = 300000, (24000, 12500)
    sim = dict(SIM, barnes_hut_theta=theta)
    a = ParticleSystem(n, dict(sim), np.random.default_rng(5), bounds, walk_mode=narrow)
    b = ParticleSystem(n, dict(sim), np.random.default_rng(5), bounds, walk_mode=wide)
    a._calculate_gravity()
    b._calculate_gravity()
    assert np.array_equal(a.accelerations, b.accelerations)
    a.update(3)
    b.update(3)
    for name in ("positions", "velocities", "accelerations", "temperatures"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    a.close()
    b.close()


def test_walk_counts_after_explosions_use_the_trees_own_size():
    """the counting walk after a step that removed particles: the tree still describes the
    pre-explosion particles; counts must equal the ones of a walk-mode-2 system's own tree"""
    g = parity.load_golden("hot300")
    s = parity.make_system(g, HIP, 0)
    s._calculate_gravity()
    s.update()
    assert s.counts()["exploded"] > 30
    visits, inter = s.walk_counts()
    assert visits > inter > 0
    assert s.time_walk(1, 0.0) > 0
    # hot start at scale: about a tenth of the particles leave in the first step
    sim = dict(SIM, min_temp=0.8, max_temp=1.022)
    b = ParticleSystem(400000, sim, np.random.default_rng(8), (28000, 14000))
    b._calculate_gravity()
    b.update()
    assert b.counts()["exploded"] > 30000
    v, i = b.walk_counts()
    assert v > i > 400000 * 100
    assert b.time_walk(1, 6.0) > 0
    s.close()
    b.close()


@pytest.mark.parametrize("name", ["hot300", "hotall60"])
def test_explosion_log_lines_match_the_reference(name):
    parity.check_explosion_log(name, HIP)


def test_host_views_are_pinned_and_writes_reach_the_device():
    """public arrays: NumPy views of page-locked PyTorch tensors; assignment is seen by the
    next step; in-place edits of a handed-out view raise"""
    n, bounds = 5000, (2000, 1000)
    a = ParticleSystem(n, dict(SIM), np.random.default_rng(2), bounds)
    b = ParticleSystem(n, dict(SIM), np.random.default_rng(2), bounds)
    for name in ("positions", "temperatures", "radii"):
        t = a._store[name][0]
        assert t is not None and t.is_pinned()
        assert getattr(a, name).ctypes.data == t.data_ptr()            # zero-copy view
    with pytest.raises(ValueError):
        a.velocities[0, 0] = 1.0
    v = np.random.default_rng(9).standard_normal((n, 2))
    a.velocities = v
    with b.editing() as arrays:
        arrays["velocities"][:] = v
    a._calculate_gravity()
    b._calculate_gravity()
    a.update()
    b.update()
    assert np.array_equal(a.positions, b.positions)
    c = ParticleSystem(n, dict(SIM), np.random.default_rng(2), bounds)
    c._calculate_gravity()
    c.update()
    assert not np.array_equal(a.positions, c.positions)
    xy, rgba = a.render_feed()
    np.testing.assert_array_equal(xy, a.positions.astype(np.int64))
    for s in (a, b, c):
        s.close()


def test_headless_runner_on_the_hip_library(tmp_path, monkeypatch):
    """the headless twin of main.py (main.py:111-167) on libpsim_hip.so: 101 ticks of the
    default configuration, the energy line at ticks 0 and 100 with main.py's keys, and the
    same final state as stepping the class directly"""
    from psim_amd import headless
    monkeypatch.chdir(tmp_path)
    config = json.load(open(os.path.join(ROOT, "particle-simulator_amd", "config.json")))
    headless.setup_logging(config)
    system = headless.run(config, ticks=101)
    assert system.num_particles == config["simulation"]["particle_count"]
    log = open(tmp_path / "runs" / config["run_id"] / "simulation.log").read()
    lines = [l for l in log.splitlines() if "Tick=" in l]
    assert len(lines) == 2 and "Tick=0," in lines[0] and "Tick=100," in lines[1]
    for key in ("Kinetic=", "Thermal=", "Potential=", "TotalSystemEnergy=", "Delta_E=", "AccumulatedKE_Loss=",
                "AccumulatedPE_Gain=", "AccumulatedHeat="):
        assert key in lines[1]
    ref = ParticleSystem(config["simulation"]["particle_count"], dict(config["simulation"]),
                         np.random.default_rng(config["master_seed"]), (2000, 1000))
    ref._calculate_gravity()
    ref.update(101)
    for name in ("positions", "velocities", "temperatures"):
        np.testing.assert_array_equal(getattr(system, name), getattr(ref, name), err_msg=name)
    system.close()
    ref.close()


def test_checkpoint_restore_on_the_hip_library(tmp_path):
    """6 steps in one go == 3 steps, checkpoint, restore into a new context, 3 steps (hot
    start: explosions shrink the arrays on both sides of the checkpoint)"""
    sim = dict(SIM, particle_count=20000, min_temp=0.6, max_temp=1.01)
    bounds = (3000, 1700)
    a = ParticleSystem(20000, dict(sim), np.random.default_rng(8), bounds)
    a._calculate_gravity()
    a.update(6)
    b = ParticleSystem(20000, dict(sim), np.random.default_rng(8), bounds)
    b._calculate_gravity()
    b.update(3)
    assert b.num_particles < 20000
    path = str(tmp_path / "ckpt.npz")
    b.save_checkpoint(path)
    b.close()
    c = ParticleSystem.from_checkpoint(path)
    c.update(3)
    assert c.num_particles == a.num_particles < 20000
    for name in ("positions", "velocities", "accelerations", "temperatures", "masses", "radii"):
        np.testing.assert_array_equal(getattr(c, name), getattr(a, name), err_msg=name)
    a.close()
    c.close()

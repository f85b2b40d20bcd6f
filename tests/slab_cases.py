"""Cases shared by the CPU and GPU tests of the collision stage split over bands of grid rows
(psim_stage_slab_* in include/psim.h): one process plays every rank of a world in turn
(`ShardedParticleSystem(emulate=(world, rank))`), calling exactly the entry points with
exactly the ranges the real ranks use, and the result must be bit-identical to the unsharded
step of the same library."""
import types

import numpy as np

import bench
from conftest import load_golden  # noqa: F401 (re-exported)
from psim_amd.distributed import ShardedParticleSystem
from psim_amd.particle_system import ParticleSystem


def case(name):
    """-> (namespace with n, sim, bounds, rng(), init_velocities(), state)"""
    if name.startswith("c4:") or name.startswith("uni:"):
        if name.startswith("c4:"):
            _, n, bounds, state, _ = bench.workload("c4", int(name[3:]))
        else:
            n = int(name[4:])
            side = float(np.sqrt(n / 1e-3))
            bounds, state = (side, side), None
        return types.SimpleNamespace(n=n, sim=dict(bench.SIM), bounds=bounds, rng=lambda: np.random.default_rng(42),
                                     init_velocities=lambda: None, state=state)
    g = load_golden(name)
    return types.SimpleNamespace(n=g.n, sim=dict(g.sim), bounds=g.bounds, rng=g.rng, init_velocities=g.init_velocities,
                                 state=None)


def chain_state(n_background, chain_lo, chain_hi, spacing=3.5, x=1000.0, seed=5):
    """a sparse uniform cloud in the reference's 2000 x 1000 box plus a vertical chain of
    overlapping particles from y = chain_lo to y = chain_hi (radii 2..4, so neighbours at
    `spacing` px overlap): a chain of contacts, each depending on the one before"""
    rng = np.random.default_rng(seed)
    pos = rng.random((n_background, 2)) * np.array([2000.0, 1000.0])
    ys = np.arange(chain_lo, chain_hi, spacing)
    chain = np.stack([np.full(len(ys), x) + 0.25 * np.sin(ys), ys], axis=1)
    # (particle indices ascend down the chain, and so does the sweep's visiting order -- home
    # cells in row-major order, particle_system.py:148 -- so every contact of the chain
    # depends on the one above it)
    pos = np.concatenate([pos, chain])
    n = len(pos)
    masses = rng.uniform(5.0, 20.0, (n, 1))
    temps = rng.uniform(0.05, 0.1, (n, 1))
    vel = rng.standard_normal((n, 2)) * 0.5
    return n, dict(positions=pos, masses=masses, temperatures=temps, velocities=vel)


def compare(ref, sh, where):
    assert ref.num_particles == sh.num_particles, where
    for f in ("positions", "velocities", "accelerations", "temperatures"):
        a, b = getattr(ref, f), getattr(sh, f)
        assert np.array_equal(a, b), (where, f, float(np.abs(a - b).max()), int((a != b).sum()))
    ca, cb = ref.counts(), sh.counts()
    for k in ("particles", "tree_nodes", "candidates", "contacts", "exploded", "internal_nodes"):
        assert ca[k] == cb[k], (where, k, ca, cb)
    # (the three totals are sums of the ranks' partial sums: the same terms in another order)
    t = (ref.pe_gain_collisions, ref.ke_loss_collisions, ref.heat_generated_collisions)
    u = (sh.pe_gain_collisions, sh.ke_loss_collisions, sh.heat_generated_collisions)
    assert np.allclose(t, u, rtol=1e-11, atol=1e-300), (where, t, u)


def run_emulated(library, c, world, steps, ranks=None, row_bounds=None, halo_rows=2, shrink_blocks_at=None,
                 walk_mode=None, stage_mask=31):
    """every rank in `ranks` (default: all) of an emulated world against the unsharded step.
    Returns (fallbacks, contacts) summed over the ranks played."""
    kw = dict(library=library, walk_mode=(1 if library else 0) if walk_mode is None else walk_mode, debug=True,
              stage_mask=stage_mask)
    if c.state is not None:
        kw["state"] = c.state
    fallbacks = contacts = 0
    for rank in (range(world) if ranks is None else ranks):
        ref = ParticleSystem(c.n, dict(c.sim), c.rng(), c.bounds, **kw)
        sh = ShardedParticleSystem(c.n, dict(c.sim), c.rng(), c.bounds, emulate=(world, rank), halo_rows=halo_rows, **kw)
        if row_bounds is not None:
            sh._row_bounds = list(row_bounds)
        v = c.init_velocities()
        for s in (ref, sh):
            if v is not None:
                s.velocities = v
            s._calculate_gravity()
        for step in range(1, steps + 1):
            if shrink_blocks_at == step:
                sh._alloc_blocks(32)        # room for the header only: the export must ask for more
            ref.update()
            sh.update()
            compare(ref, sh, (world, rank, step))
            contacts += sh.counts()["contacts"]
        fallbacks += sh.fallbacks
        ref.close()
        sh.close()
    return fallbacks, contacts

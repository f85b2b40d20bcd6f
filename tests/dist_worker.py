"""Worker of tests/test_distributed_gloo.py (and of the single-GPU RCCL test):
runs one golden case sharded over WORLD_SIZE ranks and checks that the state
is bit-identical to an unsharded run of the same library."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "particle-simulator_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from conftest import load_golden
from psim_amd.particle_system import ParticleSystem
from psim_amd.distributed import ShardedParticleSystem, shard_range


def main():
    backend, library, name, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    # a rank that is still running after this many seconds prints where it is and exits
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("DIST_WORKER_DUMP_AFTER", "200")), exit=True)
    library = None if library == "hip" else library
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend, rank=rank, world_size=world)
    kw = dict(library=library, walk_mode=0 if library is None else 1, debug=True)
    if name.startswith("c4:"):
        # the dense hot disk of the benchmark's stress recipe at the given size: contacts in
        # every slab, explosions from the first steps on
        import types
        import bench
        _, n, bounds, state, _ = bench.workload("c4", int(name[3:]))
        g = types.SimpleNamespace(n=n, sim=dict(bench.SIM), bounds=bounds, rng=lambda: np.random.default_rng(42),
                                  init_velocities=lambda: None)
        kw["state"] = state
    else:
        g = load_golden(name)
    # the unsharded run first, its states kept on the host, and its context closed: the sharded
    # run that follows is then the only user of the device in this process
    ref = ParticleSystem(g.n, dict(g.sim), g.rng(), g.bounds, **kw)
    v = g.init_velocities()
    if v is not None:
        ref.velocities = v
    ref._calculate_gravity()
    want = []
    for step in range(1, steps + 1):
        ref.update()
        want.append(dict(n=ref.num_particles, counts=ref.counts(),
                         totals=(ref.pe_gain_collisions, ref.ke_loss_collisions, ref.heat_generated_collisions),
                         **{f: getattr(ref, f).copy() for f in ("positions", "velocities", "accelerations", "temperatures")}))
    ref.close()
    sh = ShardedParticleSystem(g.n, dict(g.sim), g.rng(), g.bounds, **kw)
    if v is not None:
        sh.velocities = v
    sh._calculate_gravity()
    for step in range(1, steps + 1):
        if step == 2:
            # blocks with room for the header only: psim_stage_slab_export must report what it
            # needs and write nothing else, psim_stage_apply must say so on every rank, and the
            # step must succeed once all ranks have made room
            sh._alloc_blocks(32)
        if step == 3:
            sh.warm_up_exchanges()      # (between two steps the exchange buffers are scratch)
        sh.update()
        w = want[step - 1]
        assert w["n"] == sh.num_particles, (step, rank)
        for f in ("positions", "velocities", "accelerations", "temperatures"):
            a, b = w[f], getattr(sh, f)
            assert np.array_equal(a, b), (step, rank, f, float(np.abs(a - b).max()))
        ca, cb = w["counts"], sh.counts()
        for k in ("particles", "tree_nodes", "candidates", "contacts", "exploded", "internal_nodes"):
            assert ca[k] == cb[k], (step, rank, k, ca, cb)
        # (the three totals are sums of the ranks' partial sums: the same terms in another order)
        u = (sh.pe_gain_collisions, sh.ke_loss_collisions, sh.heat_generated_collisions)
        assert np.allclose(w["totals"], u, rtol=1e-11, atol=1e-300), (step, rank, w["totals"], u)
    # the shards tile the index space exactly
    n = 1000003
    cover = [shard_range(n, r, world) for r in range(world)]
    assert cover[0][0] == 0 and cover[-1][1] == n
    assert all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok: %s, %d steps, n %d -> %d, fallbacks %d" % (rank, name, steps, g.n, sh.num_particles, sh.fallbacks))


if __name__ == "__main__":
    main()

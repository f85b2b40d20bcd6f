"""One simulation sharded over the GPUs of a node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

The reference is single-process, so nothing here has a counterpart in it.  Every
rank holds the full particle state and rebuilds the same grid and tree (the
builds are deterministic, so no broadcast is needed).  The two stages that
dominate the step are split by range and their outputs exchanged:

  * gravity: rank r walks the tree for a contiguous range of Morton-sorted
    target particles; accelerations are all-gathered in sorted order (16 B per
    particle) -- one all-gather per step;
  * conduction + cooling: when the step needs the full neighbour sweep, rank r
    sweeps a contiguous range of grid slots (whole runs of row-major cells, i.e. a
    spatial slab) and temperatures are all-gathered in slot order (8 B per
    particle); normally conduction only involves the few slots the contacts
    touched, every rank does all of it and nothing is exchanged.

Each element is computed by exactly one rank with the same arithmetic as the
single-GPU step, so the result is bit-identical for any number of ranks.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .particle_system import ParticleSystem

_ALIGN = 256


def shard_range(n, rank, world):
    """Contiguous [lo, hi) share of n items for `rank`, in whole blocks of 256;
    every rank's share has the same padded length `per` (for all_gather)."""
    per = (n + world * _ALIGN - 1) // (world * _ALIGN) * _ALIGN
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


class ShardedParticleSystem(ParticleSystem):
    """`ParticleSystem` whose `update()` splits the walk and the conduction sweep
    over the ranks of `group`.  All ranks must construct it with the same
    arguments and call `update()` together."""

    def __init__(self, *args, group=None, **kw):
        super().__init__(*args, **kw)
        self._group = group
        self._rank = dist.get_rank(group)
        self._world = dist.get_world_size(group)
        on_gpu = kw.get("library") is None
        self._tdev = torch.device("cuda", self._device) if on_gpu else torch.device("cpu")
        n = self.num_particles
        _, _, per = shard_range(n, 0, self._world)
        self._acc = torch.zeros((per * self._world, 2), dtype=torch.float64, device=self._tdev)
        self._tmp = torch.zeros((per * self._world,), dtype=torch.float64, device=self._tdev)

    def _gather(self, buf, per, async_op=False):
        """in-place all-gather of the ranks' equal shards of buf[: per * world];
        with async_op the collective is only started and its handle returned"""
        if self._world == 1:
            return None
        flat = buf[: per * self._world]
        mine = flat[self._rank * per:(self._rank + 1) * per]
        return dist.all_gather_into_tensor(flat, mine.clone(), group=self._group, async_op=async_op)

    def update(self, steps: int = 1):
        L, ctx = self._lib, self._ctx
        for _ in range(int(steps)):
            self._sync_config()
            _capi.check(L, L.psim_stage_build(ctx))
            n = int(L.psim_num_particles(ctx))
            lo, hi, per = shard_range(n, self._rank, self._world)
            # The walk is only submitted here (to the device's default stream, which is
            # also PyTorch's); the exchange is started behind it, and while both are in
            # flight the half of the contact solver that needs no velocities runs on the
            # library's second stream.
            _capi.check(L, L.psim_stage_gravity(ctx, lo, hi, ctypes.c_void_p(self._acc.data_ptr())))
            work = self._gather(self._acc, per, async_op=True)
            _capi.check(L, L.psim_stage_settle_positions(ctx))
            if work is not None:
                work.wait()
            if self._tdev.type == "cuda":
                torch.cuda.current_stream().synchronize()
            _capi.check(L, L.psim_stage_kick_collide(ctx, ctypes.c_void_p(self._acc.data_ptr())))
            _capi.check(L, L.psim_stage_conduction(ctx, lo, hi, ctypes.c_void_p(self._tmp.data_ptr())))
            exchanges = ctypes.c_int32(0)
            _capi.check(L, L.psim_stage_conduction_exchanges(ctx, ctypes.byref(exchanges)))
            if exchanges.value:      # same answer on every rank
                self._gather(self._tmp, per)
                if self._tdev.type == "cuda":
                    torch.cuda.current_stream().synchronize()
            _capi.check(L, L.psim_stage_finish(ctx, ctypes.c_void_p(self._tmp.data_ptr())))
            self._stale = True
            t = (ctypes.c_double * 3)()
            _capi.check(L, L.psim_collision_stats(ctx, t))
            self.pe_gain_collisions, self.ke_loss_collisions, self.heat_generated_collisions = t[0], t[1], t[2]
            old = self.num_particles
            self.num_particles = int(L.psim_num_particles(ctx))
            self._drain_explosion_log(old)

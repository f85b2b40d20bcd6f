"""One simulation sharded over the GPUs of a node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

The reference is single-process, so nothing here has a counterpart in it.  Every
rank holds the full particle state and rebuilds the same grid and tree (the
builds are deterministic, so no broadcast is needed).  The two stages that
dominate the step are split by range and their outputs exchanged:

  * gravity: rank r walks the tree for a contiguous range of Morton-sorted
    target particles; accelerations are all-gathered in sorted order (16 B per
    particle) -- one all-gather per step;
  * contact detection: rank r sweeps a contiguous range of grid slots (cells are
    row-major, so that is a slab of the domain; every rank holds all positions,
    so a slab needs no halo) and the ranks' contact key lists are all-gathered
    (8 B per contact, a few per cent of the particle count); the contact solver
    itself runs on every rank on the gathered list;
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
        self._keys = torch.zeros((max(4096, per // 8),), dtype=torch.int64, device=self._tdev)
        self._cat_keys = None
        self._all_keys = None
        # the small exchanges of the contact lists run on a stream of their own, so that they
        # are not ordered behind the walk (which is on the default stream)
        self._side = torch.cuda.Stream(device=self._tdev) if self._tdev.type == "cuda" else None

    def _detect_and_gather(self, lo, hi):
        """contact detection of the first collision sweep on this rank's slab of grid slots,
        then the ranks' key lists gathered -> (tensor of all keys, contacts, candidates,
        coincident pairs).

        One collective: every rank sends a block of `cap` words -- three counters (contacts,
        candidates, coincident pairs), then its keys -- and reads the counters of all blocks
        back once.  A rank whose list did not fit makes everybody grow the block and repeat
        (the counters tell all ranks the same thing)."""
        L, ctx = self._lib, self._ctx
        out = (ctypes.c_int64 * 4)()
        import contextlib
        side = torch.cuda.stream(self._side) if self._side is not None else contextlib.nullcontext()
        while True:
            cap = self._keys.numel() - 3
            _capi.check(L, L.psim_stage_detect(ctx, lo, hi, ctypes.c_void_p(self._keys.data_ptr() + 24), cap, out))
            mine = [int(out[0]), int(out[1]), int(out[2])]
            if self._world == 1:
                if out[3] == out[0]:
                    return self._keys[3:], mine[0], mine[1], mine[2]
                # (no fill: the library writes the keys on a stream of its own, and a fill queued
                # on this one behind the walk would land on top of them later)
                self._keys = torch.empty((mine[0] + mine[0] // 4 + 4096,), dtype=torch.int64, device=self._tdev)
                continue
            with side:
                self._keys[:3] = torch.tensor(mine, dtype=torch.int64).to(self._tdev, non_blocking=False)
                block = self._keys.numel()
                if self._all_keys is None or self._all_keys.numel() != self._world * block:
                    self._all_keys = torch.empty((self._world * block,), dtype=torch.int64, device=self._tdev)
                dist.all_gather_into_tensor(self._all_keys, self._keys, group=self._group)
                blocks = self._all_keys.view(self._world, block)
                every = blocks[:, :3].cpu()
                n_each = [int(v) for v in every[:, 0]]
                most = max(n_each)
                if most > cap:
                    self._keys = torch.empty((most + most // 4 + 4096,), dtype=torch.int64, device=self._tdev)
                    continue
                total = sum(n_each)
                keys = self._keys
                if total > 0:
                    keys = torch.cat([blocks[r, 3: 3 + n_each[r]] for r in range(self._world)])
                    if self._side is not None:
                        self._side.synchronize()
                self._cat_keys = keys      # (kept alive while the library copies from it)
            return keys, total, int(every[:, 1].sum()), int(every[:, 2].sum())

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
            self._sync_state()
            _capi.check(L, L.psim_stage_build(ctx))
            n = int(L.psim_num_particles(ctx))
            lo, hi, per = shard_range(n, self._rank, self._world)
            # The walk is only submitted here (to the device's default stream, which is
            # also PyTorch's); the exchange is started behind it, and while both are in
            # flight the half of the contact solver that needs no velocities runs on the
            # library's second stream.
            _capi.check(L, L.psim_stage_gravity(ctx, lo, hi, ctypes.c_void_p(self._acc.data_ptr())))
            detected = None
            if (self._stage_mask & _capi.STAGE_COLLISIONS) and n > 0:
                detected = self._detect_and_gather(lo, hi)
            work = self._gather(self._acc, per, async_op=True)
            if detected is not None:
                keys, n_keys, cand, coinc = detected
                _capi.check(L, L.psim_stage_settle_positions_from(ctx, ctypes.c_void_p(keys.data_ptr()), n_keys,
                                                                  cand, coinc))
            else:
                _capi.check(L, L.psim_stage_settle_positions(ctx))
            # (the library's main stream is the device's default stream, which is also the
            # stream work.wait() makes wait for the collective: no host synchronisation)
            if work is not None:
                work.wait()
            _capi.check(L, L.psim_stage_kick_collide(ctx, ctypes.c_void_p(self._acc.data_ptr())))
            _capi.check(L, L.psim_stage_conduction(ctx, lo, hi, ctypes.c_void_p(self._tmp.data_ptr())))
            exchanges = ctypes.c_int32(0)
            _capi.check(L, L.psim_stage_conduction_exchanges(ctx, ctypes.byref(exchanges)))
            if exchanges.value:      # same answer on every rank
                self._gather(self._tmp, per)
            _capi.check(L, L.psim_stage_finish(ctx, ctypes.c_void_p(self._tmp.data_ptr())))
            self._stale = True
            t = (ctypes.c_double * 3)()
            _capi.check(L, L.psim_collision_stats(ctx, t))
            self.pe_gain_collisions, self.ke_loss_collisions, self.heat_generated_collisions = t[0], t[1], t[2]
            old = self.num_particles
            self.num_particles = int(L.psim_num_particles(ctx))
            self._drain_explosion_log(old)

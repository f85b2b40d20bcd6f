"""One simulation split over the GPUs of a node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

The reference is single-process, so nothing here has a counterpart in it.  Every
rank holds the full particle state and rebuilds the same tree (the build is
deterministic, so no broadcast is needed).  The two stages that dominate the step
are split and their outputs exchanged:

  * gravity: rank r walks the tree for a contiguous range of Morton-sorted target
    particles; accelerations are all-gathered in sorted order (16 B per particle);
  * collisions + conduction: rank r owns a slab of the domain -- a band of whole
    grid rows (cells are row-major, particle_system.py:749) -- and solves it together
    with `halo_rows` rows on either side, on a grid of that band only: cell sort,
    contact detection, the exact Gauss-Seidel contact solver, conduction sums.  It
    exports the results of its OWN rows (state of the touched particles, conduction
    sums) and, for each seam, a record of every contact it saw in the two rows along
    it.  The blocks are all-gathered; when both ranks along every seam recorded the
    same contacts, every rank's view of its neighbour's rows was the true one, the
    owned results together are exactly the sequential sweep's, and every rank applies
    all of them.  When a seam disagrees (a chain of dependent contacts longer than the
    halo reached it) every rank runs the step's collision stage again on the whole
    grid -- always exact, sharded whenever the halo suffices.

Explosions (particle_system.py:651-730) and the full conduction sweep that a pair of
exactly coinciding particles forces are done by every rank alike on the whole grid,
built only in the steps that need it.  The result is bit-identical to the unsharded
step for any number of ranks.

The per-step exchange allocates nothing: every buffer (accelerations, the rank's
block, the gathered blocks, and with the "host" transport their page-locked host
twins) exists before the first step and only grows, on all ranks together, when a
block did not fit.

`emulate=(world, rank)` plays all `world` ranks in turn in ONE process on one
device and times the share of `rank` alone (bench.py --emulate-rank): what one GPU
of an N-GPU job computes per step, and the bytes it exchanges, measured without N
GPUs.  It calls exactly the entry points with exactly the ranges the real ranks use.
"""
import contextlib
import ctypes
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .particle_system import ParticleSystem

_ALIGN = 256
HEADER_WORDS = 32


def shard_range(n, rank, world):
    """Contiguous [lo, hi) share of n items for `rank`, in whole blocks of 256;
    every rank's share has the same padded length `per` (for all_gather)."""
    per = (n + world * _ALIGN - 1) // (world * _ALIGN) * _ALIGN
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


def balanced_rows(row_counts, world):
    """Row boundaries B[0..world] (B[0] = 0, B[world] = number of rows) that give every rank
    about the same number of particles: rank r owns rows [B[r], B[r+1])."""
    rows = len(row_counts)
    cum = np.concatenate(([0], np.cumsum(np.asarray(row_counts, dtype=np.int64))))
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        want = (total * r) // world
        # first row boundary at which at least `want` particles lie before it
        b = int(np.searchsorted(cum, want, side="left"))
        b = max(bounds[-1], min(b, rows))
        bounds.append(b)
    bounds.append(rows)
    return bounds


class ShardedParticleSystem(ParticleSystem):
    """`ParticleSystem` whose `update()` splits the walk over ranges of sorted particles and the
    collision stage over slabs of grid rows.  All ranks must construct it with the same
    arguments and call `update()` together."""

    def __init__(self, *args, group=None, halo_rows=2, emulate=None, transport=None, **kw):
        super().__init__(*args, **kw)
        self._group = group
        self._emulate = emulate
        if emulate is not None:
            self._world, self._rank = int(emulate[0]), int(emulate[1])
            if not (0 <= self._rank < self._world):
                raise ValueError("emulate=(world, rank): rank out of range")
        else:
            self._rank = dist.get_rank(group)
            self._world = dist.get_world_size(group)
        # (asked of the library itself: the pointers it is handed must be of the memory it works on)
        on_gpu = b"HIP" in self._lib.psim_version()
        self._tdev = torch.device("cuda", self._device) if on_gpu else torch.device("cpu")
        # how the exchanged buffers travel: "device" = the collective works on the device
        # tensors (RCCL); "host" = through page-locked host copies (any backend that moves CPU
        # tensors: several ranks sharing one GPU over gloo); "cpu" = the buffers are host memory
        # already (the test-only host-loop build)
        if transport is None:
            if not on_gpu:
                transport = "cpu"
            elif emulate is not None or self._world == 1 or dist.get_backend(group) == "nccl":
                transport = "device"
            else:
                transport = "host"
        self._transport = transport
        self._halo = max(2, int(halo_rows))
        n = self.num_particles
        W = self._world
        _, _, per = shard_range(n, 0, W)
        self._acc = torch.zeros((per * W, 2), dtype=torch.float64, device=self._tdev)
        self._shard = torch.zeros((per, 2), dtype=torch.float64, device=self._tdev)
        self._block_words = 0
        self._blocks = self._block = self._blocks_h = self._block_h = None
        self._acc_h = self._shard_h = None
        if transport == "host":
            self._acc_h = torch.zeros((per * W, 2), dtype=torch.float64, pin_memory=True)
            self._shard_h = torch.zeros((per, 2), dtype=torch.float64, pin_memory=True)
        self._alloc_blocks(HEADER_WORDS + 4096 + per)
        self._own_slots = (ctypes.c_int64 * W)()
        # rows of the grid per rank, balanced by particle count (all ranks hold the same state)
        y = self._host["positions"][:, 1]
        row = np.clip((y / self.cell_size).astype(np.int64), 0, self.grid_height - 1)
        self._row_bounds = balanced_rows(np.bincount(row, minlength=self.grid_height), W)
        self.fallbacks = 0
        self.emulated_ms = {}
        self.exchange_bytes = {}

    # ------------------------------------------------------------------ buffers
    def _alloc_blocks(self, words):
        """(all ranks call this with the same size)"""
        W = self._world
        self._block_words = int(words)
        self._block = torch.zeros((self._block_words,), dtype=torch.int64, device=self._tdev)
        self._blocks = torch.zeros((W * self._block_words,), dtype=torch.int64, device=self._tdev)
        if self._transport == "host":
            self._block_h = torch.zeros((self._block_words,), dtype=torch.int64, pin_memory=True)
            self._blocks_h = torch.zeros((W * self._block_words,), dtype=torch.int64, pin_memory=True)

    def _sync(self):
        if self._tdev.type == "cuda":
            torch.cuda.synchronize(self._tdev)

    def band_of(self, rank):
        """(row_lo, row_hi, own_lo, own_hi) of `rank`: its own rows and the halo around them"""
        lo, hi = self._row_bounds[rank], self._row_bounds[rank + 1]
        if hi <= lo:
            return lo, lo, lo, lo
        return max(0, lo - self._halo), min(self.grid_height, hi + self._halo), lo, hi

    def _rebalance(self):
        """next step's rows from this step's per-rank populations (the same numbers on every
        rank, so the same rows): each rank's rows are taken as evenly populated"""
        W = self._world
        counts = [int(v) for v in self._own_slots]
        total = sum(counts)
        if W == 1 or total == 0:
            return
        if max(counts) * W <= total * 1.08:
            return
        per_row = np.zeros(self.grid_height, dtype=np.int64)
        for r in range(W):
            lo, hi = self._row_bounds[r], self._row_bounds[r + 1]
            if hi > lo:
                base, extra = divmod(counts[r], hi - lo)
                per_row[lo:hi] = base
                per_row[lo:lo + extra] += 1
        self._row_bounds = balanced_rows(per_row, W)

    # ------------------------------------------------------------------ exchanges
    def _gather_acc_start(self, per):
        """all-gather of the ranks' equal shards of self._acc; returns a handle to wait for
        (device transport) or None (done, or nothing to do)"""
        W = self._world
        if W == 1 or self._emulate is not None:
            return None
        flat = self._acc[: per * W]
        mine = flat[self._rank * per:(self._rank + 1) * per]
        if self._transport == "host":
            # (the walk and this copy are on the device's default stream)
            self._shard_h[:per].copy_(mine, non_blocking=False)
            dist.all_gather_into_tensor(self._acc_h[: per * W], self._shard_h[:per], group=self._group)
            flat.copy_(self._acc_h[: per * W], non_blocking=False)
            return None
        self._shard[:per].copy_(mine)
        return dist.all_gather_into_tensor(flat, self._shard[:per], group=self._group, async_op=self._transport == "device")

    def _gather_blocks(self):
        W = self._world
        if self._emulate is not None:
            return
        if W == 1:
            self._blocks.copy_(self._block)
            return
        if self._transport == "host":
            self._block_h.copy_(self._block, non_blocking=False)
            dist.all_gather_into_tensor(self._blocks_h, self._block_h, group=self._group)
            self._blocks.copy_(self._blocks_h, non_blocking=False)
            return
        dist.all_gather_into_tensor(self._blocks, self._block, group=self._group)
        self._sync()

    def warm_up_exchanges(self):
        """one all-gather of each kind on the step's own buffers (both are scratch between steps),
        so that a backend's set-up on first use (communicator, channels, registration of the
        buffers) is not inside the first timed step.  All ranks call it together."""
        if self._world == 1 or self._emulate is not None:
            return
        _, _, per = shard_range(self.num_particles, 0, self._world)
        work = self._gather_acc_start(per)
        if work is not None:
            work.wait()
        self._gather_blocks()
        self._sync()

    # ------------------------------------------------------------------ pieces of a step
    def _export(self, rank):
        """the current band's block -> self._block (real ranks) / its place in self._blocks (emulation)"""
        L, ctx = self._lib, self._ctx
        need = ctypes.c_int64(0)
        if self._emulate is not None:
            ptr = self._blocks.data_ptr() + 8 * rank * self._block_words
        else:
            ptr = self._block.data_ptr()
        _capi.check(L, L.psim_stage_slab_export(ctx, ctypes.c_void_p(ptr), self._block_words, ctypes.byref(need)))
        return int(need.value)

    def _slab(self, rank):
        L, ctx = self._lib, self._ctx
        _capi.check(L, L.psim_stage_slab_positions(ctx, *self.band_of(rank)))

    def _timed(self, name):
        sps = self

        class _T:
            def __enter__(self):
                sps._sync()
                self.t0 = time.perf_counter()

            def __exit__(self, *a):
                sps._sync()
                sps.emulated_ms[name] = sps.emulated_ms.get(name, 0.0) + (time.perf_counter() - self.t0) * 1e3
        return _T() if self._emulate is not None else contextlib.nullcontext()

    def update(self, steps: int = 1):
        L, ctx = self._lib, self._ctx
        W, me = self._world, self._rank
        emu = self._emulate is not None
        acc_ptr = ctypes.c_void_p(self._acc.data_ptr())
        for _ in range(int(steps)):
            self._sync_config()
            self._sync_state()
            with self._timed("build"):
                _capi.check(L, L.psim_stage_build(ctx))
            n = int(L.psim_num_particles(ctx))
            lo, hi, per = shard_range(n, me, W)
            # The walk is only submitted here (to the device's default stream, which is also
            # PyTorch's); the exchange of accelerations is started behind it, and while both are
            # in flight the position half of this rank's slab runs on the library's second stream.
            with self._timed("walk_and_slab_positions"):
                _capi.check(L, L.psim_stage_gravity(ctx, lo, hi, acc_ptr))
                work = self._gather_acc_start(per) if self._transport == "device" else None
                self._slab(me)
                if self._transport != "device":
                    self._gather_acc_start(per)
                if work is not None:
                    work.wait()      # (makes the default stream, the library's main stream, wait: no host synchronisation)
            if emu:
                for k in range(W):
                    if k != me:
                        klo, khi, _ = shard_range(n, k, W)
                        _capi.check(L, L.psim_stage_gravity(ctx, klo, khi, acc_ptr))
                self._sync()
            with self._timed("kick_velocities_export"):
                _capi.check(L, L.psim_stage_kick(ctx, acc_ptr))
                _capi.check(L, L.psim_stage_slab_finish(ctx))
                need = self._export(me)
            self.exchange_bytes = {"accelerations_per_rank": 16 * per, "block_needed_this_rank": 8 * need,
                                   "block_sent_per_rank": 8 * self._block_words}
            while True:
                if emu:
                    # the other ranks' slabs, one after the other (the state arrays are untouched until the apply)
                    needs = {me: need}
                    for k in range(W):
                        if k != me:
                            self._slab(k)
                            _capi.check(L, L.psim_stage_slab_finish(ctx))
                            needs[k] = self._export(k)
                            if needs[k] > self._block_words:
                                break
                    if max(needs.values()) > self._block_words:
                        # a block did not fit: every block again with room (emulation only: the
                        # real ranks keep their band's results and only export again)
                        self._alloc_blocks(max(needs.values()) * 5 // 4 + 4096)
                        self._slab(me)
                        _capi.check(L, L.psim_stage_slab_finish(ctx))
                        need = self._export(me)
                        continue
                else:
                    self._gather_blocks()
                status = ctypes.c_int32(0)
                most = ctypes.c_int64(0)
                with self._timed("apply"):
                    _capi.check(L, L.psim_stage_apply(ctx, ctypes.c_void_p(self._blocks.data_ptr()), W, self._block_words,
                                                      ctypes.byref(status), self._own_slots, ctypes.byref(most)))
                if status.value != 2:
                    # (blocks much larger than the step needed shrink, on all ranks alike: the
                    # whole block travels, whatever it holds)
                    self.exchange_bytes["block_sent_per_rank"] = 8 * self._block_words
                    if int(most.value) * 6 + 16384 < self._block_words:
                        self._alloc_blocks(int(most.value) * 3 + 4096)
                    break
                # some rank's block did not fit: all ranks grow alike and export again
                self._alloc_blocks(int(most.value) * 5 // 4 + 4096)
                need = self._export(me)
            with self._timed("finish"):
                if status.value == 1:
                    self.fallbacks += 1
                    _capi.check(L, L.psim_stage_fallback(ctx))
                _capi.check(L, L.psim_stage_finish(ctx))
            self._rebalance()
            self._stale = True
            self._stepped = True
            t = (ctypes.c_double * 3)()
            _capi.check(L, L.psim_collision_stats(ctx, t))
            self.pe_gain_collisions, self.ke_loss_collisions, self.heat_generated_collisions = t[0], t[1], t[2]
            old = self.num_particles
            self.num_particles = int(L.psim_num_particles(ctx))
            self._drain_explosion_log(old)

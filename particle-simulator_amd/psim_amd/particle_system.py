"""`ParticleSystem` with the constructor, attributes and methods of the
reference class (particle_system.py:339-804), driving the HIP library through
the C ABI of include/psim.h.  main.py of the reference runs unchanged on it:

    from psim_amd.particle_system import ParticleSystem

State lives on the GPU.  The public arrays (`positions`, `velocities`,
`accelerations`, `masses`, `inverse_masses`, `temperatures`, `radii`) are NumPy
views over page-locked host tensors owned by PyTorch (`torch.empty(...,
pin_memory=True)`; PyTorch is used for nothing else here), refreshed lazily:
reading one after `update()` downloads the current state once, by DMA straight
into the memory the view shows -- `draw()` reads them without a further copy.

Writing: assigning an array (`ps.velocities = v`) takes effect at the next
device call, as in the reference.  The views themselves are handed out
read-only, so an in-place edit (`ps.velocities[0] = ...`) raises instead of
being lost; `with ps.editing() as arrays:` gives writable views and uploads
them on exit.
"""
import collections
import ctypes
import logging

import numpy as np

from . import _capi
from . import constants

logger = logging.getLogger("particle_sim")

# quadtree.py:11 (the root box the reference's constructor logs, particle_system.py:400,407)
BoundingBox = collections.namedtuple('BoundingBox', ['x', 'y', 'width', 'height'])


class _QuadTreeHandle:
    """What `ParticleSystem.qtree` is here: the root box (quadtree.py:146); the tree itself
    lives on the device (`ParticleSystem.canonical_tree()` reads it back)."""

    def __init__(self, boundary):
        self.boundary = boundary
        self.capacity = 1


def grid_geometry(config, bounds):
    """Cell size and grid shape exactly as particle_system.py:381-390."""
    max_radius = np.sqrt(config['max_mass']).astype(int)
    base_cell_size = max_radius * 2
    multiplier = config.get('grid_cell_size_multiplier', 1.0)
    cell_size = int(base_cell_size * multiplier)
    if cell_size == 0:
        cell_size = 10
    b = np.array(bounds)
    return cell_size, int(np.ceil(b[0] / cell_size)), int(np.ceil(b[1] / cell_size))


def make_config(config, bounds, stage_mask=_capi.STAGE_ALL, walk_mode=0, debug=False, overlap=True):
    """dict from config.json's "simulation" section -> struct psim_config.
    Required and optional keys and their defaults follow particle_system.py:359-361,
    383, 397, 430-431, 462, 555, 673."""
    cell_size, gw, gh = grid_geometry(config, bounds)
    c = _capi.PsimConfig()
    c.bounds_w, c.bounds_h = float(bounds[0]), float(bounds[1])
    c.gravity_constant = float(config['gravity_constant'])
    c.softening = float(config['softening_factor'])
    c.theta = float(config['barnes_hut_theta'])
    c.heat_coeff = float(config['heat_transfer_coefficient'])
    c.restitution = float(config['coefficient_of_restitution'])
    c.pos_correct = float(config.get('position_correction_factor', 1.0))
    c.explosion_eff = float(config.get('explosion_efficiency', 0.1))
    c.damping = float(config.get('thermal_damping_factor', 0.0))
    c.explode_temp = float(constants.COLOR_MAX_TEMP)
    c.cell_size, c.grid_w, c.grid_h = cell_size, gw, gh
    c.solver_iterations = int(config.get('collision_solver_iterations', 1))
    c.stage_mask = int(stage_mask)
    c.walk_mode = int(walk_mode)
    c.debug = int(debug)        # True / 1: keep debug arrays; 3: also force the large-hypothesis re-test path (tests)
    c.overlap = 1 if overlap else 0
    return c


_FIELDS = ("positions", "velocities", "accelerations", "masses", "inverse_masses",
           "temperatures", "radii")


class ParticleSystem:
    def __init__(self, num_particles: int, config: dict, rng: np.random.Generator, bounds: tuple,
                 *, device: int = 0, stage_mask: int = _capi.STAGE_ALL, walk_mode: int = 0,
                 debug: bool = False, state: dict = None, library: str = None, overlap: bool = True):
        """Positional arguments as particle_system.py:355.  Keyword-only extras:
        device ordinal, stage mask (benchmarks), tree-walk variant, debug arrays,
        an explicit initial `state` dict (positions, masses, temperatures[,
        velocities, accelerations]) instead of the RNG draws, and the library path
        (CPU test-suite only)."""
        self._lib = _capi.load(library)
        self.num_particles = int(num_particles)
        self.config = config
        self.bounds = np.array(bounds)
        self.restitution = config['coefficient_of_restitution']
        self.position_correction_factor = config.get('position_correction_factor', 1.0)
        self.collision_solver_iterations = config.get('collision_solver_iterations', 1)
        self.barnes_hut_theta = config['barnes_hut_theta']
        self.pe_gain_collisions = 0.0
        self.ke_loss_collisions = 0.0
        self.heat_generated_collisions = 0.0
        self._stage_mask, self._walk_mode, self._debug = stage_mask, walk_mode, debug
        self._overlap = overlap
        self._device = device
        n = self.num_particles

        if state is None:
            # the three draws, in the reference's order (particle_system.py:370,373,375)
            positions = rng.random((n, 2)) * self.bounds
            masses = rng.uniform(config['min_mass'], config['max_mass'], (n, 1))
            temperatures = rng.uniform(config['min_temp'], config['max_temp'], (n, 1))
            velocities = np.zeros((n, 2), dtype=float)
            accelerations = np.zeros((n, 2), dtype=float)
        else:
            positions = np.ascontiguousarray(state['positions'], dtype=np.float64).reshape(n, 2)
            masses = np.ascontiguousarray(state['masses'], dtype=np.float64).reshape(n, 1)
            temperatures = np.ascontiguousarray(state['temperatures'], dtype=np.float64).reshape(n, 1)
            velocities = np.ascontiguousarray(state.get('velocities', np.zeros((n, 2))), dtype=np.float64).reshape(n, 2)
            accelerations = np.ascontiguousarray(state.get('accelerations', np.zeros((n, 2))), dtype=np.float64).reshape(n, 2)
        radii = np.sqrt(masses).astype(int)                 # particle_system.py:376 (int64)

        self.cell_size, self.grid_width, self.grid_height = grid_geometry(config, bounds)
        self._cfg = make_config(config, bounds, stage_mask, walk_mode, debug, overlap)
        self._cfg_key = self._config_key()
        self._ctx = ctypes.c_void_p()
        _capi.check(self._lib, self._lib.psim_create(ctypes.byref(self._cfg), device, ctypes.byref(self._ctx)))
        self._pinned = library is None      # the test-only host-loop build has no device to pin for
        self._store = {}
        self._host = {}
        self._feed = None
        self._alloc_host(n)
        for name, value in (("positions", positions), ("velocities", velocities), ("accelerations", accelerations),
                            ("masses", masses), ("inverse_masses", 1.0 / masses), ("temperatures", temperatures),
                            ("radii", radii.astype(np.int64))):
            np.copyto(self._host[name], value)
        self._stale = False
        self._dirty = False
        self._stepped = False
        self._upload()
        qtree_boundary = BoundingBox(x=0, y=0, width=self.bounds[0], height=self.bounds[1])
        self.qtree = _QuadTreeHandle(qtree_boundary)
        logger.info(f"ParticleSystem created for {num_particles} particles.")
        logger.info(f"Spatial grid initialized with cell size {self.cell_size} "
                    f"({self.grid_width}x{self.grid_height} cells).")
        logger.info(f"QuadTree initialized with boundary: {qtree_boundary}")

    # ------------------------------------------------------------------ plumbing
    def _config_key(self):
        c = self.config
        return (c['gravity_constant'], c['softening_factor'], c['barnes_hut_theta'],
                c['heat_transfer_coefficient'], c['coefficient_of_restitution'],
                c.get('position_correction_factor', 1.0), c.get('explosion_efficiency', 0.1),
                c.get('thermal_damping_factor', 0.0), c.get('collision_solver_iterations', 1))

    def _sync_config(self):
        """The reference holds the config dict by reference and re-reads it every
        step (particle_system.py:430,462,555,673): push it when it changed."""
        key = self._config_key()
        if key != self._cfg_key:
            self._cfg = make_config(self.config, self.bounds, self._stage_mask, self._walk_mode, self._debug, self._overlap)
            _capi.check(self._lib, self._lib.psim_set_config(self._ctx, ctypes.byref(self._cfg)))
            self._cfg_key = key

    _SHAPES = dict(positions=(2, np.float64), velocities=(2, np.float64), accelerations=(2, np.float64),
                   masses=(1, np.float64), inverse_masses=(1, np.float64), temperatures=(1, np.float64),
                   radii=(1, np.int64))

    def _host_block(self, rows, cols, dtype):
        """(rows, cols) host array: a NumPy view of a page-locked PyTorch tensor (the tensor
        owns the memory; the device copies into and out of it by DMA)."""
        if not self._pinned:
            return None, np.empty((rows, cols), dtype=dtype)
        try:
            import torch
        except ImportError:
            # (without PyTorch the blocks are ordinary memory: copies then go through the
            # runtime's staging buffer, everything else is unchanged)
            self._pinned = False
            return None, np.empty((rows, cols), dtype=dtype)
        t = torch.empty((max(rows, 1), cols), dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True)
        return t, t.numpy()[:rows]

    def _alloc_host(self, n):
        """Backing store for n rows of every public array (allocated once: explosions only
        shrink the count, and the public arrays are then views of the leading rows)."""
        for name, (cols, dtype) in self._SHAPES.items():
            self._store[name] = self._host_block(n, cols, dtype)
        self._bind_host(n)

    def _bind_host(self, n):
        # the reference rebinds (does not resize) its arrays on explosion (:716-722)
        for name in _FIELDS:
            self._host[name] = self._store[name][1][:n]

    def _ptr(self, name):
        return self._host[name].ctypes.data_as(ctypes.c_void_p)

    def _upload(self):
        _capi.check(self._lib, self._lib.psim_upload(
            self._ctx, self.num_particles, self._ptr("positions"), self._ptr("velocities"),
            self._ptr("accelerations"), self._ptr("masses"), self._ptr("temperatures"), self._ptr("radii")))
        self._stale = False
        self._dirty = False

    def push_state(self):
        """Upload the host arrays now (they are uploaded before the next device call anyway
        once one of them was assigned or edited through editing())."""
        self._refresh()
        self._upload()

    def _sync_state(self):
        """host-side writes reach the device before it is used again"""
        if self._dirty:
            self._upload()

    def editing(self):
        """Context manager: writable views of the public arrays; uploaded on exit."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self._refresh()
            yield dict(self._host)
            self._dirty = True
            self._sync_state()
        return scope()

    def _refresh(self):
        if not self._stale:
            return
        n = int(self._lib.psim_num_particles(self._ctx))
        if n != self._host["positions"].shape[0]:
            self._bind_host(n)
        _capi.check(self._lib, self._lib.psim_download(
            self._ctx, self._ptr("positions"), self._ptr("velocities"), self._ptr("accelerations"),
            self._ptr("masses"), self._ptr("inverse_masses"), self._ptr("temperatures"), self._ptr("radii")))
        self._stale = False

    # ------------------------------------------------------------------ checkpoint (the reference has none)
    def save_checkpoint(self, path):
        """Full state + configuration as one .npz: a run restored from it continues
        bit-identically (the step holds no other state between calls)."""
        import json
        self._sync_state()
        self._refresh()
        h = self._host
        np.savez(path, positions=h["positions"], velocities=h["velocities"], accelerations=h["accelerations"],
                 masses=h["masses"], temperatures=h["temperatures"], bounds=np.asarray(self.bounds, dtype=np.float64),
                 config=np.array(json.dumps(self.config)),
                 collision_totals=np.array([self.pe_gain_collisions, self.ke_loss_collisions,
                                            self.heat_generated_collisions]))

    @classmethod
    def from_checkpoint(cls, path, **kwargs):
        import json
        z = np.load(path, allow_pickle=False)
        config = json.loads(str(z["config"]))
        n = z["positions"].shape[0]
        bounds = tuple(float(b) for b in z["bounds"])
        if all(b == int(b) for b in bounds):
            bounds = tuple(int(b) for b in bounds)
        state = {k: z[k] for k in ("positions", "velocities", "accelerations", "masses", "temperatures")}
        ps = cls(n, config, None, bounds, state=state, **kwargs)
        ps.pe_gain_collisions, ps.ke_loss_collisions, ps.heat_generated_collisions = (float(v) for v in z["collision_totals"])
        return ps

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.psim_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference surface
    def _calculate_gravity(self):
        """particle_system.py:409-437 (main.py:151 primes a(0) with it)."""
        self._sync_config()
        self._sync_state()
        _capi.check(self._lib, self._lib.psim_prime_gravity(self._ctx))
        self._stale = True

    def update(self, steps: int = 1):
        """particle_system.py:518-563.  `steps` > 1 runs several updates in one call."""
        self._sync_config()
        self._sync_state()
        _capi.check(self._lib, self._lib.psim_step(self._ctx, int(steps)))
        self._stale = True
        self._stepped = True
        t = (ctypes.c_double * 3)()
        _capi.check(self._lib, self._lib.psim_collision_stats(self._ctx, t))
        self.pe_gain_collisions, self.ke_loss_collisions, self.heat_generated_collisions = t[0], t[1], t[2]
        old = self.num_particles
        self.num_particles = int(self._lib.psim_num_particles(self._ctx))
        self._drain_explosion_log(old)

    def _drain_explosion_log(self, old_count):
        """Log lines of particle_system.py:683-689,727-730."""
        buf = (_capi.ExplosionEvent * 256)()
        total = 0
        while True:
            k = int(self._lib.psim_explosion_events(self._ctx, buf, 256))
            if k <= 0:
                break
            total += k
            for e in buf[:k]:
                if e.neighbours == 0:
                    logger.warning(f"Particle {e.particle} exploded with no valid neighbors. Energy was lost.")
                else:
                    logger.info(f"Particle {e.particle} exploded with {e.energy:.1f} thermal energy, "
                                f"affecting {e.neighbours} neighbors.")
        if total and self.num_particles != old_count:
            logger.info(f"{old_count - self.num_particles} particle(s) removed by explosion. "
                        f"New count: {self.num_particles}.")

    def _energies(self):
        self._sync_state()
        out = (ctypes.c_double * 5)()
        _capi.check(self._lib, self._lib.psim_energies(self._ctx, out))
        return list(out)

    def get_total_kinetic_energy(self):
        return self._energies()[0]

    def get_total_potential_energy(self):
        return self._energies()[1]

    def get_total_thermal_energy(self):
        return self._energies()[2]

    def get_energies(self):
        """(KE, PE, TE) in one call."""
        return tuple(self._energies()[:3])

    def get_total_momentum(self):
        e = self._energies()
        return e[3], e[4]

    # ------------------------------------------------------------------ diagnostics
    def counts(self):
        self._sync_state()
        out = (ctypes.c_int64 * 8)()
        _capi.check(self._lib, self._lib.psim_counts(self._ctx, out))
        keys = ("particles", "tree_nodes", "candidates", "contacts", "exploded", "internal_nodes",
                "solver_iterations", "solver_rounds")
        return dict(zip(keys, out))

    def walk_counts(self):
        self._sync_state()
        out = (ctypes.c_int64 * 4)()
        _capi.check(self._lib, self._lib.psim_walk_counts(self._ctx, out))
        self.last_walk_trips = (int(out[2]), int(out[3]))
        return int(out[0]), int(out[1])

    def walk_profile(self):
        """lane statistics of the walk's loop on the current tree (psim_walk_profile)"""
        self._sync_state()
        out = (ctypes.c_int64 * 16)()
        _capi.check(self._lib, self._lib.psim_walk_profile(self._ctx, out))
        keys = ("visits", "interactions", "trips", "force_trips", "uniform_accept_trips", "uniform_open_trips",
                "box_accept_trips", "box_open_trips", "full_trips")
        return dict(zip(keys, [int(v) for v in out]))

    def time_walk(self, reps=3, slots_per_simd=0.0):
        """Average milliseconds of the default walk kernel on the current tree with the
        device to itself (0 = whole grid at once; else that many resident waves per SIMD)."""
        self._sync_state()
        ms = ctypes.c_double(0.0)
        _capi.check(self._lib, self._lib.psim_time_walk(self._ctx, int(reps), float(slots_per_simd), ctypes.byref(ms)))
        return ms.value

    def brute_force_accelerations(self):
        """All-pairs softened gravity of the current state (O(n^2) cross-check).
        Returns (accelerations (n,2), kernel milliseconds)."""
        self._sync_state()
        out = np.empty((self.num_particles, 2))
        ms = ctypes.c_double(0.0)
        _capi.check(self._lib, self._lib.psim_brute_force_gravity(
            self._ctx, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ms)))
        return out, ms.value

    def render_feed(self):
        """What draw() needs, computed on the device (particle_system.py:587-624 up to the
        rasteriser call): (xy int32 (n,2) = int(position), rgba uint8 (n,4))."""
        self._sync_state()
        n = self.num_particles
        if self._feed is None or self._feed[0][1].shape[0] < n:
            self._feed = (self._host_block(n, 2, np.int32), self._host_block(n, 4, np.uint8))
        xy, rgba = self._feed[0][1][:n], self._feed[1][1][:n]
        if n:
            _capi.check(self._lib, self._lib.psim_render_feed(
                self._ctx, xy.ctypes.data_as(ctypes.c_void_p), rgba.ctypes.data_as(ctypes.c_void_p)))
        return xy, rgba

    def exact_potential_energy(self):
        """-G sum_{i<j} m_i m_j / sqrt(d^2 + s) over all pairs (O(n^2) diagnostic; the
        reference's own potential energy pairs sibling cells only)."""
        self._sync_state()
        out = ctypes.c_double(0.0)
        _capi.check(self._lib, self._lib.psim_pair_potential_energy(self._ctx, ctypes.byref(out)))
        return out.value

    def particle_updates(self):
        """particles alive after each step, summed over the steps of the last update() call"""
        return int(self._lib.psim_particle_updates(self._ctx))

    def stage_times_ms(self):
        out = (ctypes.c_double * 8)()
        _capi.check(self._lib, self._lib.psim_stage_times(self._ctx, out))
        keys = ("drift_keys", "grid", "tree", "gravity", "kick", "collisions", "conduction", "walls_explosions")
        return dict(zip(keys, out))

    _DEBUG_DTYPES = {
        "key": (np.uint64, ()), "perm": (np.uint32, ()), "L": (np.int8, ()), "owned": (np.uint32, ()),
        "base": (np.uint32, ()), "rec": (np.dtype([("mass", "f8"), ("x", "f8"), ("y", "f8"),
                                                   ("skip", "u4"), ("tag", "u4")]), ()),
        "node_first": (np.uint32, ()), "node_last": (np.uint32, ()), "node_depth": (np.uint32, ()),
        "depth_offsets": (np.int32, ()), "grid_offsets": (np.int32, ()), "grid_indices": (np.int32, ()),
        "cell_key": (np.uint32, ()), "contact_ij": (np.int32, (2,)), "contact_vals": (np.float64, (3,)),
        "delta_temps": (np.float64, ()), "exploded": (np.int32, ()), "tree_path": (np.uint32, ()),
    }

    @property
    def grid_offsets(self):
        """particle_system.py:393 (read-only copy): CSR offsets of the last step's grid, zeros
        before the first step as in the reference"""
        cells = self.grid_width * self.grid_height
        a = self.debug_array("grid_offsets") if self._stepped else np.zeros(cells + 1, dtype=np.int32)
        a.flags.writeable = False
        return a

    @property
    def grid_indices(self):
        """particle_system.py:395 (read-only copy): particle indices sorted by cell, of the last
        step's grid (it holds the particles the step started with)"""
        a = self.debug_array("grid_indices") if self._stepped else np.zeros(self.num_particles, dtype=np.int32)
        a.flags.writeable = False
        return a

    def debug_array(self, name, holes=False):
        self._sync_state()
        dtype, tail = self._DEBUG_DTYPES[name]
        cnt = int(self._lib.psim_debug_size(self._ctx, name.encode()))
        if cnt < 0:
            raise KeyError(name)
        out = np.empty((cnt,) + tail, dtype=dtype)
        _capi.check(self._lib, self._lib.psim_debug_fetch(self._ctx, name.encode(),
                                                          out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        if name == "contact_vals":
            return out[self.debug_array("contact_ij", holes=True)[:, 0] >= 0]
        if name == "contact_ij" and not holes:
            # rows of -1: pairs of the solver's list that did not touch when their turn came, or
            # (one rank's band of a split step) contacts another rank counts
            return out[out[:, 0] >= 0]
        return out

    def canonical_tree(self):
        """The tree in the order-independent form the golden fixtures use:
        internal nodes as (depth, path) sorted, with mass and COM; occupied leaves
        as (depth, path, particle) sorted by particle."""
        key = self.debug_array("key")
        perm = self.debug_array("perm")
        L = self.debug_array("L").astype(np.int64)
        rec = self.debug_array("rec")
        first = self.debug_array("node_first").astype(np.int64)
        n = len(key)
        m = len(first)
        ids = np.arange(m, dtype=np.int64)
        r = rec[first + ids] if m else rec[:0]
        # particle s owns the internal nodes of depths L[s-1]+1 .. L[s]; node ids run base[s] ..
        base = self.debug_array("base").astype(np.int64)
        lp_all = np.concatenate(([-1], L[:-1])) if n else L
        depth = (lp_all[first] + 1 + (ids - base[first])) if m else np.zeros(0, np.int64)
        shift = (64 - 2 * depth).astype(np.uint64)
        path = np.where(depth > 0, key[first] >> np.minimum(shift, np.uint64(63)), np.uint64(0)) if m else np.zeros(0, np.uint64)
        order = np.lexsort((path, depth)) if m else np.zeros(0, np.int64)
        # leaf depth = 1 + deepest internal ancestor = 1 + max shared levels with a neighbour
        lp = np.concatenate(([-1], L[:-1])) if n else L
        ld = (np.maximum(L, lp) + 1) if n > 1 else np.zeros(n, np.int64)
        lshift = (64 - 2 * ld).astype(np.uint64)
        lpath = np.where(ld > 0, np.where(lshift >= 64, np.uint64(0), key >> np.minimum(lshift, np.uint64(63))), np.uint64(0))
        lorder = np.argsort(perm, kind="stable")
        return dict(
            tree_nodes=(1 + 4 * m) if n else 0, tree_max_depth=int(ld.max()) if n else 0,
            int_depth=depth[order].astype(np.int32), int_path=path[order].astype(np.uint64),
            int_mass=r["mass"][order], int_comx=r["x"][order], int_comy=r["y"][order],
            leaf_depth=ld[lorder].astype(np.int32), leaf_path=lpath[lorder].astype(np.uint64),
            leaf_particle=perm[lorder].astype(np.int32))

    # ------------------------------------------------------------------ rendering (main.py:95,106)
    def draw(self, screen, is_glow_pass: bool):
        """particle_system.py:587-624: one filled circle per particle, coloured by
        temperature through the keyframe ramp of constants.py:37-45."""
        import pygame
        xy, rgba = self.render_feed()
        # colours, centres and radii become plain Python lists in three vectorised
        # conversions; the loop only issues the rasteriser calls
        colours = [tuple(c) for c in (rgba if is_glow_pass else rgba[:, :3]).tolist()]
        centres = [tuple(c) for c in xy.tolist()]
        radii = self.radii[:, 0].tolist()
        circle = pygame.draw.circle
        for colour, centre, radius in zip(colours, centres, radii):
            circle(screen, colour, centre, radius)


def _make_property(name):
    def getter(self):
        self._refresh()
        view = self._host[name].view()
        view.flags.writeable = False      # an in-place edit would never reach the device: make it raise
        return view

    def setter(self, value):
        # the reference's attributes are plain arrays: an assignment is seen by the next
        # update().  Here it is copied into the page-locked block and uploaded before the
        # next device call.
        self._refresh()
        value = np.asarray(value)
        if value.shape != self._host[name].shape:
            raise ValueError("%s: expected shape %s, got %s" % (name, self._host[name].shape, value.shape))
        np.copyto(self._host[name], value)
        if name == "masses":
            np.copyto(self._host["inverse_masses"], 1.0 / self._host["masses"])
        self._dirty = True
    return property(getter, setter)


for _name in _FIELDS:
    setattr(ParticleSystem, _name, _make_property(_name))

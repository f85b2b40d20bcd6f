"""ctypes front end of the C oracle (oracle/psim_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

`OracleSystem` has the constructor and the public surface of the reference's
`ParticleSystem` (particle_system.py:339-804): same three RNG draws in the same
order, same derived grid geometry, `update()`, `_calculate_gravity()`, the three
energy getters and the three collision-energy attributes.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpsim_oracle.so")
_SO_FAST = os.path.join(_HERE, "libpsim_oracle_fast.so")
_SRC = os.path.join(_HERE, "psim_oracle.c")

COLOR_MAX_TEMP = 1.0   # constants.py:33

STAGE_ALL = 31
STAGE_GRAVITY_ONLY = 0


def build(force=False, fast_stages=False):
    """Compile the oracle with gcc (strict FP64, no contraction).  fast_stages: the second
    build, used for CPU timing only, in which the functions restating the reference's six
    `fastmath=True` kernels carry -ffast-math (the strict ones stay strict)."""
    so = _SO_FAST if fast_stages else _SO
    if (not force and os.path.exists(so)
            and os.path.getmtime(so) >= os.path.getmtime(_SRC)):
        return so
    extra = ["-DORC_FAST_STAGES"] if fast_stages else []
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp"] + extra +
                          ["-shared", "-fPIC", "-o", so, _SRC, "-lm"])
    return so


_libs = {}


def lib(fast_stages=False):
    if fast_stages not in _libs:
        so = _SO_FAST if fast_stages else _SO
        if not os.path.exists(so):
            build(fast_stages=fast_stages)
        L = ctypes.CDLL(so)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_ptr.restype = ctypes.c_void_p
        L.orc_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_get_i.restype = ctypes.c_int64
        L.orc_get_i.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_get_collision_totals.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        for name in ("orc_build_grid", "orc_collisions", "orc_conduction", "orc_walls",
                     "orc_explosions"):
            getattr(L, name).argtypes = [ctypes.c_void_p]
            getattr(L, name).restype = None
        L.orc_build_tree.argtypes = [ctypes.c_void_p]
        L.orc_build_tree.restype = ctypes.c_int
        L.orc_gravity.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                  ctypes.POINTER(ctypes.c_int64)]
        L.orc_potential_energy.argtypes = [ctypes.c_void_p]
        L.orc_potential_energy.restype = ctypes.c_double
        L.orc_update.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_update.restype = ctypes.c_int
        L.orc_enable_contact_log.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.orc_set_fast_pow.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _libs[fast_stages] = L
    return _libs[fast_stages]


def grid_geometry(config, bounds):
    """particle_system.py:381-390."""
    max_radius = np.sqrt(config['max_mass']).astype(int)
    base_cell_size = max_radius * 2
    multiplier = config.get('grid_cell_size_multiplier', 1.0)
    cell_size = int(base_cell_size * multiplier)
    if cell_size == 0:
        cell_size = 10
    b = np.array(bounds)
    gw = int(np.ceil(b[0] / cell_size))
    gh = int(np.ceil(b[1] / cell_size))
    return cell_size, gw, gh


def config_vector(config, bounds):
    return np.array([
        config['min_mass'], config['max_mass'], config['gravity_constant'],
        config['softening_factor'], config['barnes_hut_theta'],
        config['heat_transfer_coefficient'], config['coefficient_of_restitution'],
        config.get('position_correction_factor', 1.0),
        config.get('explosion_efficiency', 0.1),
        config.get('thermal_damping_factor', 0.0),
        COLOR_MAX_TEMP, float(bounds[0]), float(bounds[1])], dtype=np.float64)


def _view(addr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OracleSystem:
    def __init__(self, num_particles, config, rng, bounds, *, state=None, stage_mask=STAGE_ALL, fast_stages=False):
        """state: optional dict(positions, velocities, masses, temperatures[, accelerations])
        used instead of the RNG draws (synthetic benchmark clouds).  fast_stages: bind the
        timing build (see build())."""
        self._L = lib(fast_stages)
        self.num_particles = int(num_particles)
        self.config = config
        self.bounds = np.array(bounds)
        self.stage_mask = stage_mask
        n = self.num_particles
        if state is None:
            positions = rng.random((n, 2)) * self.bounds                                   # :370
            masses = rng.uniform(config['min_mass'], config['max_mass'], (n, 1))           # :373
            temperatures = rng.uniform(config['min_temp'], config['max_temp'], (n, 1))     # :375
            velocities = np.zeros((n, 2))
            accelerations = np.zeros((n, 2))
        else:
            positions = np.asarray(state['positions'], dtype=np.float64).reshape(n, 2)
            masses = np.asarray(state['masses'], dtype=np.float64).reshape(n, 1)
            temperatures = np.asarray(state['temperatures'], dtype=np.float64).reshape(n, 1)
            velocities = np.asarray(state.get('velocities', np.zeros((n, 2))), dtype=np.float64).reshape(n, 2)
            accelerations = np.asarray(state.get('accelerations', np.zeros((n, 2))), dtype=np.float64).reshape(n, 2)
        self.cell_size, self.grid_width, self.grid_height = grid_geometry(config, bounds)
        L = self._L
        cfg = config_vector(config, bounds)
        self._h = L.orc_create(n, cfg.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               int(config.get('collision_solver_iterations', 1)),
                               self.cell_size, self.grid_width, self.grid_height)
        self._bind()
        self.positions[:] = positions
        self.velocities[:] = velocities
        self.accelerations[:] = accelerations
        self.masses[:] = masses
        self.inverse_masses[:] = 1.0 / masses                                              # :374
        self.temperatures[:] = temperatures
        self.radii[:] = np.sqrt(masses).astype(int)                                        # :376
        self.pe_gain_collisions = 0.0
        self.ke_loss_collisions = 0.0
        self.heat_generated_collisions = 0.0
        rc = L.orc_build_tree(self._h)                                                     # :403
        if rc != 0:
            raise RuntimeError("oracle: node pool overflow (coincident particles)")

    def _p(self, which):
        return self._L.orc_ptr(self._h, which)

    def _bind(self):
        n = int(self._L.orc_get_i(self._h, 0))
        self.num_particles = n
        self.positions = _view(self._p(0), (n, 2), np.float64)
        self.velocities = _view(self._p(1), (n, 2), np.float64)
        self.accelerations = _view(self._p(2), (n, 2), np.float64)
        self.masses = _view(self._p(3), (n, 1), np.float64)
        self.inverse_masses = _view(self._p(4), (n, 1), np.float64)
        self.temperatures = _view(self._p(5), (n, 1), np.float64)
        self.radii = _view(self._p(6), (n, 1), np.int64)

    def __del__(self):
        try:
            if self._h:
                self._L.orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- reference surface -------------------------------------------------
    def _calculate_gravity(self):
        v = ctypes.c_int64(0)
        i = ctypes.c_int64(0)
        self._L.orc_gravity(self._h, ctypes.byref(v), ctypes.byref(i))
        self.last_visits, self.last_interactions = v.value, i.value

    def update(self):
        rc = self._L.orc_update(self._h, self.stage_mask)
        if rc != 0:
            raise RuntimeError("oracle: node pool overflow (coincident particles)")
        t = (ctypes.c_double * 3)()
        self._L.orc_get_collision_totals(self._h, t)
        self.pe_gain_collisions, self.ke_loss_collisions, self.heat_generated_collisions = t[0], t[1], t[2]
        if int(self._L.orc_get_i(self._h, 0)) != self.num_particles:
            self._bind()

    def get_total_kinetic_energy(self):
        vel_sq = np.sum(self.velocities ** 2, axis=1, keepdims=True)       # :772-774
        return np.sum(0.5 * self.masses * vel_sq)

    def get_total_thermal_energy(self):
        return np.sum(self.masses * self.temperatures)                     # :783

    def get_total_potential_energy(self):
        return self._L.orc_potential_energy(self._h)

    # -- stage access for tests -------------------------------------------
    def enable_contact_log(self, cap):
        self._L.orc_enable_contact_log(self._h, cap)

    def set_fast_pow(self, on):
        self._L.orc_set_fast_pow(self._h, 1 if on else 0)

    def set_threads(self, k):
        """threads of the per-particle gravity loop (results do not depend on it)"""
        self._L.orc_set_threads(self._h, int(k))

    def build_grid(self):
        self._L.orc_build_grid(self._h)

    def build_tree(self):
        return self._L.orc_build_tree(self._h)

    @property
    def grid_offsets(self):
        return _view(self._p(7), (self.grid_width * self.grid_height + 1,), np.int32)

    @property
    def grid_indices(self):
        return _view(self._p(8), (int(self._L.orc_get_i(self._h, 6)),), np.int32)

    @property
    def num_active_nodes(self):
        return int(self._L.orc_get_i(self._h, 1))

    @property
    def n_candidates(self):
        return int(self._L.orc_get_i(self._h, 2))

    @property
    def n_contacts(self):
        return int(self._L.orc_get_i(self._h, 3))

    @property
    def exploded(self):
        k = int(self._L.orc_get_i(self._h, 4))
        return _view(self._p(17), (k,), np.int32).copy()

    @property
    def delta_temps(self):
        return _view(self._p(16), (int(self._L.orc_get_i(self._h, 6)),), np.float64)

    def contact_log(self):
        k = int(self._L.orc_get_i(self._h, 5))
        return (_view(self._p(18), (k, 2), np.int32).copy(),
                _view(self._p(19), (k, 3), np.float64).copy())

    def tree_arrays(self):
        m = self.num_active_nodes
        return dict(
            bounds=_view(self._p(9), (m, 4), np.float64),
            data=_view(self._p(10), (m, 4), np.float64),
            children=_view(self._p(11), (m, 4), np.int32),
            is_leaf=_view(self._p(12), (m,), np.uint8),
            particle=_view(self._p(13), (m,), np.int32),
        )

    def canonical_tree(self):
        """Same canonical form as tests/golden/make_golden.py: internal nodes
        sorted by (depth, path), occupied leaves sorted by particle."""
        t = self.tree_arrays()
        ch, leaf, pidx, data = t['children'], t['is_leaf'], t['particle'], t['data']
        internal, leaves = [], []
        stack = [(0, 0, 0)]
        max_depth = 0
        while stack:
            node, depth, path = stack.pop()
            max_depth = max(max_depth, depth)
            if leaf[node]:
                if pidx[node] != -1:
                    leaves.append((depth, path, int(pidx[node])))
            else:
                internal.append((depth, path, data[node, 0], data[node, 1], data[node, 2]))
                for q in range(4):
                    stack.append((int(ch[node, q]), depth + 1, (path << 2) | q))
        internal.sort(key=lambda r: (r[0], r[1]))
        leaves.sort(key=lambda r: r[2])
        return dict(
            tree_nodes=len(leaf), tree_max_depth=max_depth,
            int_depth=np.array([r[0] for r in internal], dtype=np.int32),
            int_path=np.array([r[1] for r in internal], dtype=np.uint64),
            int_mass=np.array([r[2] for r in internal], dtype=np.float64),
            int_comx=np.array([r[3] for r in internal], dtype=np.float64),
            int_comy=np.array([r[4] for r in internal], dtype=np.float64),
            leaf_depth=np.array([r[0] for r in leaves], dtype=np.int32),
            leaf_path=np.array([r[1] for r in leaves], dtype=np.uint64),
            leaf_particle=np.array([r[2] for r in leaves], dtype=np.int32),
        )

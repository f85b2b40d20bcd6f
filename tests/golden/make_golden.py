#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference's own source.

Runs ONLY in the build container (needs /root/reference).  The reference is
imported unmodified and executed under plain CPython: `numba` and `pygame` are
not installed, so two stand-in modules are placed in sys.modules first --
`numba.jit` as an identity decorator (the 13 kernels are plain Python/NumPy)
and an empty `pygame` (only `pygame.Surface` appears, as an annotation).
Strict IEEE FP64 throughout, which is the definition for the non-fastmath
reference functions; fastmath ones are compared with a tolerance by the tests.

Nothing of the reference's text is written to the fixtures: they hold inputs
(config, seed, bounds) and recorded outputs only.

    python tests/golden/make_golden.py [case ...]
"""
import sys
import types
import json
import os
import time

sys.dont_write_bytecode = True

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = jit
    numba.njit = jit
    pygame = types.ModuleType("pygame")
    pygame.Surface = object
    sys.modules["numba"] = numba
    sys.modules["pygame"] = pygame
    sys.path.insert(0, REF)
    import logging
    logging.getLogger("particle_sim").addHandler(logging.NullHandler())
    logging.getLogger("particle_sim").propagate = False
    import particle_system
    import quadtree
    import constants
    return particle_system, quadtree, constants


ps_mod, qt_mod, constants = _import_reference()

DEFAULT_SIM = json.load(open(os.path.join(REF, "config.json")))["simulation"]


class LogCapture:
    """Handler on the reference's own `particle_sim` logger (logger_setup.py:32): keeps
    (step, level name, message) of every record the step emits (the explosion lines of
    particle_system.py:683-689, 727-730)."""

    def __init__(self):
        import logging

        capture = self

        class H(logging.Handler):
            def emit(self, record):
                capture.records.append((capture.step, record.levelname, record.getMessage()))

        self.records = []
        self.step = 0
        self.handler = H(level=logging.DEBUG)
        self.logger = logging.getLogger("particle_sim")
        self.old_level = self.logger.level
        self.logger.setLevel(logging.DEBUG)
        self.logger.addHandler(self.handler)

    def close(self):
        self.logger.removeHandler(self.handler)
        self.logger.setLevel(self.old_level)


class Recorder:
    """Hooks the module-level kernels of the reference to record what a step did."""

    def __init__(self):
        self.contacts = []      # (i, j, pe, ke, heat) in execution order, this step
        self.candidates = 0
        self.delta_temps = None
        self.exploded = None
        self._orig_resolve = ps_mod._resolve_collision_jit
        self._orig_heat = ps_mod._handle_heat_transfer_jit
        ps_mod._resolve_collision_jit = self._resolve
        ps_mod._handle_heat_transfer_jit = self._heat

    def _resolve(self, i, j, *a):
        self.candidates += 1
        try:
            pe, ke, heat = self._orig_resolve(i, j, *a)
        except UnboundLocalError:
            # no-contact branch: numba returns a zero-initialised heat_generated
            return 0.0, 0.0, 0.0
        self.contacts.append((int(i), int(j), float(pe), float(ke), float(heat)))
        return pe, ke, heat

    def _heat(self, *a):
        self._orig_heat(*a)
        self.delta_temps = a[-1].copy()

    def reset(self):
        self.contacts = []
        self.candidates = 0
        self.delta_temps = None
        self.exploded = None


def canonical_tree(qt):
    """(depth, path) for every internal node and every occupied leaf.

    path = quadrant digits from the root packed 2 bits per level, most
    significant first (root child digit at bits [2*(depth-1)+1 : 2*(depth-1)]).
    """
    n = qt.num_active_nodes
    ch = qt.node_children[:n]
    leaf = qt.node_is_leaf[:n]
    pidx = qt.node_particle_idx[:n]
    data = qt.node_data[:n]
    internal = []   # depth, path, mass, comx, comy
    leaves = []     # depth, path, particle
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
                c = ch[node, q]
                stack.append((int(c), depth + 1, (path << 2) | q))
    internal.sort(key=lambda t: (t[0], t[1]))
    leaves.sort(key=lambda t: t[2])
    return dict(
        tree_nodes=np.int64(n),
        tree_max_depth=np.int64(max_depth),
        int_depth=np.array([t[0] for t in internal], dtype=np.int32),
        int_path=np.array([t[1] for t in internal], dtype=np.uint64),
        int_mass=np.array([t[2] for t in internal], dtype=np.float64),
        int_comx=np.array([t[3] for t in internal], dtype=np.float64),
        int_comy=np.array([t[4] for t in internal], dtype=np.float64),
        leaf_depth=np.array([t[0] for t in leaves], dtype=np.int32),
        leaf_path=np.array([t[1] for t in leaves], dtype=np.uint64),
        leaf_particle=np.array([t[2] for t in leaves], dtype=np.int32),
    )


def make_system(n, sim, seed, bounds):
    rng = np.random.default_rng(seed)
    return ps_mod.ParticleSystem(n, sim, rng, bounds)


def snapshot(prefix, system, out, with_static=False):
    out[prefix + "positions"] = system.positions.copy()
    out[prefix + "velocities"] = system.velocities.copy()
    out[prefix + "accelerations"] = system.accelerations.copy()
    out[prefix + "temperatures"] = system.temperatures.copy()
    if with_static:
        out[prefix + "masses"] = system.masses.copy()
        out[prefix + "radii"] = system.radii.copy()


def energies(system):
    return np.array([system.get_total_kinetic_energy(),
                     system.get_total_potential_energy(),
                     system.get_total_thermal_energy()], dtype=np.float64)


def run_case(name, n, sim_overrides, seed, bounds, steps, detail_steps,
             init_velocity_scale=0.0, state_steps=None):
    """Prime + `steps` updates.  Full per-stage detail is kept for steps in
    detail_steps; post-step state for steps in state_steps (default = detail)."""
    t0 = time.time()
    sim = dict(DEFAULT_SIM)
    sim.update(sim_overrides)
    rec = Recorder()
    system = make_system(n, sim, seed, bounds)
    if init_velocity_scale:
        # extra deterministic draw AFTER the reference's three draws
        rng2 = np.random.default_rng(seed + 1000)
        system.velocities[:] = rng2.standard_normal((n, 2)) * init_velocity_scale
    out = {}
    meta = dict(name=name, n=n, seed=seed, bounds=list(bounds), steps=steps, sim=sim,
                init_velocity_scale=init_velocity_scale,
                detail_steps=sorted(detail_steps))
    snapshot("init_", system, out, with_static=True)
    out["init_tree_nodes"] = np.int64(system.qtree.num_active_nodes)
    for k, v in canonical_tree(system.qtree).items():
        out["init_" + k] = v
    system._calculate_gravity()
    out["prime_accelerations"] = system.accelerations.copy()
    out["prime_energies"] = energies(system)
    state_steps = set(detail_steps) if state_steps is None else set(state_steps)
    per_step_energy = []
    per_step_counts = []
    per_step_coll = []
    ids = np.arange(n, dtype=np.int64)
    log = LogCapture()
    for step in range(1, steps + 1):
        rec.reset()
        log.step = step
        pre_n = system.num_particles
        # explosions: record which indices (pre-compaction) explode
        orig_expl = system._handle_explosions
        holder = {}

        def hooked():
            mask = system.temperatures.flatten() >= constants.COLOR_MAX_TEMP
            holder["idx"] = np.where(mask)[0].astype(np.int32)
            holder["pre_vel"] = system.velocities.copy()
            orig_expl()

        system._handle_explosions = hooked
        system.update()
        system._handle_explosions = orig_expl
        exploded = holder["idx"]
        e = energies(system)
        per_step_energy.append(e)
        per_step_counts.append((rec.candidates, len(rec.contacts),
                                system.qtree.num_active_nodes, system.num_particles,
                                len(exploded)))
        per_step_coll.append((system.pe_gain_collisions, system.ke_loss_collisions,
                              system.heat_generated_collisions))
        if len(exploded):
            keep = np.ones(pre_n, dtype=bool)
            keep[exploded] = False
            ids = ids[keep]
        p = "s%d_" % step
        if step in detail_steps:
            out[p + "grid_offsets"] = system.grid_offsets.copy()
            out[p + "grid_indices"] = system.grid_indices[:pre_n].copy()
            c = rec.contacts
            out[p + "contact_ij"] = np.array([(a, b) for a, b, *_ in c], dtype=np.int32).reshape(-1, 2)
            out[p + "contact_vals"] = np.array([t[2:] for t in c], dtype=np.float64).reshape(-1, 3)
            out[p + "delta_temps"] = rec.delta_temps
            out[p + "exploded"] = exploded
            out[p + "alive_ids"] = ids.copy()
            for k, v in canonical_tree(system.qtree).items():
                out[p + k] = v
        if step in state_steps or step in detail_steps:
            snapshot(p, system, out)
            out[p + "alive_ids"] = ids.copy()
    out["energies"] = np.array(per_step_energy, dtype=np.float64).reshape(-1, 3)
    out["counts"] = np.array(per_step_counts, dtype=np.int64).reshape(-1, 5)
    out["collision_totals"] = np.array(per_step_coll, dtype=np.float64).reshape(-1, 3)
    log.close()
    # the log records of the steps, as the reference's logger produced them
    out["log_json"] = np.frombuffer(json.dumps(log.records).encode(), dtype=np.uint8)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    ps_mod._resolve_collision_jit = rec._orig_resolve
    ps_mod._handle_heat_transfer_jit = rec._orig_heat
    print("%s: n=%d steps=%d -> %s (%.1f KB) in %.1fs; contacts/step max %d, final n %d" % (
        name, n, steps, path, os.path.getsize(path) / 1024, time.time() - t0,
        max(c[1] for c in per_step_counts), system.num_particles), flush=True)


def tree_gravity_case(name, n, seed, bounds):
    """Tree + gravity only at a larger N (one build, one traversal)."""
    t0 = time.time()
    sim = dict(DEFAULT_SIM)
    system = make_system(n, sim, seed, bounds)
    out = {}
    meta = dict(name=name, n=n, seed=seed, bounds=list(bounds), steps=0, sim=sim)
    snapshot("init_", system, out, with_static=True)
    tree = canonical_tree(system.qtree)
    # keep the fixture small: topology in full, COM only as a digest sample
    for k, v in tree.items():
        out["init_" + k] = v
    system._calculate_gravity()
    out["prime_accelerations"] = system.accelerations.copy()
    out["prime_energies"] = energies(system)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: n=%d -> %s (%.1f KB) in %.1fs" % (name, n, path, os.path.getsize(path) / 1024,
                                                  time.time() - t0), flush=True)


CASES = {
    # tiny, every step in detail
    "tiny64": lambda: run_case("tiny64", 64, {}, 7, (300, 200), 10, range(1, 11)),
    # dense: many contacts and contact chains per step, 2 solver iterations
    "dense256": lambda: run_case("dense256", 256, {"collision_solver_iterations": 2,
                                                   "grid_cell_size_multiplier": 2.0},
                                 11, (220, 160), 20, range(1, 21)),
    # dense with the default (large) cells and one solver iteration
    "dense400": lambda: run_case("dense400", 400, {}, 13, (400, 240), 12, range(1, 13)),
    # hot start: explosions fire from step 1 on
    "hot300": lambda: run_case("hot300", 300, {"min_temp": 0.85, "max_temp": 1.05},
                               17, (500, 400), 6, range(1, 7)),
    # nearly everything explodes at once: exploders whose neighbours all explode too take the
    # warning line (particle_system.py:682-684).  (An exploder with NO particle at all in its
    # 3x3 cells makes the reference raise IndexError at :680 -- an empty index array is float64 --
    # so that case has no reference output to record.)
    "hotall60": lambda: run_case("hotall60", 60, {"min_temp": 0.99, "max_temp": 1.08},
                                 29, (320, 240), 2, range(1, 3)),
    # fast particles: leave the domain between drift and the wall clamp
    "fast200": lambda: run_case("fast200", 200, {}, 19, (320, 200), 8, range(1, 9),
                                init_velocity_scale=25.0),
    # the reference's own default run (config.json verbatim), 100 steps
    "default2000": lambda: run_case("default2000", 2000, {}, 42, (2000, 1000), 100,
                                    [1, 2], state_steps=[1, 2, 10, 50, 100]),
    # tree + gravity at a larger N
    "tree20k": lambda: tree_gravity_case("tree20k", 20000, 23, (6000, 3000)),
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for nm in names:
        CASES[nm]()

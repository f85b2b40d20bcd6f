"""Host logic on CPU: the C ABI library loads and exports every symbol that
include/psim.h declares; the headless runner mirrors main.py's bookkeeping;
config handling and error behaviour of the binding."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest

from psim_amd import _capi
from psim_amd.particle_system import ParticleSystem, grid_geometry, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTLOOP_DIR = os.path.join(ROOT, "tests", "hostloop")
HOSTLOOP = os.path.join(HOSTLOOP_DIR, "libpsim_hostloop.so")
SIM = json.load(open(os.path.join(ROOT, "particle-simulator_amd", "config.json")))["simulation"]


@pytest.fixture(scope="module", autouse=True)
def build_hostloop():
    subprocess.check_call(["make", "-s", "-C", HOSTLOOP_DIR])


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "psim.h")).read()
    return sorted(set(re.findall(r"\b(psim_[a-z_]+)\s*\(", text)))


def test_hip_library_exports_declared_abi():
    """no compute calls: only dlopen + symbol lookup (works without a GPU)"""
    assert os.path.exists(_capi.HIP_LIBRARY), "build libpsim_hip.so first (make -C particle-simulator_amd/csrc)"
    L = ctypes.CDLL(_capi.HIP_LIBRARY)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(L, name), name
    assert set(_capi.EXPORTS) == set(names)
    assert b"HIP gfx950" in _capi.load().psim_version()


def test_missing_library_raises():
    with pytest.raises(ImportError):
        _capi.load("/nonexistent/libpsim_hip.so")


def test_grid_geometry_matches_reference_defaults():
    # particle_system.py:381-390 at the default config: cell 80, 25 x 13 cells
    assert grid_geometry(SIM, (2000, 1000)) == (80, 25, 13)
    c = make_config(SIM, (2000, 1000))
    assert (c.cell_size, c.grid_w, c.grid_h, c.solver_iterations) == (80, 25, 13, 1)
    assert c.explode_temp == 1.0 and c.pos_correct == 1.05
    sparse = {k: v for k, v in SIM.items() if k not in ("position_correction_factor", "explosion_efficiency",
                                                        "thermal_damping_factor", "collision_solver_iterations",
                                                        "grid_cell_size_multiplier")}
    c = make_config(sparse, (2000, 1000))           # optional keys fall back to the reference's defaults
    assert (c.pos_correct, c.explosion_eff, c.damping, c.solver_iterations, c.cell_size) == (1.0, 0.1, 0.0, 1, 8)


def test_bad_arguments_are_value_errors():
    L = _capi.load(HOSTLOOP)
    cfg = make_config(SIM, (2000, 1000))
    cfg.grid_w = 0
    ctx = ctypes.c_void_p()
    rc = L.psim_create(ctypes.byref(cfg), 0, ctypes.byref(ctx))
    assert rc == -1 and b"grid" in L.psim_last_error()
    with pytest.raises(ValueError):
        _capi.check(L, rc)
    assert L.psim_step(None, 1) == -1


def test_staged_entry_points_reject_bad_ranges_and_work_by_hand():
    """the multi-GPU stage calls driven directly through the C ABI, two bands solved one after
    the other by hand: bad ranges are PSIM_ERR_INVALID, a block that is too small reports what
    it needs and is marked incomplete, calls out of order are errors, and the two bands'
    blocks applied together give the unsharded step"""
    n, bounds = 600, (420, 300)
    ref = ParticleSystem(n, dict(SIM), np.random.default_rng(4), bounds, library=HOSTLOOP, walk_mode=1, debug=True)
    s = ParticleSystem(n, dict(SIM), np.random.default_rng(4), bounds, library=HOSTLOOP, walk_mode=1, debug=True)
    ref._calculate_gravity(); s._calculate_gravity()
    ref.update()
    L, ctx = s._lib, s._ctx
    gh = s.grid_height
    assert gh == 4
    acc = np.zeros((n, 2))
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert L.psim_stage_kick(ctx, vp(acc)) == -2                       # not inside a staged step
    assert L.psim_stage_build(ctx) == 0
    assert L.psim_stage_gravity(ctx, 5, 3, vp(acc)) == -1              # lo > hi
    assert L.psim_stage_gravity(ctx, 0, n, vp(acc)) == 0
    assert L.psim_stage_slab_positions(ctx, 0, gh + 1, 0, 2) == -1     # rows beyond the grid
    assert L.psim_stage_slab_positions(ctx, 1, gh, 0, 2) == -1         # own rows outside the band
    assert L.psim_stage_finish(ctx) == -2                              # nothing applied yet
    assert L.psim_stage_kick(ctx, vp(acc)) == 0
    words = 1 << 16
    blocks = np.zeros(2 * words, dtype=np.uint64)
    need = ctypes.c_int64(0)
    for r, (own_lo, own_hi) in enumerate([(0, 2), (2, gh)]):
        assert L.psim_stage_slab_positions(ctx, 0, gh, own_lo, own_hi) == 0     # (two halo rows cover this small grid)
        assert L.psim_stage_slab_finish(ctx) == 0
        block = blocks[r * words:(r + 1) * words]
        assert L.psim_stage_slab_export(ctx, vp(block), 8, ctypes.byref(need)) == -1      # smaller than a header
        assert L.psim_stage_slab_export(ctx, vp(block), 32, ctypes.byref(need)) == 0
        assert need.value > 32 and block[15] == 0 and block[11] == need.value             # incomplete: header only
        assert L.psim_stage_slab_export(ctx, vp(block), words, ctypes.byref(need)) == 0
        assert block[15] == 1 and block[0] > 0                                           # touched particles of the own rows
    status = ctypes.c_int32(-1)
    own = (ctypes.c_int64 * 2)()
    assert L.psim_stage_apply(ctx, vp(blocks), 2, words, ctypes.byref(status), own, ctypes.byref(need)) == 0
    assert status.value == 0 and own[0] + own[1] == n
    assert L.psim_stage_finish(ctx) == 0
    s._stale = True
    cs, cr = s.counts(), ref.counts()
    for k in ("particles", "tree_nodes", "candidates", "contacts", "exploded"):
        assert cs[k] == cr[k], k
    for name in ("positions", "velocities", "temperatures"):
        np.testing.assert_array_equal(getattr(s, name), getattr(ref, name), err_msg=name)
    ms = ctypes.c_double(0.0)
    assert L.psim_time_walk(ctx, 0, 0.0, ctypes.byref(ms)) == -1      # reps < 1
    assert L.psim_time_walk(ctx, 1, 9.0, ctypes.byref(ms)) == -1      # more slots than a SIMD has


def test_empty_and_single_particle_systems():
    for n in (0, 1, 2):
        s = ParticleSystem(n, dict(SIM), np.random.default_rng(1), (200, 100), library=HOSTLOOP, walk_mode=1)
        s._calculate_gravity()
        s.update(3)
        assert s.num_particles == n
        assert s.positions.shape == (n, 2) and s.radii.dtype == np.int64
        assert np.isfinite(s.get_energies()).all()
        s.close()


def test_config_is_reread_each_step():
    """the reference holds the config dict by reference (particle_system.py:430,462,555)"""
    cfg = dict(SIM)
    a = ParticleSystem(300, cfg, np.random.default_rng(3), (400, 300), library=HOSTLOOP, walk_mode=1)
    b = ParticleSystem(300, dict(SIM, gravity_constant=0.05), np.random.default_rng(3), (400, 300),
                       library=HOSTLOOP, walk_mode=1)
    a._calculate_gravity(); b._calculate_gravity()
    assert not np.array_equal(a.accelerations, b.accelerations)
    cfg['gravity_constant'] = 0.05          # mutate the dict the system holds
    a._calculate_gravity()
    np.testing.assert_array_equal(a.accelerations, b.accelerations)


def test_headless_runner_logs_like_main(tmp_path, monkeypatch):
    from psim_amd import headless
    monkeypatch.chdir(tmp_path)
    config = json.load(open(os.path.join(ROOT, "particle-simulator_amd", "config.json")))
    config["simulation"]["particle_count"] = 400
    headless.setup_logging(config)
    system = headless.run(config, ticks=101, bounds=(800, 400), library=HOSTLOOP)
    assert system.num_particles == 400
    log = open(tmp_path / "runs" / "sim_run_001" / "simulation.log").read()
    lines = [l for l in log.splitlines() if "Tick=" in l]
    assert len(lines) == 2 and "Tick=0," in lines[0] and "Tick=100," in lines[1]
    for key in ("Kinetic=", "Thermal=", "Potential=", "TotalSystemEnergy=", "Delta_E=", "AccumulatedKE_Loss=",
                "AccumulatedPE_Gain=", "AccumulatedHeat="):
        assert key in lines[1]
    assert "particle-updates/s" in log and "momentum=" in log


def test_render_feed_and_pair_potential_hostloop():
    import parity
    parity.check_render_feed_and_pair_potential(HOSTLOOP)


def test_checkpoint_restore_continues_bit_identically(tmp_path):
    """6 steps in one go == 3 steps, checkpoint, restore into a new system, 3 steps"""
    sim = dict(SIM, particle_count=300, min_temp=0.5, max_temp=0.99)
    bounds = (300, 200)
    a = ParticleSystem(300, dict(sim), np.random.default_rng(8), bounds, library=HOSTLOOP)
    a._calculate_gravity()
    a.update(6)
    b = ParticleSystem(300, dict(sim), np.random.default_rng(8), bounds, library=HOSTLOOP)
    b._calculate_gravity()
    b.update(3)
    path = str(tmp_path / "ckpt.npz")
    b.save_checkpoint(path)
    b.close()
    c = ParticleSystem.from_checkpoint(path, library=HOSTLOOP)
    c.update(3)
    assert c.num_particles == a.num_particles
    for name in ("positions", "velocities", "accelerations", "temperatures", "masses", "radii"):
        np.testing.assert_array_equal(getattr(c, name), getattr(a, name), err_msg=name)


def test_drop_in_launcher_rebinds_the_reference_import(tmp_path):
    """`python -m psim_amd.drop_in main.py`: a script that imports ParticleSystem the way the
    reference's main.py does (main.py:9) gets this build's class, with no edit to the script;
    the script's own sibling modules still resolve"""
    (tmp_path / "constants.py").write_text("WIDTH, HEIGHT = 640, 480\n")
    (tmp_path / "main.py").write_text(
        "import constants\n"
        "from particle_system import ParticleSystem\n"
        "result = (ParticleSystem.__module__, constants.WIDTH, __name__)\n")
    from psim_amd import drop_in
    import sys
    saved = sys.modules.pop("particle_system", None), sys.modules.pop("constants", None)
    try:
        ns = drop_in.run(str(tmp_path / "main.py"))
    finally:
        for name, mod in zip(("particle_system", "constants"), saved):
            sys.modules.pop(name, None)
            if mod is not None:
                sys.modules[name] = mod
        if str(tmp_path) in sys.path:
            sys.path.remove(str(tmp_path))
    assert ns["result"] == ("psim_amd.particle_system", 640, "__main__")

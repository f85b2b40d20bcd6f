"""Headless twin of the reference's entry point (main.py:111-167): same config
file, same master seed, same logger, same priming call, same per-tick energy
bookkeeping and the same log line every 100 ticks (main.py:58-84) -- without the
Pygame window.

    python -m psim_amd.headless [--config config.json] [--ticks 1000] [--bounds W H]
"""
import argparse
import json
import logging
import os
import time

import numpy as np

from . import constants
from .particle_system import ParticleSystem

logger = logging.getLogger("particle_sim")


def setup_logging(config, log_root="runs"):
    """Dedicated non-propagating 'particle_sim' logger writing to the console and to
    runs/<run_id>/simulation.log (what logger_setup.py:7-61 configures)."""
    run_id = config['run_id']
    log_config = config['logging']
    logger.setLevel(log_config['level'])
    logger.propagate = False
    log_dir = os.path.join(log_root, run_id)
    os.makedirs(log_dir, exist_ok=True)
    log_file = os.path.join(log_dir, 'simulation.log')
    formatter = logging.Formatter(log_config['format'])
    for h in list(logger.handlers):
        logger.removeHandler(h)
    for h in (logging.FileHandler(log_file), logging.StreamHandler()):
        h.setFormatter(formatter)
        logger.addHandler(h)
    logger.info(f"Logging initialized. Run ID: {run_id}. Log file: {log_file}")


def run(config, ticks, bounds=(constants.WIDTH, constants.HEIGHT), device=0, log_every=100, library=None,
        restore=None, save=None):
    """restore / save: .npz checkpoint paths (no counterpart in the reference).  A restored
    run carries its accelerations, so it is not primed again."""
    sim_config = config['simulation']
    if restore:
        system = ParticleSystem.from_checkpoint(restore, device=device, library=library)
        logger.info(f"Restored {system.num_particles} particles from {restore}")
    else:
        rng = np.random.default_rng(config['master_seed'])
        logger.info(f"Master RNG initialized with seed: {config['master_seed']}")
        system = ParticleSystem(num_particles=sim_config['particle_count'], config=sim_config, rng=rng,
                                bounds=bounds, device=device, library=library)
        system._calculate_gravity()          # main.py:151
    accumulated = [0.0, 0.0, 0.0]
    last_logged_energy = 0.0
    first_log = True
    t0 = time.perf_counter()
    updates = 0
    for tick in range(ticks):
        system.update()
        updates += system.num_particles
        accumulated[0] += system.pe_gain_collisions
        accumulated[1] += system.ke_loss_collisions
        accumulated[2] += system.heat_generated_collisions
        if tick % log_every == 0:
            ke, pe, te = system.get_energies()
            total_energy = ke + te + pe
            if first_log:
                delta_e = 0.0
                first_log = False
            else:
                delta_e = total_energy - last_logged_energy
            last_logged_energy = total_energy
            logger.debug(
                f"Tick={tick}, Kinetic={ke:.2f}, Thermal={te:.2f}, Potential={pe:.2f}, "
                f"TotalSystemEnergy={total_energy:.2f}, Delta_E={delta_e:+.2f}, "
                f"AccumulatedKE_Loss={accumulated[1]:+.2f}, AccumulatedPE_Gain={accumulated[0]:+.2f}, "
                f"AccumulatedHeat={accumulated[2]:+.2f}")
            accumulated = [0.0, 0.0, 0.0]
    dt = time.perf_counter() - t0
    px, py = system.get_total_momentum()
    logger.info(f"{ticks} ticks in {dt:.3f} s: {updates / dt:.4g} particle-updates/s; "
                f"momentum=({px:+.6g}, {py:+.6g}); particles={system.num_particles}")
    if save:
        system.save_checkpoint(save)
        logger.info(f"Checkpoint written to {save}")
    return system


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config.json")
    ap.add_argument("--ticks", type=int, default=1000)
    ap.add_argument("--bounds", type=float, nargs=2, default=[constants.WIDTH, constants.HEIGHT])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--restore", default=None, help=".npz checkpoint to continue from")
    ap.add_argument("--save", default=None, help="write a .npz checkpoint after the last tick")
    args = ap.parse_args()
    with open(args.config, 'r') as f:
        config = json.load(f)
    setup_logging(config)
    logger.info("Application starting...")
    logger.info(f"Loaded configuration: {config}")
    run(config, args.ticks, tuple(args.bounds), args.device, restore=args.restore, save=args.save)
    logger.info("Application shutting down.")


if __name__ == "__main__":
    main()

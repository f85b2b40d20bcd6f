"""Run the reference's own entry script, unedited, on this build:

    cd /path/to/Particle-Simulator && python -m psim_amd.drop_in main.py

The reference's `main.py` does `from particle_system import ParticleSystem`
(main.py:9).  This launcher registers this package's `ParticleSystem` module
under that name before the script runs, so the script's import resolves to the
HIP-backed class; everything else the script imports (`constants`,
`logger_setup`, Pygame) and reads (`config.json`, relative to the working
directory) stays the reference's own.
"""
import os
import runpy
import sys

from . import particle_system as _hip_particle_system


def run(script, argv=()):
    script = os.path.abspath(script)
    sys.modules["particle_system"] = _hip_particle_system
    script_dir = os.path.dirname(script)
    if script_dir not in sys.path:
        sys.path.insert(0, script_dir)          # what `python main.py` would have put there
    old_argv = sys.argv
    sys.argv = [script] + list(argv)
    try:
        return runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old_argv


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m psim_amd.drop_in <script.py> [args...]")
    run(sys.argv[1], sys.argv[2:])


if __name__ == "__main__":
    main()

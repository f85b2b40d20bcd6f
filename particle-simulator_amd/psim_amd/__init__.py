"""MI355X-native physics step for NeoLorenzo/Particle-Simulator (HIP, gfx950).

Importing this package loads the HIP extension; it raises ImportError when
libpsim_hip.so is missing (there is no CPU fallback)."""
from . import _capi
from .particle_system import ParticleSystem, grid_geometry, make_config

__all__ = ["ParticleSystem", "grid_geometry", "make_config", "_capi"]

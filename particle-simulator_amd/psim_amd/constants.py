"""Fixed values the step and the renderer read (the reference keeps the same
names in constants.py:15-52)."""
import numpy as np

WIDTH = 2000            # physics bounds passed by main.py:140
HEIGHT = 1000
FPS = 60
TITLE = "Particle Simulator"

COLOR_MIN_TEMP = 0.0
COLOR_MAX_TEMP = 1.0    # also the explosion threshold (particle_system.py:658)

# (normalised temperature, RGB) keyframes of the colour ramp (constants.py:37-45)
COLOR_GRADIENT_KEYFRAMES = [
    (0.0, (0, 0, 0)), (0.1, (75, 0, 130)), (0.25, (0, 0, 255)), (0.4, (0, 255, 0)),
    (0.55, (255, 255, 0)), (0.7, (255, 0, 0)), (1.0, (255, 255, 255)),
]

TRAIL_EFFECT_COLOR = (0, 0, 0, 30)
BLOOM_RADIUS = 20
BLOOM_INTENSITY = 30

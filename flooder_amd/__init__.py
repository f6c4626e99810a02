"""flooder_amd - MI355X-native coverage-radius sweep of the Flood complex.

Drop-in for the public names of plus-rkwitt/flooder (``flooder/__init__.py:1-19``): the two hot-path functions
``flood_complex`` and ``generate_landmarks``, plus ``save_to_disk`` and the synthetic generators; the console
script is ``python -m flooder_amd.cli``.  ROCm tensors run hand-written HIP kernels for gfx950
(``libflooder_hip.so``, C ABI in ``include/flooder_hip.h``); see DESIGN.md.
"""

from .core import (flood_complex, generate_landmarks, generate_grid, generate_uniform_weights, PointIndex,
                   index_from_host, forget_index)
from .simplex_tree import SimplexTree, DelaunayComplex
from .io import save_to_disk
from .synthetic import (
    generate_swiss_cheese_points,
    generate_annulus_points_2d,
    generate_noisy_torus_points_3d,
    generate_figure_eight_points_2d,
)

__version__ = "0.1"

__all__ = [
    "flood_complex",
    "generate_landmarks",
    "generate_grid",
    "generate_uniform_weights",
    "PointIndex",
    "index_from_host",
    "forget_index",
    "SimplexTree",
    "DelaunayComplex",
    "save_to_disk",
    "generate_swiss_cheese_points",
    "generate_annulus_points_2d",
    "generate_noisy_torus_points_3d",
    "generate_figure_eight_points_2d",
]

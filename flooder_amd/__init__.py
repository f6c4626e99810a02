"""flooder_amd - MI355X-native coverage-radius sweep of the Flood complex.

Drop-in for the two hot-path functions of plus-rkwitt/flooder (``flooder/__init__.py:2``):
``flood_complex`` and ``generate_landmarks``.  ROCm tensors run hand-written HIP kernels for gfx950
(``libflooder_hip.so``, C ABI in ``include/flooder_hip.h``); see DESIGN.md.
"""

from .core import flood_complex, generate_landmarks, generate_grid, generate_uniform_weights
from .simplex_tree import SimplexTree, DelaunayComplex

__version__ = "0.1"

__all__ = [
    "flood_complex",
    "generate_landmarks",
    "generate_grid",
    "generate_uniform_weights",
    "SimplexTree",
    "DelaunayComplex",
]

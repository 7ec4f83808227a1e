"""gaussianip_amd — MI355X-native hot path of GaussianIP (rasterizer + guidance step) behind the reference's API.

Importing the package does not load the HIP library; the first rasterizer call does, and fails loudly if it has
not been built (no CPU fallback on the product path).
"""
__version__ = "0.1.0"

from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                         rasterize_views)


def install_dropin():
    """Make `import diff_gaussian_rasterization` and `import simple_knn._C` resolve to this package."""
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")
    if d not in sys.path:
        sys.path.insert(0, d)

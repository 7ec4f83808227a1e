"""Drop-in for the `diff_gaussian_rasterization` package: put gaussianip_amd/dropin on PYTHONPATH (or call
gaussianip_amd.install_dropin()) and the reference's import lines
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
(gaussiansplatting/gaussian_renderer/__init__.py:14, gs_renderer.py:10-13) resolve to the MI355X implementation."""
from gaussianip_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                       rasterize_gaussians)

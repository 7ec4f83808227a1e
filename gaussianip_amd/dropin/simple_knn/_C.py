"""Drop-in for `simple_knn._C` (gaussiansplatting/scene/gaussian_model.py:9 imports distCUDA2 from here)."""
from gaussianip_amd.knn import distCUDA2  # noqa: F401
